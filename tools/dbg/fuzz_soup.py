"""Probe, not a test: closest hit and any hit of all three traversal modes against the oracle on many seeded
triangle soups (the generator of tests/test_gpu_trace.py), including rays that start on the surfaces (two
generations of bounce-like rays from the hit points).
   python tools/dbg/fuzz_soup.py FIRST LAST"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests import util
from tests.test_gpu_trace import _soup

first, last = int(sys.argv[1]), int(sys.argv[2])
total = 0
for seed in range(first, last + 1):
    t0 = time.time()
    rng, scale, meshes, mats = _soup(seed)
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    allv = np.concatenate([m[0] for m in meshes])
    n = 20000
    o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)]
    sets = {"random": (o, d), "at_vertices": (o, (tgt - o).astype(np.float32))}
    # bounce-like rays: from the hit points of the aimed rays, in random directions, twice
    so, sd = sets["at_vertices"]
    for gen in range(2):
        ref = w.castray(np.ascontiguousarray(so), np.ascontiguousarray(sd))
        hit = ref["tri"] >= 0
        if hit.sum() < 10: break
        p = (so[hit] + sd[hit] * ref["t"][hit, None]).astype(np.float32)   # x = o + d*t in float, camera.cpp:68
        nd = rng.standard_normal(p.shape).astype(np.float32)
        sets["surface%d" % gen] = (p, nd)
        so, sd = p, nd
    bad_seed = 0
    for label, (ro, rd) in sets.items():
        ro = np.ascontiguousarray(ro, np.float32); rd = np.ascontiguousarray(rd, np.float32)
        ref = w.castray(ro, rd)
        for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
            got = ds.trace_closest(ro, rd, mode)
            hit = ref["tri"] >= 0
            bad = (util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"]) | \
                  (((util.bits(ref["u"]) != util.bits(got["u"])) | (util.bits(ref["v"]) != util.bits(got["v"]))) & hit)
            anyh = ds.trace_any(ro, rd, mode)
            bad_any = (hit != anyh)
            if bad.any() or bad_any.any():
                bad_seed += int(bad.sum()) + int(bad_any.sum())
                i = int(np.nonzero(bad | bad_any)[0][0])
                print("  seed %d %s mode %d: closest %d any %d of %d; first ray %d o %s d %s ref (t %r tri %d) got (t %r tri %d) any %d" % (
                    seed, label, mode, bad.sum(), bad_any.sum(), len(bad), i, ro[i].tolist(), rd[i].tolist(),
                    float(ref["t"][i]), ref["tri"][i], float(got["t"][i]), got["tri"][i], anyh[i]), flush=True)
    ds.close()
    total += bad_seed
    print("seed %d: %d meshes, %d triangles, scale %g, %d sets, mismatches %d (%.1f s)" % (
        seed, len(meshes), len(allv) // 3, scale, len(sets), bad_seed, time.time() - t0), flush=True)
print("TOTAL mismatches", total)

"""Probe, not a test: HitData::hits (object.cpp:203,281-282) of pt_trace_hits against the oracle on seeded
triangle soups, exact and pruned walk, random / vertex-aimed / surface rays.
   python tools/dbg/fuzz_hits.py FIRST LAST"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests.test_gpu_trace import _soup
first, last = int(sys.argv[1]), int(sys.argv[2])
total = 0
for seed in range(first, last + 1):
    rng, scale, meshes, mats = _soup(seed)
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    allv = np.concatenate([m[0] for m in meshes])
    n = 20000
    o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)]
    sets = {"random": (o, d), "at_vertices": (o, (tgt - o).astype(np.float32))}
    ref = w.castray(np.ascontiguousarray(o), np.ascontiguousarray(sets["at_vertices"][1]))
    hit = ref["tri"] >= 0
    if hit.sum() >= 10:
        p = (o[hit] + sets["at_vertices"][1][hit] * ref["t"][hit, None]).astype(np.float32)
        sets["surface"] = (p, rng.standard_normal(p.shape).astype(np.float32))
    bad_seed = 0
    for label, (ro, rd) in sets.items():
        ro = np.ascontiguousarray(ro, np.float32); rd = np.ascontiguousarray(rd, np.float32)
        r = w.castray(ro, rd, want_hits=True)
        for mode in (0, 1):
            got = ds.trace_hits(ro, rd, mode=mode)
            nb = int((got != r["hits"]).sum())
            if nb:
                bad_seed += nb
                i = int(np.nonzero(got != r["hits"])[0][0])
                print("  seed %d %s mode %d: %d differ; ray o %s d %s ref hits %d (t %r tri %d) got %d" % (
                    seed, label, mode, nb, ro[i].tolist(), rd[i].tolist(), r["hits"][i], float(r["t"][i]), r["tri"][i], got[i]), flush=True)
    ds.close()
    total += bad_seed
    if bad_seed or seed % 25 == 0: print("seed %d: mismatches %d" % (seed, bad_seed), flush=True)
print("TOTAL mismatches", total)

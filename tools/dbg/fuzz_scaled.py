"""Probe, not a test: the structured small scenes (axis-aligned walls: flat gates) at several scales, rays that
start on the surfaces with direction vectors of random length, many of them nearly parallel to the surface
they leave; closest hit and any hit of the three traversal modes against the oracle.
   python tools/dbg/fuzz_scaled.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes
from tests import util

builders = {"box": lambda: scenes.box_room(10), "atrium": lambda: scenes.atrium(24, 6, 16, 8),
            "terrain": lambda: scenes.terrain_lights(48, 20), "blob": lambda: scenes.dense_blob(48)}
total = 0
for name, make in builders.items():
    for scale in (0.01, 1.0, 100.0, 3000.0):
        t0 = time.time()
        meshes, mats = make()
        meshes = [(np.ascontiguousarray(v * np.float32(scale), np.float32), t, m) for (v, t, m) in meshes]
        w = O.World.from_meshes(meshes, mats)
        ds = api.DeviceScene(w.export())
        rng = np.random.default_rng(int(scale * 100) + len(name))
        allv = np.concatenate([m[0] for m in meshes])
        lo, hi = allv.min(0), allv.max(0)
        n = 40000
        o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
        d = (allv[rng.integers(0, len(allv), n)] - o).astype(np.float32)
        bad_scene = 0
        for gen in range(3):
            o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
            ref = w.castray(o, d)
            hit = ref["tri"] >= 0
            for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
                got = ds.trace_closest(o, d, mode)
                bad = (util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"]) | \
                      (((util.bits(ref["u"]) != util.bits(got["u"])) | (util.bits(ref["v"]) != util.bits(got["v"]))) & hit)
                bad_any = hit != ds.trace_any(o, d, mode)
                if bad.any() or bad_any.any():
                    bad_scene += int(bad.sum()) + int(bad_any.sum())
                    i = int(np.nonzero(bad | bad_any)[0][0])
                    print("  %s x%g gen %d mode %d: closest %d any %d; ray o %s d %s ref (t %r tri %d) got (t %r tri %d)" % (
                        name, scale, gen, mode, bad.sum(), bad_any.sum(), o[i].tolist(), d[i].tolist(), float(ref["t"][i]), ref["tri"][i],
                        float(got["t"][i]), got["tri"][i]), flush=True)
            if hit.sum() < 10: break
            p = (o[hit] + d[hit] * ref["t"][hit, None]).astype(np.float32)
            nd = rng.standard_normal(p.shape).astype(np.float32)
            # every other ray nearly parallel to a coordinate plane, and lengths from 1e-2 to 1e2
            k = rng.integers(0, 3, len(nd)); graze = rng.random(len(nd)) < 0.5
            nd[graze, k[graze]] *= np.float32(1e-4)
            nd *= (10.0 ** rng.uniform(-2, 2, (len(nd), 1))).astype(np.float32)
            o, d = p, nd
        ds.close()
        total += bad_scene
        print("%s x%g: mismatches %d (%.1f s)" % (name, scale, bad_scene, time.time() - t0), flush=True)
print("TOTAL mismatches", total)

"""Fast and exact traversal must return the same closest hit (t, u, v bits, triangle) and the same
any-hit answer for every ray.  Random rays plus rays that start on surfaces (second-generation
rays built from the first hits), on the named large scenes.  usage: stress_fast_vs_exact.py [n_million]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import api, host, scenes

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4_000_000
bad = 0
for name in ("box1k", "atrium260k", "terrain1m", "blob7m"):
    meshes, mats = scenes.CONFIGS[name]["scene"]()
    s = host.Scene.from_meshes(meshes, mats)
    ds = s.device_scene(0)
    rng = np.random.default_rng(hash(name) % 1000)
    allv = np.concatenate([m[0] for m in meshes])
    lo, hi = allv.min(0) - 0.5, allv.max(0) + 0.5
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[rng.random(n) < 0.02, 0] = 0.0          # some axis-parallel directions
    d[rng.random(n) < 0.02, 2] = 0.0
    for gen in range(2):
        t0 = time.time()
        a = ds.trace_closest(o, d, mode=api.PT_TRACE_EXACT)
        b = ds.trace_closest(o, d, mode=api.PT_TRACE_FAST)
        ha = ds.trace_any(o, d, mode=api.PT_TRACE_EXACT)
        hb = ds.trace_any(o, d, mode=api.PT_TRACE_FAST)
        diff = (a["tri"] != b["tri"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32)) | \
               (a["u"].view(np.uint32) != b["u"].view(np.uint32)) | (a["v"].view(np.uint32) != b["v"].view(np.uint32))
        dany = np.asarray(ha) != np.asarray(hb)
        anyhit_consistent = (np.asarray(ha) != 0) == (a["tri"] >= 0)
        print("%-10s gen %d: %d rays, %d hits, closest mismatches %d, any-hit mismatches %d, any != (closest hit) %d  (%.1f s)"
              % (name, gen, len(o), int((a["tri"] >= 0).sum()), int(diff.sum()), int(dany.sum()), int((~anyhit_consistent).sum()),
                 time.time() - t0), flush=True)
        bad += int(diff.sum()) + int(dany.sum()) + int((~anyhit_consistent).sum())
        hit = a["tri"] >= 0
        o = (o[hit] + d[hit] * a["t"][hit, None]).astype(np.float32)   # start on the surfaces that were hit
        d = rng.standard_normal((len(o), 3)).astype(np.float32)
        n2 = len(o)
        if n2 == 0:
            break
    s.close()
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)

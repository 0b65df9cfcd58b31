"""Probe, not a test: small renders of many seeded triangle soups against the oracle, bit for bit, in fast and
exact traversal, with random integrator settings (bounce cap, cosine sampling, thin lens, background).
   python tools/dbg/fuzz_render.py FIRST LAST"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests import util
from tests.test_gpu_trace import _soup

first, last = int(sys.argv[1]), int(sys.argv[2])
tot_px = tot_bad = 0
for seed in range(first, last + 1):
    t0 = time.time()
    rng, scale, meshes, mats = _soup(seed)
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    W, H, spp = 48, 40, 3
    cam = dict(origin=(float(rng.uniform(-1, 1)) * scale, float(rng.uniform(-1, 1)) * scale, -1.5 * scale),
               euler=(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), 0.0))
    if rng.random() < 0.5:  # a wide viewport: camera rays with long direction vectors
        cam["vw"] = int(rng.choice([4, 16, 64])); cam["vh"] = cam["vw"] * 3 // 4
    kw = {}
    if rng.random() < 0.5: kw["max_depth"] = int(rng.integers(1, 6))
    if rng.random() < 0.5: kw["sampling"] = 1
    if rng.random() < 0.3: kw["bgcolor"] = (0.2, 0.3, 0.4)
    if rng.random() < 0.3:
        cam["aperture"] = 0.05 * scale; cam["focus"] = 1.5 * scale
    ref, rc = w.render(O.default_camera(**cam), W, H, spp, seed=seed, opt=O.default_options(**kw), rng=0, form=1)
    line = []
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_EXACT):
        got, gc = ds.render(api.make_camera(**cam), W, H, spp, seed=seed, opt=api.make_options(trace_mode=mode, **kw))
        fin_r, fin_g = np.isfinite(ref).all(-1), np.isfinite(got).all(-1)
        same = (util.bits(ref) == util.bits(got)).all(-1) | (~fin_r & ~fin_g)
        nb = int((~same).sum())
        tot_px += same.size; tot_bad += nb
        line.append("mode %d: %d px differ, nonfinite ref %d got %d, rays %d/%d" % (mode, nb, (~fin_r).sum(), (~fin_g).sum(), gc.rays_closest, rc.rays_closest))
        if nb:
            y, x = np.argwhere(~same)[0]
            line.append("first (%d,%d) ref %s got %s" % (x, y, ref[y, x].tolist(), got[y, x].tolist()))
    ds.close()
    print("seed %d scale %g %s %s | %s (%.1f s)" % (seed, scale, {k: v for k, v in cam.items() if k != "origin"}, kw, "; ".join(line), time.time() - t0), flush=True)
print("TOTAL differing pixels %d of %d" % (tot_bad, tot_px))

"""GPU image against the oracle (same counter-based random numbers, same operation order) at a
larger size than the test-suite uses: radiance sums compared as bit patterns.
usage: parity_large.py [W H spp]     (box1k scene and a small atrium; needs the built oracle and a GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes

W, H, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (960, 540, 8)
bad = False
for name, scene in (("box1k", scenes.box_room(10)), ("atrium24", scenes.atrium(24, 6, 16, 8)), ("terrain48x20", scenes.terrain_lights(48, 20))):
    w = O.World.from_meshes(*scene)
    ds = api.DeviceScene(w.export())
    kw = dict(vw=16, vh=9)
    if name.startswith("terrain"):
        kw.update(origin=(0, 0.5, 0), euler=(0.15, 0, 0))
    t0 = time.time()
    ref, rc = w.render(O.default_camera(**kw), W, H, spp, seed=7, rng=0, form=1)
    t1 = time.time()
    got, gc = ds.render(api.make_camera(**kw), W, H, spp, seed=7)
    t2 = time.time()
    same = (np.ascontiguousarray(ref).view(np.uint32) == np.ascontiguousarray(got).view(np.uint32)).all(-1)
    frac = 1.0 - same.mean()
    rel = abs(float(ref.mean()) - float(got.mean())) / max(1e-12, abs(float(ref.mean())))
    print("%-13s %dx%d %d spp: %.4f%% of pixels differ in any bit, mean differs by %.2e, rays %d vs %d (oracle %.1f s, GPU %.2f s)"
          % (name, W, H, spp, 100 * frac, rel, rc.rays_closest + rc.rays_shadow, gc.rays_closest + gc.rays_shadow, t1 - t0, t2 - t1), flush=True)
    bad = bad or frac > 2e-3 or rel > 1e-3
    ds.close()
sys.exit(1 if bad else 0)

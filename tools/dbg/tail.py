import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes
from tests import util
w = util.world("box")
cam = api.make_camera(vw=16, vh=9)
W, H, spp = 160, 90, 5
ref, rc = w.render(O.default_camera(vw=16, vh=9), W, H, spp, seed=9, rng=0, form=1)
def run(env, mode=1):
    for k in ("PT_NO_RESIDENT", "PT_TAIL", "PT_SPLIT", "PT_QUORUM_ANY", "PT_QUORUM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ds = api.DeviceScene(w.export())
    img, c = ds.render(cam, W, H, spp, seed=9, opt=api.make_options(trace_mode=mode))
    ds.close()
    bad = (util.bits(img) != util.bits(ref)).any(-1)
    print(env, "mode", mode, "pixels differing from oracle:", int(bad.sum()), "rays", c.rays_closest, c.rays_shadow, "oracle", rc.rays_closest, rc.rays_shadow, flush=True)
    return img
for mode in (1, 0, 2):
    run({}, mode)
    run({"PT_TAIL": "0"}, mode)
    run({"PT_TAIL": "1000000000"}, mode)
    run({"PT_NO_RESIDENT": "1", "PT_TAIL": "0"}, mode)
    run({"PT_NO_RESIDENT": "1", "PT_TAIL": "1000000000"}, mode)
    run({"PT_NO_RESIDENT": "1", "PT_TAIL": "1000000000", "PT_QUORUM_ANY": "32"}, mode)
    run({"PT_NO_RESIDENT": "1", "PT_TAIL": "0", "PT_SPLIT": "0"}, mode)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes, host
from tests import util
meshes, mats = scenes.dense_blob(96)
w = O.World.from_meshes(meshes, mats)
os.environ["PT_NO_RESIDENT"] = "1"
ds = api.DeviceScene(w.export())
print(ds.info)
o, d = util.primary_rays(O.default_camera(vw=16, vh=9), 96, 54)
ref = w.castray(o, d)
for mode in (1, 2, 0):
    got = ds.trace_closest(o, d, mode)
    bad = (util.bits(ref["u"]) != util.bits(got["u"])) & (ref["tri"] >= 0)
    badt = util.bits(ref["t"]) != util.bits(got["t"])
    print("mode", mode, "bad uv", bad.sum(), "bad t", badt.sum(), "bad tri", (ref["tri"] != got["tri"]).sum(), "hits", (ref["tri"] >= 0).sum())
    idx = np.nonzero(bad)[0][:6]
    for i in idx:
        print(i, ref["tri"][i], got["tri"][i], ref["t"][i], got["t"][i], ref["u"][i], got["u"][i], ref["v"][i], got["v"][i])

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes
from tests import util
for name, mk in (("box40", lambda: scenes.box_room(40)), ("atrium_small", lambda: scenes.atrium(24, 6, 16, 8)), ("blob_small", lambda: scenes.dense_blob(96))):
    meshes, mats = mk()
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    allv = np.concatenate([m[0] for m in meshes])
    rng = np.random.default_rng(4)
    n = 200000
    for far in (1, 10, 100, 1000, 10000, 100000):
        o = (rng.standard_normal((n, 3)) * far * 3 + np.float32([0, 0, 2])).astype(np.float32)
        tgt = allv[rng.integers(0, len(allv), n)] + rng.normal(0, 0.05, (n, 3))
        d = (tgt - o).astype(np.float32)
        ref = w.castray(o, d)
        out = []
        for mode in (0, 1):
            got = ds.trace_closest(o, d, mode)
            bad = (util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"])
            out.append(int(bad.sum()))
        print(name, "origins ~%g x scene: hits %d, mismatches exact %d fast %d" % (far, (ref["tri"] >= 0).sum(), out[0], out[1]), flush=True)
    ds.close()

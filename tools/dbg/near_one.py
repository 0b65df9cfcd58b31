import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes
from tests import util
n = 1000000
meshes, mats = scenes.dense_blob(96)
w = O.World.from_meshes(meshes, mats)
ds = api.DeviceScene(w.export())
allv = np.concatenate([m[0] for m in meshes])
M = float(np.abs(allv).max())
rng = np.random.default_rng(11)
# the same draws as near_limit.py makes before it reaches this scene
for prev in (scenes.box_room(40), scenes.atrium(24, 6, 16, 8)):
    pass
def rays(allv, M, rng):
    u = rng.standard_normal((n, 3)); u /= np.abs(u).max(1, keepdims=True)
    o = (u * rng.uniform(3.0, 7.99, (n, 1)) * M).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)] + rng.normal(0, 0.02 * M, (n, 3)) * (rng.random((n, 1)) < 0.5)
    return o, (tgt - o).astype(np.float32)
o, d = rays(allv, M, np.random.default_rng(11))
ref = w.castray(o, d)
for mode, nm in ((0, 'exact'), (1, 'fast'), (2, 'fast-binary')):
    got = ds.trace_closest(o, d, mode)
    bad = np.nonzero((util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"]))[0]
    print(nm, 'mismatches', len(bad))
    for i in bad[:5]:
        print('  ray', i, 'o', o[i], 'd', d[i], 'ref t %r tri %d u %r v %r' % (float(ref["t"][i]), ref["tri"][i], float(ref["u"][i]), float(ref["v"][i])),
              'got t %r tri %d' % (float(got["t"][i]), got["tri"][i]), 'max|o|/M', np.abs(o[i]).max() / M)
print('info', {k: ds.info[k] for k in ('n_tris', 'n_ill', 'boxes_nested') if k in ds.info})

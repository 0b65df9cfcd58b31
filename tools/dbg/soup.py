import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests import util
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(1000 + seed)
n_mesh = int(rng.integers(1, 9)); scale = float(10.0 ** rng.integers(-2, 3))
meshes, mats = [], []
for m in range(n_mesh):
    nt = int(rng.integers(1, 400))
    c = rng.uniform(-1, 1, (nt, 1, 3))
    v = (c + rng.normal(0, rng.uniform(0.02, 0.5), (nt, 3, 3))) * scale
    if rng.random() < 0.5: v = np.round(v / (0.125 * scale)) * (0.125 * scale)
    k = rng.random(nt) < 0.05; v[k, 2] = v[k, 1]
    if m > 0 and rng.random() < 0.5:
        prev = meshes[int(rng.integers(0, m))][0].reshape(-1, 3, 3); v[: min(nt, len(prev))] = prev[: min(nt, len(prev))]
    v = np.ascontiguousarray(v.reshape(-1, 3), np.float32)
    meshes.append((v, np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3), m)); mats.append(((0.5,)*3, (0,)*3))
w = O.World.from_meshes(meshes, mats)
ds = api.DeviceScene(w.export())
allv = np.concatenate([m[0] for m in meshes])
n = 20000
o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(np.float32)
d = rng.standard_normal((n, 3)).astype(np.float32)
tgt = allv[rng.integers(0, len(allv), n)]
ro = np.ascontiguousarray(o, np.float32); rd = np.ascontiguousarray((tgt - o).astype(np.float32))
ref = w.castray(ro, rd)
for mode in (0, 1, 2):
    got = ds.trace_closest(ro, rd, mode)
    bad = np.nonzero((util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"]))[0]
    print("mode", mode, "bad", bad)
    for i in bad:
        print("  ray", i, "o", ro[i].tolist(), "d", rd[i].tolist())
        print("  ref t %r tri %d u %r v %r" % (float(ref["t"][i]), ref["tri"][i], float(ref["u"][i]), float(ref["v"][i])))
        print("  got t %r tri %d u %r v %r" % (float(got["t"][i]), got["tri"][i], float(got["u"][i]), float(got["v"][i])))
        tv = allv.reshape(-1, 3, 3)
        for t in (ref["tri"][i], got["tri"][i]):
            if t >= 0: print("   tri", t, tv[t].tolist())
        one = ds.trace_closest(ro[i:i+1], rd[i:i+1], mode)
        print("  alone: t %r tri %d" % (float(one["t"][0]), one["tri"][0]))

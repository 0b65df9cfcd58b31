"""Probe: one ray of a seeded soup in the three traversal modes, with the triangles involved.
   python tools/dbg/soup_ray.py SEED ox oy oz dx dy dz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests.test_gpu_trace import _soup
seed = int(sys.argv[1])
o = np.array([[float(x) for x in sys.argv[2:5]]], np.float32)
d = np.array([[float(x) for x in sys.argv[5:8]]], np.float32)
rng, scale, meshes, mats = _soup(seed)
w = O.World.from_meshes(meshes, mats)
ds = api.DeviceScene(w.export())
allv = np.concatenate([m[0] for m in meshes]).reshape(-1, 3, 3)
ref = w.castray(o, d)
print("env", {k: v for k, v in os.environ.items() if k.startswith("PT_")}, "scale", scale, "max|coord|", np.abs(allv).max())
print("ref  t %r tri %d u %r v %r" % (float(ref["t"][0]), ref["tri"][0], float(ref["u"][0]), float(ref["v"][0])))
for mode in (0, 1, 2):
    g = ds.trace_closest(o, d, mode)
    print("mode %d t %r tri %d any %d" % (mode, float(g["t"][0]), g["tri"][0], ds.trace_any(o, d, mode)[0]))
for t in sorted(set([int(ref["tri"][0])] + [int(ds.trace_closest(o, d, 1)["tri"][0])])):
    if t >= 0:
        print("tri", t, allv[t].tolist(), "bounds lo", allv[t].min(0).tolist(), "hi", allv[t].max(0).tolist())
# the same ray with a normalised direction
n = d / np.linalg.norm(d)
r2 = w.castray(o, n.astype(np.float32)); g2 = ds.trace_closest(o, n.astype(np.float32), 1)
print("normalised d: ref t %r tri %d, fast t %r tri %d" % (float(r2["t"][0]), r2["tri"][0], float(g2["t"][0]), g2["tri"][0]))
ds.close()

"""How much would ordering the rays of a pass buy?  Closest-hit rate of one population of bounce-like rays (origins on
surfaces, random directions) in four orders: as the pixels produced them, shuffled, sorted by origin cell (Morton), and
sorted by direction octant + origin cell.  usage: coherence.py CONFIG"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pathtracer_amd import api, host, scenes
from oracle import pt_oracle as O

name = sys.argv[1]
cfg = scenes.CONFIGS[name]
meshes, mats = cfg["scene"]()
hs = host.Scene.from_meshes(meshes, mats)
ds = hs.device_scene(0)
W, H = 1920, 1080
cam = api.make_camera(**cfg["cam"])
d0 = api.camera_rays(cam, W, H).reshape(-1, 3)
o0 = np.tile(np.float32(cfg["cam"]["origin"]), (len(d0), 1))
rng = np.random.default_rng(1)
def bounce(o, d):
    r = ds.trace_closest(o, d)
    ok = r["tri"] >= 0
    x = (o[ok] + d[ok] * r["t"][ok, None]).astype(np.float32)
    nd = rng.standard_normal((len(x), 3)).astype(np.float32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    return x, nd.astype(np.float32)
o1, d1 = bounce(o0, d0)
o2, d2 = bounce(o1, d1)
o2, d2 = bounce(o2, d2)
o2 = np.repeat(o2, 8, axis=0)          # eight rays from every origin, neighbours in the "as produced" order
d2 = rng.standard_normal((len(o2), 3)).astype(np.float32)
d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
n = len(o2)
print(name, "rays", n, flush=True)
dev = torch.device("cuda", 0)
def morton(o):
    lo, hi = o.min(0), o.max(0)
    q = np.clip(((o - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32), 0, 1023)
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
m = morton(o2)
octant = ((d2[:, 0] < 0).astype(np.uint64) | ((d2[:, 1] < 0).astype(np.uint64) << 1) | ((d2[:, 2] < 0).astype(np.uint64) << 2))
orders = {"as produced": np.arange(n), "shuffled": rng.permutation(n), "origin cell": np.argsort(m, kind="stable"),
          "octant, then origin cell": np.argsort((octant << 30) | m, kind="stable"),
          "coarse cell (2^15), then octant, then fine": np.argsort(((m >> 15) << 33) | (octant << 30) | (m & 0x7fff), kind="stable")}
for label, idx in orders.items():
    o, d = o2[idx], d2[idx]
    cols = [torch.from_numpy(np.ascontiguousarray(a[:, i])).to(dev) for a in (o, d) for i in range(3)]
    t = torch.empty(n, dtype=torch.float32, device=dev)
    tri = torch.empty(n, dtype=torch.int32, device=dev)
    args = (n, [c.data_ptr() for c in cols], [t.data_ptr(), tri.data_ptr(), 0, 0])
    ds.trace_closest_dev(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ds.trace_closest_dev(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-45s %.2f ms  %.2f G rays/s" % (label, ms, n / ms / 1e6), flush=True)

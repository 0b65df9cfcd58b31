"""Shared helpers of the parity tests: seeded ray sets and scene plumbing."""
import numpy as np

from oracle import pt_oracle as O
from pathtracer_amd import scenes


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


_WORLDS = {}


def world(name):
    """Oracle world for a named small scene (cached per process)."""
    if name not in _WORLDS:
        if name == "plane":
            import os
            w = O.World.from_obj(os.path.join(os.path.dirname(__file__), "golden", "plane.obj"))
        elif name == "box":
            w = O.World.from_meshes(*scenes.box_room(10))
        elif name == "box40":
            w = O.World.from_meshes(*scenes.box_room(40))
        elif name == "atrium_small":
            w = O.World.from_meshes(*scenes.atrium(24, 6, 16, 8))
        elif name == "blob_small":
            w = O.World.from_meshes(*scenes.dense_blob(96))
        elif name == "terrain_small":
            w = O.World.from_meshes(*scenes.terrain_lights(48, 20))
        else:
            raise KeyError(name)
        _WORLDS[name] = w
    return _WORLDS[name]


def primary_rays(cam, W, H):
    d = O.camera_rays(cam, W, H).reshape(-1, 3)
    o = np.tile(np.array(cam.origin[:], np.float32), (len(d), 1))
    return o, d


def random_rays(n, seed, lo=(-2, -2, -1), hi=(2, 2, 5)):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    return o, d


def axis_rays(n, seed, lo=(-2, -2, -1), hi=(2, 2, 5)):
    """Directions with one or two exactly-zero components (AABB::intersect skips those axes)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    kill = rng.integers(0, 3, n)
    d[np.arange(n), kill] = 0
    kill2 = rng.integers(0, 3, n)
    two = rng.random(n) < 0.3
    d[np.arange(n)[two], kill2[two]] = 0
    d[(d == 0).all(1)] = (0, 0, 1)
    return o, d


def surface_rays(w, n, seed, cam=None):
    """Rays that start ON surfaces (no offset, camera.cpp:155): hit points of primary rays with
    random directions, which exercises the t < 1e-6 rejection and boxes entered at t ~ 0."""
    cam = cam or O.default_camera()
    o, d = primary_rays(cam, 192, 192)
    r = w.castray(o, d)
    ok = r["tri"] >= 0
    x = (o[ok] + d[ok] * r["t"][ok, None]).astype(np.float32)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(x), n)
    nd = rng.standard_normal((n, 3)).astype(np.float32)
    return x[idx], nd


def lattice_rays(k=41):
    """Rays through the corners and edges of the wall tessellation of box_room(10): many exact
    ties in t between neighbouring triangles."""
    g = np.linspace(-2, 2, k).astype(np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    tgt = np.stack([X.ravel(), Y.ravel(), np.full(X.size, 5, np.float32)], 1)
    o = np.zeros_like(tgt)
    o[:, 2] = -0.5
    d = (tgt - o).astype(np.float32)
    o2 = tgt.copy()
    o2[:, 2] = 1.0
    d2 = np.tile(np.array([0, 0, 1], np.float32), (len(tgt), 1))
    return np.concatenate([o, o2]), np.concatenate([d, d2])


def GOLD_PHILOX():
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "plane_kat.json")))
    return [([int(x, 16) for x in k["ctr"]], [int(x, 16) for x in k["key"]], [int(x, 16) for x in k["out"]])
            for k in g["philox4x32_10"]]

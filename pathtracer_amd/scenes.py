"""Deterministic synthetic scenes for the parity tests and the benchmark.

A scene is ``(meshes, materials)``:
  meshes    list of (verts[nv,3] float32, tris[nt,3] uint32 with ids local to the mesh, material id)
            -- one entry per OBJ ``o`` group, i.e. per reference ``Polygon`` (object.cpp:451-527);
  materials list of (kd[3], ke[3]) -- the two Material fields the integrator reads
            (object.cpp:144-156).

Every emitter has a non-zero area when projected on the XY plane: the reference's
Triangle::area (object.cpp:395-404) is that projection, and 1/area is a factor of the
direct-light term (camera.cpp:113-114).
"""
import numpy as np


def _grid_quad(origin, du, dv, n, m=None):
    """n x m quads spanning origin + a*du + b*dv, a,b in [0,1]; two triangles per quad."""
    m = n if m is None else m
    a = np.linspace(0.0, 1.0, n + 1)
    b = np.linspace(0.0, 1.0, m + 1)
    A, B = np.meshgrid(a, b, indexing="ij")
    o = np.asarray(origin, np.float64)
    du = np.asarray(du, np.float64)
    dv = np.asarray(dv, np.float64)
    verts = o[None, None, :] + A[..., None] * du[None, None, :] + B[..., None] * dv[None, None, :]
    verts = verts.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    v00 = (i * (m + 1) + j).ravel()
    v10 = v00 + (m + 1)
    v01 = v00 + 1
    v11 = v10 + 1
    tris = np.empty((n * m * 2, 3), np.uint32)
    tris[0::2] = np.stack([v00, v10, v11], 1)
    tris[1::2] = np.stack([v00, v11, v01], 1)
    return verts, tris


def _cylinder(cx, cz, radius, y0, y1, seg, rings):
    th = np.linspace(0.0, 2 * np.pi, seg, endpoint=False)
    ys = np.linspace(y0, y1, rings + 1)
    T, Y = np.meshgrid(th, ys, indexing="ij")
    verts = np.stack([cx + radius * np.cos(T), Y, cz + radius * np.sin(T)], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(seg), np.arange(rings), indexing="ij")
    i2 = (i + 1) % seg
    v00 = (i * (rings + 1) + j).ravel()
    v10 = (i2 * (rings + 1) + j).ravel()
    v01 = v00 + 1
    v11 = v10 + 1
    tris = np.empty((seg * rings * 2, 3), np.uint32)
    tris[0::2] = np.stack([v00, v10, v11], 1)
    tris[1::2] = np.stack([v00, v11, v01], 1)
    return verts, tris


def _emitter_quad(x0, x1, y0, y1, z):
    """Vertical quad in the plane z=const whose geometric normal (v1-v0)x(v2-v0) points to -z."""
    verts = np.array([[x0, y0, z], [x0, y1, z], [x1, y1, z], [x1, y0, z]], np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    return verts, tris


WHITE, RED, GREEN = (0.8, 0.8, 0.8), (0.8, 0.1, 0.1), (0.1, 0.8, 0.1)


def box_room(n=10, light_ke=10.0):
    """Inward-facing room x,y in [-2,2], z in [-1,5], open towards -z, five walls of n x n quads and
    one emissive unit quad just in front of the back wall.  n=10 gives 1 002 triangles in 6 meshes."""
    mats = [(WHITE, (0, 0, 0)), (RED, (0, 0, 0)), (GREEN, (0, 0, 0)), ((0.5, 0.5, 0.5), (light_ke,) * 3)]
    meshes = []
    v, t = _grid_quad((-2, -2, -1), (4, 0, 0), (0, 0, 6), n)   # floor
    meshes.append((v, t, 0))
    v, t = _grid_quad((-2, 2, -1), (4, 0, 0), (0, 0, 6), n)    # ceiling
    meshes.append((v, t, 0))
    v, t = _grid_quad((-2, -2, 5), (4, 0, 0), (0, 4, 0), n)    # back wall
    meshes.append((v, t, 0))
    v, t = _grid_quad((-2, -2, -1), (0, 4, 0), (0, 0, 6), n)   # left wall
    meshes.append((v, t, 1))
    v, t = _grid_quad((2, -2, -1), (0, 4, 0), (0, 0, 6), n)    # right wall
    meshes.append((v, t, 2))
    v, t = _emitter_quad(-0.5, 0.5, 0.5, 1.5, 4.99)
    meshes.append((v, t, 3))
    return meshes, mats


def atrium(n=150, columns=12, seg=48, rings=30, light_ke=10.0):
    """The room of box_room with finely tessellated walls plus two rows of columns, each column its
    own mesh.  Defaults: 5*150*150*2 + 12*48*30*2 + 2 = 259 562 triangles in 18 meshes."""
    meshes, mats = box_room(n, light_ke)
    light = meshes.pop()
    mats = mats + [((0.7, 0.7, 0.6), (0, 0, 0))]
    per_row = columns // 2
    for r, x in enumerate((-1.2, 1.2)):
        for k in range(per_row):
            z = 0.8 + (4.0 - 0.8) * (k + 0.5) / per_row
            v, t = _cylinder(x, z, 0.18, -2.0, 2.0, seg, rings)
            meshes.append((v, t, 4))
    meshes.append(light)
    return meshes, mats


def dense_blob(n=1871, seed=7, light_ke=10.0):
    """One mesh: a latitude-longitude sphere of radius 1.5 about (0,0,3.5) with seeded radial noise of
    0.3 grid spacings (a finely bumpy surface, not spikes), n x n quads (n=1871 gives 6 997 540
    triangles, 3.5 M shared vertices), and one vertical emitter."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.0, np.pi, n + 1)
    ph = np.linspace(0.0, 2 * np.pi, n + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = 1.5 * (1.0 + (0.3 * np.pi / n) * rng.standard_normal(T.shape))
    r[:, -1] = r[:, 0]
    r[0, :] = r[0, 0]
    r[-1, :] = r[-1, 0]
    verts = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T), 3.5 + r * np.sin(T) * np.sin(P)], -1)
    verts = verts.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v00 = (i * (n + 1) + j).ravel()
    v10 = v00 + (n + 1)
    v01 = v00 + 1
    v11 = v10 + 1
    tris = np.empty((n * n * 2, 3), np.uint32)
    tris[0::2] = np.stack([v00, v10, v11], 1)
    tris[1::2] = np.stack([v00, v11, v01], 1)
    # the quads of the first and last row touch a pole with two corners: one of their two triangles has two
    # identical vertices there; those 2 n zero-area triangles are left out (a sphere's cap is a fan)
    keep = np.ones(len(tris), bool)
    keep[1:2 * n:2] = False
    keep[2 * n * (n - 1)::2] = False
    tris = np.ascontiguousarray(tris[keep])
    mats = [((0.7, 0.7, 0.75), (0, 0, 0)), ((0.5, 0.5, 0.5), (light_ke,) * 3)]
    ev, et = _emitter_quad(-0.5, 0.5, 1.8, 2.8, 6.0)
    return [(verts, tris, 0), (ev, et, 1)], mats


def blob_room(n=1871, seed=7, light_ke=10.0):
    """The dense blob of dense_blob(n) standing inside the room of box_room(10): every primary ray hits
    something and paths bounce between the walls and the finely tessellated surface, so most rays walk the
    deep tree (n=1871: 6 997 540 + 1 002 triangles in 7 meshes; the room's emitter is the only light)."""
    room, mats = box_room(10, light_ke)
    blob, _ = dense_blob(n, seed, light_ke)
    mats = mats + [((0.7, 0.7, 0.75), (0, 0, 0))]
    v, t, _ = blob[0]
    return room + [(v, t, len(mats) - 1)], mats


def terrain_lights(n=707, lights=64, seed=11, light_ke=4.0):
    """A height-field terrain of n x n quads (n=707 gives 999 698 triangles) under `lights` vertical
    emissive quads, each its own mesh, so the top-level tree is a real tree and the direct-light loop
    (camera.cpp:96) runs over many lights."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-8.0, 8.0, n + 1)
    zs = np.linspace(0.0, 16.0, n + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = -1.5 + 0.35 * np.sin(1.3 * X) * np.cos(0.9 * Z) + 0.15 * np.sin(3.1 * X + 1.0) * np.sin(2.7 * Z)
    Y = Y + 0.01 * rng.standard_normal(Y.shape)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v00 = (i * (n + 1) + j).ravel()
    v10 = v00 + (n + 1)
    v01 = v00 + 1
    v11 = v10 + 1
    tris = np.empty((n * n * 2, 3), np.uint32)
    tris[0::2] = np.stack([v00, v10, v11], 1)
    tris[1::2] = np.stack([v00, v11, v01], 1)
    mats = [((0.6, 0.55, 0.5), (0, 0, 0))]
    meshes = [(verts, tris, 0)]
    side = int(np.ceil(np.sqrt(lights)))
    for k in range(lights):
        gx, gz = k % side, k // side
        x = -7.0 + 14.0 * (gx + 0.5) / side
        z = 1.0 + 14.0 * (gz + 0.5) / side
        col = 0.5 + 0.5 * rng.random(3)
        mats.append(((0.5, 0.5, 0.5), tuple(float(light_ke * c) for c in col)))
        ev, et = _emitter_quad(x - 0.25, x + 0.25, 0.6, 1.1, z)
        meshes.append((ev, et, len(mats) - 1))
    return meshes, mats


def count_tris(meshes):
    return int(sum(len(m[1]) for m in meshes))


def write_obj(path, meshes, materials):
    """Writes the scene in the OBJ/MTL dialect the reference's parser accepts (object.cpp:555-702):
    `o` before any vertex, global 1-based indices, faces as a/b/c, decimals without exponent."""
    mtl = path[:-4] + ".mtl" if path.endswith(".obj") else path + ".mtl"
    with open(mtl, "w") as f:
        for i, (kd, ke) in enumerate(materials):
            f.write("newmtl m%d\nKd %.6f %.6f %.6f\nKe %.6f %.6f %.6f\n" % ((i,) + tuple(kd) + tuple(ke)))
    base = 0
    with open(path, "w") as f:
        f.write("mtllib %s\n" % mtl.split("/")[-1])
        for k, (v, t, m) in enumerate(meshes):
            f.write("o mesh%d\n" % k)
            for p in v:
                f.write("v %.6f %.6f %.6f\n" % (p[0], p[1], p[2]))
            f.write("vt 0.000000 0.000000\nvn 0.0000 0.0000 1.0000\n")
            f.write("usemtl m%d\n" % m)
            for a, b, c in t:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1 + base, k + 1, k + 1, b + 1 + base, k + 1, k + 1,
                                                           c + 1 + base, k + 1, k + 1))
            base += len(v)


def write_textured_obj(path, meshes, materials, textures, planar_axes=(0, 2)):
    """Like write_obj, with texture coordinates and diffuse maps: `textures` maps a material id to
    an [H,W,3] uint8 array, written as binary PPM beside the OBJ and named by map_Kd.  Every
    vertex gets the texture coordinate (p[a] - min_a) / (max_a - min_a) of its own mesh for the two
    `planar_axes`, so faces use the a/b/c form with b == a."""
    stem = path[:-4] if path.endswith(".obj") else path
    with open(stem + ".mtl", "w") as f:
        for i, (kd, ke) in enumerate(materials):
            f.write("newmtl m%d\nKd %.6f %.6f %.6f\nKe %.6f %.6f %.6f\n" % ((i,) + tuple(kd) + tuple(ke)))
            if i in textures:
                t = np.ascontiguousarray(textures[i], np.uint8)
                name = "%s_tex%d.ppm" % (stem.split("/")[-1], i)
                with open(stem.rsplit("/", 1)[0] + "/" + name if "/" in stem else name, "wb") as g:
                    g.write(b"P6\n%d %d\n255\n" % (t.shape[1], t.shape[0]))
                    g.write(t.tobytes())
                f.write("map_Kd %s\n" % name)
    base = 0
    with open(path, "w") as f:
        f.write("mtllib %s\n" % (stem + ".mtl").split("/")[-1])
        for k, (v, t, m) in enumerate(meshes):
            f.write("o mesh%d\n" % k)
            lo, hi = v.min(0), v.max(0)
            span = np.where(hi > lo, hi - lo, 1.0)
            for p in v:
                f.write("v %.6f %.6f %.6f\n" % (p[0], p[1], p[2]))
            for p in v:
                q = (p - lo) / span
                f.write("vt %.6f %.6f\n" % (q[planar_axes[0]], q[planar_axes[1]]))
            f.write("vn 0.0000 0.0000 1.0000\n")
            f.write("usemtl m%d\n" % m)
            for a, b, c in t:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1 + base, a + 1 + base, k + 1, b + 1 + base, b + 1 + base, k + 1,
                                                           c + 1 + base, c + 1 + base, k + 1))
            base += len(v)


def round_like_obj(meshes):
    """The vertices write_obj + a correctly rounding parser would give back."""
    out = []
    for v, t, m in meshes:
        txt = np.char.mod("%.6f", v.astype(np.float64))
        out.append((txt.astype(np.float64).astype(np.float32), t, m))
    return out


# Named benchmark / parity configurations: scene builder, camera, image size, samples.
CONFIGS = {
    "plane": dict(scene=None, W=256, H=256, spp=16, cam=dict(origin=(0, 0, 0), euler=(0, 0, 0), fov=65.0, vw=1, vh=1)),
    "box1k": dict(scene=lambda: box_room(10), W=1920, H=1080, spp=64,
                  cam=dict(origin=(0, 0, 0), euler=(0, 0, 0), fov=65.0, vw=16, vh=9)),
    # quoted with 5 bounces and the thin lens on
    "atrium260k": dict(scene=lambda: atrium(150), W=1920, H=1080, spp=256,
                       cam=dict(origin=(0, 0, 0), euler=(0, 0, 0), fov=65.0, vw=16, vh=9, aperture=0.05, focus=3.0),
                       opt=dict(max_depth=5)),
    "blob7m": dict(scene=lambda: blob_room(1871), W=1920, H=1080, spp=16,
                   cam=dict(origin=(0, 0, -0.5), euler=(0, 0, 0), fov=65.0, vw=16, vh=9)),
    "terrain1m": dict(scene=lambda: terrain_lights(707, 64), W=3840, H=2160, spp=4096,
                      cam=dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), fov=65.0, vw=16, vh=9)),
}

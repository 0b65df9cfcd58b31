"""ctypes front-end of the CPU oracle (oracle/pt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")


class PtNode(C.Structure):
    _fields_ = [("index_or_first", C.c_uint32), ("count", C.c_uint32),
                ("bmin", C.c_float * 3), ("bmax", C.c_float * 3)]


class PtMaterial(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ke", C.c_float * 3)]


class PtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("euler", C.c_float * 3), ("fov_deg", C.c_float),
                ("vw", C.c_int32), ("vh", C.c_int32), ("aperture", C.c_float), ("focus", C.c_float)]


class PtOptions(C.Structure):
    _fields_ = [("rr_prob", C.c_double), ("rr_min_depth", C.c_int32), ("max_depth", C.c_int32),
                ("sampling", C.c_int32), ("bgcolor", C.c_float * 3), ("trace_mode", C.c_int32),
                ("spp_stride", C.c_uint32), ("batch_spp", C.c_uint32)]


class PtCounters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("node_boxes", C.c_uint64), ("tri_tests", C.c_uint64), ("bounce_passes", C.c_uint64),
                ("shadow_node_boxes", C.c_uint64), ("shadow_tri_tests", C.c_uint64)]


class OrcRenderArgs(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
                ("seed", C.c_uint64), ("rng", C.c_int32), ("form", C.c_int32), ("threads", C.c_int32),
                ("running_mean", C.c_int32)]


NODE_DTYPE = np.dtype([("index_or_first", "<u4"), ("count", "<u4"), ("bmin", "<f4", 3), ("bmax", "<f4", 3)])
MAT_DTYPE = np.dtype([("kd", "<f4", 3), ("ke", "<f4", 3)])


def default_camera(origin=(0, 0, 0), euler=(0, 0, 0), fov=65.0, vw=1, vh=1, aperture=0.0, focus=1.0):
    c = PtCamera()
    c.origin[:] = origin
    c.euler[:] = euler
    c.fov_deg, c.vw, c.vh, c.aperture, c.focus = fov, vw, vh, aperture, focus
    return c


def default_options(**kw):
    o = PtOptions()
    o.rr_prob, o.rr_min_depth, o.max_depth, o.sampling = 0.7, 2, 0, 0
    o.bgcolor[:] = (0, 0, 0)
    o.trace_mode, o.spp_stride, o.batch_spp = 1, 1, 0
    for k, v in kw.items():
        if k == "bgcolor":
            o.bgcolor[:] = v
        else:
            setattr(o, k, v)
    return o


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_error.restype = C.c_char_p
        L.orc_error.argtypes = [C.c_void_p]
        L.orc_load_obj.restype = C.c_void_p
        L.orc_load_obj.argtypes = [C.c_char_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_from_arrays.restype = C.c_void_p
        L.orc_from_arrays.argtypes = [C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 6
        L.orc_n_textures.restype = C.c_uint32
        L.orc_n_textures.argtypes = [C.c_void_p]
        L.orc_texture_size.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_texture_data.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_export_textured.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_texel.restype = C.c_int
        L.orc_texel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_export.argtypes = [C.c_void_p] * 12
        L.orc_castray.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 12
        L.orc_camera_rays.argtypes = [C.POINTER(PtCamera), C.c_int32, C.c_int32, C.c_void_p]
        L.orc_philox.argtypes = [C.c_void_p] * 3
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtOptions),
                                 C.POINTER(OrcRenderArgs), C.c_void_p, C.POINTER(PtCounters)]
        L.orc_save_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.orc_hardware_threads.restype = C.c_int
        L.orc_roulette_pass.restype = C.c_int
        L.orc_roulette_pass.argtypes = [C.c_double, C.c_float]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class World:
    """The oracle's scene: meshes + both BVH levels, built as the reference builds them."""

    def __init__(self, handle):
        self._h = handle
        err = lib().orc_error(self._h)
        if err:
            msg = err.decode()
            lib().orc_free(self._h)
            self._h = None
            raise ValueError(msg)
        n = [C.c_uint32() for _ in range(6)]
        lib().orc_counts(self._h, *[C.byref(x) for x in n])
        (self.n_meshes, self.n_tris, self.n_tlas_nodes, self.n_blas_nodes, self.n_lights,
         self.n_mats) = [x.value for x in n]

    @classmethod
    def from_obj(cls, path):
        return cls(lib().orc_load_obj(os.fsencode(path)))

    @classmethod
    def from_meshes(cls, meshes, materials):
        """meshes: list of (verts[nv,3] f32, tris[nt,3] u32 local ids, material index);
        materials: list of (kd3, ke3)."""
        vo = np.zeros(len(meshes) + 1, np.uint32)
        to = np.zeros(len(meshes) + 1, np.uint32)
        for i, (v, t, _) in enumerate(meshes):
            vo[i + 1] = vo[i] + len(v)
            to[i + 1] = to[i] + len(t)
        verts = np.ascontiguousarray(np.concatenate([m[0] for m in meshes]).astype(np.float32))
        tris = np.ascontiguousarray(np.concatenate([m[1] for m in meshes]).astype(np.uint32))
        mm = np.array([m[2] for m in meshes], np.uint32)
        kd = np.ascontiguousarray(np.array([m[0] for m in materials], np.float32))
        ke = np.ascontiguousarray(np.array([m[1] for m in materials], np.float32))
        return cls(lib().orc_from_arrays(len(meshes), _p(vo), _p(verts), _p(to), _p(tris), _p(mm),
                                         len(materials), _p(kd), _p(ke)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    def export(self):
        """Arrays in the shape of a PtSceneDesc (include/ptgpu.h)."""
        d = dict(
            tlas_nodes=np.zeros(self.n_tlas_nodes, NODE_DTYPE),
            tlas_order=np.zeros(self.n_meshes, np.uint32),
            blas_node_offset=np.zeros(self.n_meshes + 1, np.uint32),
            blas_nodes=np.zeros(self.n_blas_nodes, NODE_DTYPE),
            tri_offset=np.zeros(self.n_meshes + 1, np.uint32),
            blas_order=np.zeros(self.n_tris, np.uint32),
            tri_verts=np.zeros((self.n_tris, 3, 3), np.float32),
            mesh_material=np.zeros(self.n_meshes, np.uint32),
            materials=np.zeros(self.n_mats, MAT_DTYPE),
            lights=np.zeros(self.n_lights, np.uint32),
            mesh_area=np.zeros(self.n_meshes, np.float64),
        )
        lib().orc_export(self._h, *[_p(d[k]) for k in (
            "tlas_nodes", "tlas_order", "blas_node_offset", "blas_nodes", "tri_offset", "blas_order",
            "tri_verts", "mesh_material", "materials", "lights", "mesh_area")])
        nt = lib().orc_n_textures(self._h)
        if nt:
            texs = []
            for i in range(nt):
                w, h = C.c_uint32(), C.c_uint32()
                lib().orc_texture_size(self._h, i, C.byref(w), C.byref(h))
                t = np.zeros((h.value, w.value, 3), np.uint8)
                lib().orc_texture_data(self._h, i, _p(t))
                texs.append(t)
            d["textures"] = texs
            d["material_texture"] = np.zeros(self.n_mats, np.int32)
            d["tri_uv"] = np.zeros((self.n_tris, 3, 2), np.float32)
            lib().orc_export_textured(self._h, _p(d["material_texture"]), _p(d["tri_uv"]))
        return d

    def texel(self, mesh, tri, u, v):
        """brdf taken from the diffuse map (camera.cpp:124-146), or None for an untextured mesh."""
        rgb = np.zeros(3, np.float32)
        return rgb if lib().orc_texel(self._h, mesh, tri, u, v, _p(rgb)) else None

    def castray(self, o, d, want_hits=False, want_counters=False):
        """o, d: [n,3] float32.  Returns dict(t, tri, u, v[, hits][, nodes, tris])."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        cols = [np.ascontiguousarray(o[:, i]) for i in range(3)] + [np.ascontiguousarray(d[:, i]) for i in range(3)]
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        hits = np.empty(n, np.int32) if want_hits else None
        cnt = np.zeros(2, np.uint64) if want_counters else None
        lib().orc_castray(self._h, n, *[_p(c) for c in cols], _p(t), _p(tri), _p(u), _p(v), _p(hits), _p(cnt))
        out = dict(t=t, tri=tri, u=u, v=v)
        if want_hits:
            out["hits"] = hits
        if want_counters:
            out["nodes"], out["tris"] = int(cnt[0]), int(cnt[1])
        return out

    def render(self, cam, W, H, spp_count, spp_begin=0, seed=0, opt=None, rng=0, form=1, threads=0,
               running_mean=False, accum=None):
        """Returns (image[H,W,3] float32, PtCounters).  running_mean=False adds radiance sums into
        `accum` (zeros when None); running_mean=True returns the reference's Camera::pixels."""
        opt = opt or default_options()
        a = OrcRenderArgs(W, H, spp_begin, spp_count, seed, rng, form, threads, int(running_mean))
        out = np.zeros((H, W, 3), np.float32) if accum is None else accum
        cnt = PtCounters()
        rc = lib().orc_render(self._h, C.byref(cam), C.byref(opt), C.byref(a), _p(out), C.byref(cnt))
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return out, cnt


def camera_rays(cam, W, H):
    out = np.empty((H, W, 3), np.float32)
    lib().orc_camera_rays(C.byref(cam), W, H, _p(out))
    return out


def philox(ctr, key):
    c = np.array(ctr, np.uint32)
    k = np.array(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().orc_philox(_p(c), _p(k), _p(o))
    return o


def save_ppm(path, pixels):
    p = np.ascontiguousarray(pixels, np.float32)
    return lib().orc_save_ppm(os.fsencode(path), _p(p), p.shape[1], p.shape[0])


def roulette_pass(rr_prob, draw):
    """The roulette test of camera.cpp:118-119 for one float draw."""
    return bool(lib().orc_roulette_pass(rr_prob, draw))


def hardware_threads():
    return lib().orc_hardware_threads()

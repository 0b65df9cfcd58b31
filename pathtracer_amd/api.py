"""ctypes binding of the render library's C ABI (include/ptgpu.h).

There is no CPU fallback: if libptgpu.so is missing or no gfx950 device is present,
loading or the first call fails with an exception.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libptgpu.so")

PT_TRACE_EXACT, PT_TRACE_FAST, PT_TRACE_FAST_BINARY = 0, 1, 2
PT_SAMPLE_UNIFORM, PT_SAMPLE_COSINE = 0, 1


class PtNode(C.Structure):
    _fields_ = [("index_or_first", C.c_uint32), ("count", C.c_uint32),
                ("bmin", C.c_float * 3), ("bmax", C.c_float * 3)]


class PtMaterial(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ke", C.c_float * 3)]


class PtTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.c_void_p)]


class PtSceneDesc(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("n_tris", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_tlas_nodes", C.c_uint32),
                ("tlas_nodes", C.c_void_p), ("tlas_order", C.c_void_p), ("blas_node_offset", C.c_void_p),
                ("blas_nodes", C.c_void_p), ("tri_offset", C.c_void_p), ("blas_order", C.c_void_p),
                ("tri_verts", C.c_void_p), ("mesh_material", C.c_void_p), ("materials", C.c_void_p),
                ("n_textures", C.c_uint32), ("textures", C.c_void_p), ("material_texture", C.c_void_p),
                ("tri_uv", C.c_void_p)]


class PtSceneInfo(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("n_tris", C.c_uint32), ("n_lights", C.c_uint32),
                ("n_wide_nodes", C.c_uint32), ("max_depth", C.c_uint32), ("n_fast_nodes", C.c_uint32),
                ("fast_max_depth", C.c_uint32), ("n_gates", C.c_uint32), ("device", C.c_uint32),
                ("device_bytes", C.c_uint64), ("boxes_nested", C.c_uint32), ("n_oct_nodes", C.c_uint32)]


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

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PtProfile(C.Structure):
    _fields_ = [("trace_ms", C.c_double), ("shade_ms", C.c_double), ("other_ms", C.c_double),
                ("trace_launches", C.c_uint64), ("shade_launches", C.c_uint64), ("tail_ms", C.c_double),
                ("tail_launches", C.c_uint64), ("trace_rays", C.c_uint64), ("visibility_ms", C.c_double),
                ("visibility_launches", C.c_uint64), ("visibility_rays", C.c_uint64)]


NODE_DTYPE = np.dtype([("index_or_first", "<u4"), ("count", "<u4"), ("bmin", "<f4", 3), ("bmax", "<f4", 3)])
MAT_DTYPE = np.dtype([("kd", "<f4", 3), ("ke", "<f4", 3)])

# Every symbol include/ptgpu.h declares.
EXPORTS = [
    "pt_last_error", "pt_device_count", "pt_options_default", "pt_camera_default", "pt_scene_upload",
    "pt_scene_free", "pt_scene_info", "pt_scene_check", "pt_scene_count_stats", "pt_scene_read_stats", "pt_scene_profile",
    "pt_scene_read_profile", "pt_trace_closest",
    "pt_trace_closest_dev", "pt_trace_any", "pt_trace_hits", "pt_camera_rays", "pt_render", "pt_render_dev", "pt_resolve",
    "pt_philox4x32", "pt_roulette_pass", "pt_render_dev_static", "pt_static_scratch_bytes", "pt_render_multi",
    "pt_static_max_pixels", "pt_shard_samples", "pt_resolve_dev", "pt_resolve_display_dev",
    "pt_device_alloc", "pt_device_free", "pt_device_clear", "pt_device_read", "pt_render_mean_dev", "pt_display_dev",
]


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ptgpu error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Loads libptgpu.so (raises OSError when it has not been built)."""
    global _lib
    if _lib is None:
        # PyTorch wheels carry a HIP runtime of their own.  If this library (linked against the
        # system's) initialises the GPU first and torch is imported afterwards, torch finds no
        # device; with torch loaded first both share one runtime.  So when torch is installed it
        # is imported here, before the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_device_count.restype = C.c_int
        L.pt_options_default.argtypes = [C.POINTER(PtOptions)]
        L.pt_camera_default.argtypes = [C.POINTER(PtCamera)]
        L.pt_scene_upload.argtypes = [C.POINTER(PtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.pt_scene_free.argtypes = [C.c_void_p]
        L.pt_scene_free.restype = None
        L.pt_scene_info.argtypes = [C.c_void_p, C.POINTER(PtSceneInfo)]
        L.pt_scene_check.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtSceneInfo)]
        L.pt_scene_count_stats.argtypes = [C.c_void_p, C.c_int]
        L.pt_scene_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pt_scene_profile.argtypes = [C.c_void_p, C.c_int]
        L.pt_scene_read_profile.argtypes = [C.c_void_p, C.POINTER(PtProfile)]
        L.pt_trace_closest.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 10 + [C.c_int]
        L.pt_trace_closest_dev.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 10 + [C.c_int, C.c_void_p]
        L.pt_trace_any.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.c_int]
        L.pt_trace_hits.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.c_int]
        L.pt_camera_rays.argtypes = [C.POINTER(PtCamera), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pt_render.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                C.c_uint64, C.POINTER(PtOptions), C.c_void_p, C.POINTER(PtCounters)]
        L.pt_render_dev.argtypes = L.pt_render.argtypes + [C.c_void_p]
        L.pt_render_dev_static.argtypes = L.pt_render_dev.argtypes
        L.pt_render_mean_dev.argtypes = L.pt_render_dev.argtypes
        L.pt_display_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.pt_static_scratch_bytes.restype = C.c_uint64
        L.pt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(PtCamera), C.c_int, C.c_int, C.c_uint32,
                                      C.c_uint32, C.c_uint64, C.POINTER(PtOptions), C.c_void_p, C.POINTER(PtCounters)]
        L.pt_resolve.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.pt_static_max_pixels.restype = C.c_uint64
        L.pt_static_max_pixels.argtypes = [C.c_void_p, C.c_int]
        L.pt_shard_samples.argtypes = [C.c_uint32] * 5 + [C.POINTER(C.c_uint32)] * 3
        L.pt_resolve_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.pt_resolve_display_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.pt_device_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.pt_device_free.argtypes = [C.c_int, C.c_void_p]
        L.pt_device_clear.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pt_device_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pt_philox4x32.argtypes = [C.c_void_p] * 3
        L.pt_philox4x32.restype = None
        L.pt_roulette_pass.argtypes = [C.c_double, C.c_float]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise PtError(rc, lib().pt_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    return _check(lib().pt_device_count())


def make_camera(origin=(0, 0, 0), euler=(0, 0, 0), fov=65.0, vw=1, vh=1, aperture=0.0, focus=1.0):
    c = PtCamera()
    lib().pt_camera_default(C.byref(c))
    c.origin[:] = origin
    c.euler[:] = euler
    c.fov_deg, c.vw, c.vh, c.aperture, c.focus = fov, vw, vh, aperture, focus
    return c


def make_options(**kw):
    o = PtOptions()
    lib().pt_options_default(C.byref(o))
    for k, v in kw.items():
        if k == "bgcolor":
            o.bgcolor[:] = v
        else:
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
    return o


def philox(ctr, key):
    c = np.array(ctr, np.uint32)
    k = np.array(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().pt_philox4x32(_p(c), _p(k), _p(o))
    return o


def camera_rays(cam, W, H, device=0):
    out = np.empty((H, W, 3), np.float32)
    _check(lib().pt_camera_rays(C.byref(cam), W, H, device, _p(out)))
    return out


def resolve(accum, spp):
    a = np.ascontiguousarray(accum, np.float32)
    out = np.empty_like(a)
    _check(lib().pt_resolve(_p(a), a.size, spp, _p(out)))
    return out


def shard_samples(spp_begin, spp_count, spp_stride, shard, n_shards):
    """pt_shard_samples: (begin, count, stride) of the samples shard `shard` of `n_shards` renders."""
    b, c, st = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(lib().pt_shard_samples(spp_begin, spp_count, spp_stride, shard, n_shards, C.byref(b), C.byref(c), C.byref(st)))
    return b.value, c.value, st.value


def resolve_dev(accum_ptr, n, spp, out_ptr, device=0, stream=None):
    _check(lib().pt_resolve_dev(C.c_void_p(accum_ptr), n, spp, C.c_void_p(out_ptr), device, C.c_void_p(stream or 0)))


def resolve_display_dev(accum_ptr, n_pixels, spp, rgb_ptr, device=0, stream=None):
    _check(lib().pt_resolve_display_dev(C.c_void_p(accum_ptr), n_pixels, spp, C.c_void_p(rgb_ptr), device,
                                        C.c_void_p(stream or 0)))


def display_dev(pixels_ptr, n_pixels, rgb_ptr, device=0, stream=None):
    _check(lib().pt_display_dev(C.c_void_p(pixels_ptr), n_pixels, C.c_void_p(rgb_ptr), device, C.c_void_p(stream or 0)))


def _make_desc(arrays):
    """PtSceneDesc over the numpy arrays of `arrays` (see DeviceScene); returns (desc, objects to keep alive)."""
    a = {
        "tlas_nodes": np.ascontiguousarray(arrays["tlas_nodes"], NODE_DTYPE),
        "tlas_order": np.ascontiguousarray(arrays["tlas_order"], np.uint32),
        "blas_node_offset": np.ascontiguousarray(arrays["blas_node_offset"], np.uint32),
        "blas_nodes": np.ascontiguousarray(arrays["blas_nodes"], NODE_DTYPE),
        "tri_offset": np.ascontiguousarray(arrays["tri_offset"], np.uint32),
        "blas_order": np.ascontiguousarray(arrays["blas_order"], np.uint32),
        "tri_verts": np.ascontiguousarray(arrays["tri_verts"], np.float32),
        "mesh_material": np.ascontiguousarray(arrays["mesh_material"], np.uint32),
        "materials": np.ascontiguousarray(arrays["materials"], MAT_DTYPE),
    }
    nm, nt = len(a["mesh_material"]), len(a["blas_order"])
    if nm == 0 or nt == 0:
        raise ValueError("empty scene")
    if len(a["tlas_order"]) != nm or len(a["blas_node_offset"]) != nm + 1 or len(a["tri_offset"]) != nm + 1:
        raise ValueError("per-mesh arrays disagree in length")
    if a["tri_verts"].size != nt * 9 or int(a["tri_offset"][-1]) != nt:
        raise ValueError("triangle arrays disagree in length")
    if int(a["blas_node_offset"][-1]) != len(a["blas_nodes"]):
        raise ValueError("blas_node_offset does not end at len(blas_nodes)")
    d = PtSceneDesc()
    d.n_meshes = len(a["mesh_material"])
    d.n_tris = len(a["blas_order"])
    d.n_materials = len(a["materials"])
    d.n_tlas_nodes = len(a["tlas_nodes"])
    for k, v in a.items():
        setattr(d, k, v.ctypes.data)
    texs = [np.ascontiguousarray(t, np.uint8) for t in arrays.get("textures", [])]
    if texs:
        for t in texs:
            if t.ndim != 3 or t.shape[2] != 3 or t.size == 0:
                raise ValueError("a texture must be a non-empty [H,W,3] uint8 array")
        mt = np.ascontiguousarray(arrays["material_texture"], np.int32)
        if len(mt) != d.n_materials:
            raise ValueError("material_texture must have one entry per material")
        recs = (PtTexture * len(texs))()
        for r, t in zip(recs, texs):
            r.width, r.height, r.rgb = t.shape[1], t.shape[0], t.ctypes.data
        d.n_textures, d.textures, d.material_texture = len(texs), C.addressof(recs), mt.ctypes.data
        if arrays.get("tri_uv") is not None:
            uv = np.ascontiguousarray(arrays["tri_uv"], np.float32)
            if uv.size != nt * 6:
                raise ValueError("tri_uv must be [n_tris,3,2]")
            d.tri_uv = uv.ctypes.data
    return d, (a, texs, locals().get('recs'), locals().get('mt'), locals().get('uv'))


def scene_check(arrays):
    """pt_scene_check: flattens the scene on the host, verifies the trees, returns the counts (no device needed)."""
    d, keep = _make_desc(arrays)
    info = PtSceneInfo()
    _check(lib().pt_scene_check(C.byref(d), C.byref(info)))
    return {k: int(getattr(info, k)) for k, _ in info._fields_}


def render_multi(scenes, cam, W, H, spp_count, spp_begin=0, seed=0, opt=None, accum=None):
    """pt_render_multi: the frame's samples dealt over the devices of `scenes` (DeviceScene copies of one scene,
    one per device), merged by one all-reduce and added to `accum`.  Returns (accum, PtCounters)."""
    if accum is None:
        accum = np.zeros((H, W, 3), np.float32)
    if accum.dtype != np.float32 or accum.shape != (H, W, 3) or not accum.flags.c_contiguous:
        raise ValueError("accum must be a contiguous float32 [H,W,3] array")
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    cnt = PtCounters()
    o = opt if opt is not None else make_options()
    _check(lib().pt_render_multi(handles, len(scenes), C.byref(cam), W, H, spp_begin, spp_count, seed, C.byref(o),
                                 _p(accum), C.byref(cnt)))
    return accum, cnt


class DeviceScene:
    """A scene resident in HBM.  `arrays` is a dict with the members of PtSceneDesc as numpy
    arrays (tlas_nodes, tlas_order, blas_node_offset, blas_nodes, tri_offset, blas_order, tri_verts,
    mesh_material, materials), and optionally textures (list of [H,W,3] uint8 arrays), material_texture
    (int32 per material, -1 = none) and tri_uv ([n_tris,3,2] float32)."""

    def __init__(self, arrays, device=0):
        self._h = None
        d, self._keep = _make_desc(arrays)
        h = C.c_void_p()
        _check(lib().pt_scene_upload(C.byref(d), device, C.byref(h)))
        self._h = h
        self.device = device
        info = PtSceneInfo()
        _check(lib().pt_scene_info(self._h, C.byref(info)))
        self.info = {k: int(getattr(info, k)) for k, _ in info._fields_}

    @classmethod
    def borrow(cls, handle, device, owner=None):
        """Wraps a PtScene* owned by someone else (it is not freed here)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(handle)
        self._owned = False
        self._owner = owner
        self.device = device
        info = PtSceneInfo()
        _check(lib().pt_scene_info(self._h, C.byref(info)))
        self.info = {k: int(getattr(info, k)) for k, _ in info._fields_}
        return self

    def close(self):
        if self._h and getattr(self, "_owned", True):
            lib().pt_scene_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def static_max_pixels(self, mode=PT_TRACE_FAST):
        return int(lib().pt_static_max_pixels(self._h, mode))

    def count_stats(self, enable):
        _check(lib().pt_scene_count_stats(self._h, int(enable)))

    def read_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().pt_scene_read_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile(self, enable):
        _check(lib().pt_scene_profile(self._h, int(enable)))

    def read_profile(self):
        p = PtProfile()
        _check(lib().pt_scene_read_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in p._fields_}

    @staticmethod
    def _cols(o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        if len(o) != len(d):
            raise ValueError("origins and directions differ in length")
        return [np.ascontiguousarray(o[:, i]) for i in range(3)] + [np.ascontiguousarray(d[:, i]) for i in range(3)]

    def trace_closest(self, o, d, mode=PT_TRACE_FAST):
        """Scene::castray for each ray: dict(t, tri, u, v); tri = -1 and t = inf on a miss."""
        cols = self._cols(o, d)
        n = len(cols[0])
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        _check(lib().pt_trace_closest(self._h, n, *[_p(c) for c in cols], _p(t), _p(tri), _p(u), _p(v), mode))
        return dict(t=t, tri=tri, u=u, v=v)

    def trace_hits(self, o, d, mode=PT_TRACE_FAST):
        """HitData::hits of Scene::castray for each ray (object.cpp:203,281-282)."""
        cols = self._cols(o, d)
        n = len(cols[0])
        hits = np.empty(n, np.int32)
        _check(lib().pt_trace_hits(self._h, n, *[_p(c) for c in cols], _p(hits), mode))
        return hits

    def trace_any(self, o, d, mode=PT_TRACE_FAST):
        cols = self._cols(o, d)
        n = len(cols[0])
        hit = np.empty(n, np.uint8)
        _check(lib().pt_trace_any(self._h, n, *[_p(c) for c in cols], _p(hit), mode))
        return hit.astype(bool)

    def render(self, cam, W, H, spp_count, spp_begin=0, seed=0, opt=None, accum=None):
        """Adds spp_count samples per pixel to `accum` ([H,W,3] float32 radiance sums, zeros when
        None).  Returns (accum, PtCounters)."""
        if accum is None:
            accum = np.zeros((H, W, 3), np.float32)
        if accum.dtype != np.float32 or accum.shape != (H, W, 3) or not accum.flags.c_contiguous:
            raise ValueError("accum must be a contiguous float32 [H,W,3] array")
        cnt = PtCounters()
        o = opt if opt is not None else make_options()
        _check(lib().pt_render(self._h, C.byref(cam), W, H, spp_begin, spp_count, seed, C.byref(o), _p(accum),
                               C.byref(cnt)))
        return accum, cnt

    def render_dev(self, cam, W, H, spp_count, accum_ptr, spp_begin=0, seed=0, opt=None, stream=None):
        """Same with `accum_ptr` an integer device address of W*H*3 floats on this scene's device."""
        cnt = PtCounters()
        o = opt if opt is not None else make_options()
        _check(lib().pt_render_dev(self._h, C.byref(cam), W, H, spp_begin, spp_count, seed, C.byref(o),
                                   C.c_void_p(accum_ptr), C.byref(cnt), C.c_void_p(stream or 0)))
        return cnt

    def render_mean_dev(self, cam, W, H, pass_count, pixels_ptr, pass_begin=0, seed=0, opt=None, stream=None):
        """pt_render_mean_dev: updates Camera::pixels (running mean of radiance*255) at `pixels_ptr` on the device."""
        cnt = PtCounters()
        o = opt if opt is not None else make_options()
        _check(lib().pt_render_mean_dev(self._h, C.byref(cam), W, H, pass_begin, pass_count, seed, C.byref(o),
                                        C.c_void_p(pixels_ptr), C.byref(cnt), C.c_void_p(stream or 0)))
        return cnt

    def render_dev_static(self, cam, W, H, spp_count, accum_ptr, spp_begin=0, seed=0, opt=None, stream=None):
        """render_dev through the entry point that allocates nothing (it works in the library's static
        device buffer; frames beyond its size raise PtError)."""
        cnt = PtCounters()
        o = opt if opt is not None else make_options()
        _check(lib().pt_render_dev_static(self._h, C.byref(cam), W, H, spp_begin, spp_count, seed, C.byref(o),
                                          C.c_void_p(accum_ptr), C.byref(cnt), C.c_void_p(stream or 0)))
        return cnt

    def trace_closest_dev(self, n, ptrs, out_ptrs, mode=PT_TRACE_FAST, stream=None):
        """ptrs = (ox,oy,oz,dx,dy,dz), out_ptrs = (t,tri,u,v): integer device addresses."""
        _check(lib().pt_trace_closest_dev(self._h, n, *[C.c_void_p(p) for p in ptrs],
                                          *[C.c_void_p(p or 0) for p in out_ptrs], mode, C.c_void_p(stream or 0)))

"""ctypes binding of the host-side library (libpthost.so): OBJ/MTL loading, two-level BVH
construction and the Scene / Camera classes that mirror the reference's modules
(object.cpp, scene.cpp, camera.cpp).  Rendering and ray queries go to the GPU library."""
import ctypes as C
import os

import numpy as np

from . import api

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpthost.so")
_lib = None
_CB = C.CFUNCTYPE(None)


def lib():
    global _lib
    if _lib is None:
        api.lib()  # libpthost.so links against libptgpu.so
        L = C.CDLL(LIB_PATH)
        L.pth_last_error.restype = C.c_char_p
        L.pth_scene_from_obj.restype = C.c_void_p
        L.pth_scene_from_obj.argtypes = [C.c_char_p]
        L.pth_scene_from_arrays.restype = C.c_void_p
        L.pth_scene_from_arrays.argtypes = [C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_void_p]
        L.pth_scene_from_cache.restype = C.c_void_p
        L.pth_scene_from_cache.argtypes = [C.c_char_p]
        L.pth_scene_save_cache.restype = C.c_int
        L.pth_scene_save_cache.argtypes = [C.c_void_p, C.c_char_p]
        L.pth_load_image.restype = C.c_int
        L.pth_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_free.restype = None
        L.pth_scene_desc.restype = C.POINTER(api.PtSceneDesc)
        L.pth_scene_desc.argtypes = [C.c_void_p]
        L.pth_scene_lights.restype = C.c_uint32
        L.pth_scene_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.pth_mesh_area.restype = C.c_double
        L.pth_mesh_area.argtypes = [C.c_void_p, C.c_uint32]
        L.pth_scene_device.restype = C.c_void_p
        L.pth_scene_device.argtypes = [C.c_void_p, C.c_int]
        L.pth_point_towards.argtypes = [C.POINTER(api.PtCamera), C.c_int, C.c_int, C.c_void_p]
        L.pth_point_towards.restype = None
        L.pth_camera_render.argtypes = [C.c_void_p, C.POINTER(api.PtCamera), C.POINTER(api.PtOptions), C.c_int, C.c_int,
                                        C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(api.PtCounters), _CB, C.c_int]
        L.pth_camera_render_display.argtypes = [C.c_void_p, C.POINTER(api.PtCamera), C.POINTER(api.PtOptions), C.c_int, C.c_int,
                                                C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, _CB, C.c_int]
        L.pth_save_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _arr(ptr, dtype, n):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Scene:
    """Host-side scene (meshes, materials, both BVH levels), built the way the reference builds
    it.  `device_scene()` uploads it and returns an api.DeviceScene view."""

    def __init__(self, handle):
        if not handle:
            raise ValueError(lib().pth_last_error().decode())
        self._h = handle
        self._dev = {}

    @classmethod
    def from_obj(cls, path):
        return cls(lib().pth_scene_from_obj(os.fsencode(path)))

    @classmethod
    def from_cache(cls, path):
        """A scene saved by save_cache(): no OBJ parsing, no tree build; uploads the same arrays."""
        return cls(lib().pth_scene_from_cache(os.fsencode(path)))

    def save_cache(self, path):
        if lib().pth_scene_save_cache(self._h, os.fsencode(path)) != 0:
            raise OSError(lib().pth_last_error().decode())

    @classmethod
    def from_meshes(cls, meshes, materials):
        vo = np.zeros(len(meshes) + 1, np.uint32)
        to = np.zeros(len(meshes) + 1, np.uint32)
        for i, (v, t, _) in enumerate(meshes):
            vo[i + 1] = vo[i] + len(v)
            to[i + 1] = to[i] + len(t)
        verts = np.ascontiguousarray(np.concatenate([m[0] for m in meshes]), np.float32)
        tris = np.ascontiguousarray(np.concatenate([m[1] for m in meshes]), np.uint32)
        mm = np.array([m[2] for m in meshes], np.uint32)
        kd = np.ascontiguousarray(np.array([m[0] for m in materials], np.float32))
        ke = np.ascontiguousarray(np.array([m[1] for m in materials], np.float32))
        return cls(lib().pth_scene_from_arrays(len(meshes), _p(vo), _p(verts), _p(to), _p(tris), _p(mm),
                                               len(materials), _p(kd), _p(ke)))

    def close(self):
        if getattr(self, "_h", None):
            lib().pth_scene_free(self._h)
            self._h = None
            self._dev = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export(self):
        """Copies of the arrays of the scene's PtSceneDesc, plus lights and mesh areas."""
        d = lib().pth_scene_desc(self._h).contents
        nm, nt = d.n_meshes, d.n_tris
        bno = _arr(d.blas_node_offset, np.uint32, nm + 1)
        out = dict(
            tlas_nodes=_arr(d.tlas_nodes, api.NODE_DTYPE, d.n_tlas_nodes),
            tlas_order=_arr(d.tlas_order, np.uint32, nm),
            blas_node_offset=bno,
            blas_nodes=_arr(d.blas_nodes, api.NODE_DTYPE, int(bno[-1])),
            tri_offset=_arr(d.tri_offset, np.uint32, nm + 1),
            blas_order=_arr(d.blas_order, np.uint32, nt),
            tri_verts=_arr(d.tri_verts, np.float32, nt * 9).reshape(nt, 3, 3),
            mesh_material=_arr(d.mesh_material, np.uint32, nm),
            materials=_arr(d.materials, api.MAT_DTYPE, d.n_materials),
        )
        if d.n_textures:
            recs = C.cast(d.textures, C.POINTER(api.PtTexture * d.n_textures)).contents
            out["textures"] = [_arr(r.rgb, np.uint8, r.width * r.height * 3).reshape(r.height, r.width, 3) for r in recs]
            out["material_texture"] = _arr(d.material_texture, np.int32, d.n_materials)
            out["tri_uv"] = _arr(d.tri_uv, np.float32, nt * 6).reshape(nt, 3, 2)
        lights = np.zeros(nm, np.uint32)
        k = lib().pth_scene_lights(self._h, _p(lights), nm)
        out["lights"] = lights[:k].copy()
        out["mesh_area"] = np.array([lib().pth_mesh_area(self._h, i) for i in range(nm)], np.float64)
        return out

    def device_scene(self, device=0):
        if device not in self._dev:
            h = lib().pth_scene_device(self._h, device)
            if not h:
                raise api.PtError(-2, lib().pth_last_error().decode())
            self._dev[device] = api.DeviceScene.borrow(h, device, owner=self)
        return self._dev[device]

    def render(self, cam, W, H, samples, seed=0, opt=None, device=0, callback=None, every=1):
        """Camera::render: returns (pixels[H,W,3] = mean radiance * 255, PtCounters)."""
        pixels = np.zeros((H, W, 3), np.float32)
        cnt = api.PtCounters()
        o = opt if opt is not None else api.make_options()
        cb = _CB(callback) if callback else _CB()
        rc = lib().pth_camera_render(self._h, C.byref(cam), C.byref(o), W, H, samples, seed, device, _p(pixels),
                                     C.byref(cnt), cb, every)
        if rc != 0:
            raise api.PtError(rc, lib().pth_last_error().decode())
        return pixels, cnt


def _render_display(self, cam, W, H, samples, seed=0, opt=None, device=0, callback=None, every=1):
    """Camera::render with the GUI callback's display conversion (main.cpp:43-47) done on the device: returns
    (rgb[H,W,3] uint8 = clamp((int)pixels, 0, 255), pixels[H,W,3]).  `callback(rgb)` sees the bytes after every
    `every` passes."""
    rgb = np.zeros((H, W, 3), np.uint8)
    pixels = np.zeros((H, W, 3), np.float32)
    o = opt if opt is not None else api.make_options()
    cb = _CB(lambda: callback(rgb.copy())) if callback else _CB()
    rc = lib().pth_camera_render_display(self._h, C.byref(cam), C.byref(o), W, H, samples, seed, device, _p(rgb),
                                         _p(pixels), cb, every)
    if rc != 0:
        raise api.PtError(rc, lib().pth_last_error().decode())
    return rgb, pixels


Scene.render_display = _render_display


def load_image(path):
    """An image file decoded the way a diffuse map is: [H,W,3] uint8.  PNG (8-bit, non-interlaced) and PPM."""
    w, h = C.c_uint32(), C.c_uint32()
    if lib().pth_load_image(os.fsencode(path), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError(lib().pth_last_error().decode())
    out = np.zeros((h.value, w.value, 3), np.uint8)
    if lib().pth_load_image(os.fsencode(path), C.byref(w), C.byref(h), _p(out), out.size) != 0:
        raise ValueError(lib().pth_last_error().decode())
    return out


def point_towards(cam, W, H):
    out = np.empty((H, W, 3), np.float32)
    lib().pth_point_towards(C.byref(cam), W, H, _p(out))
    return out


def save_ppm(path, pixels):
    p = np.ascontiguousarray(pixels, np.float32)
    lib().pth_save_ppm(os.fsencode(path), _p(p), p.shape[1], p.shape[0])

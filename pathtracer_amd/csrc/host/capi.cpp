// capi.cpp — a small C interface over the host-side Scene / Camera classes so that the
// Python package (ctypes) can drive them.  Not part of the render ABI (include/ptgpu.h).
#include <cstring>
#include <new>
#include <string>

#include "camera.h"

using namespace pt;

namespace {
thread_local std::string g_err;
struct Handle {
  Scene scene;
};
}  // namespace

extern "C" {

const char *pth_last_error(void) { return g_err.c_str(); }

void *pth_scene_from_obj(const char *path) {
  ParsedScene parsed;
  std::string err;
  if (!parseobj(path, parsed, err)) {
    g_err = err;
    return nullptr;
  }
  Handle *h = new (std::nothrow) Handle();
  if (!h) {
    g_err = "out of memory";
    return nullptr;
  }
  h->scene = Scene(std::move(parsed));
  return h;
}

// A scene read from a cache file written by pth_scene_save_cache (no parsing, no tree build).
void *pth_scene_from_cache(const char *path) {
  Handle *h = new (std::nothrow) Handle();
  if (!h) {
    g_err = "out of memory";
    return nullptr;
  }
  std::string err;
  if (!Scene::load_cache(path, h->scene, &err)) {
    g_err = err;
    delete h;
    return nullptr;
  }
  return h;
}

// Decodes an image file as a diffuse map is decoded; rgb may be null to query the size only.
int pth_load_image(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb, size_t cap) {
  Image img;
  std::string err;
  if (!load_image(path, img, err)) {
    g_err = err;
    return -1;
  }
  *width = img.width;
  *height = img.height;
  if (rgb) {
    if (cap < img.rgb.size()) {
      g_err = "buffer too small";
      return -1;
    }
    memcpy(rgb, img.rgb.data(), img.rgb.size());
  }
  return 0;
}

int pth_scene_save_cache(void *h, const char *path) {
  std::string err;
  if (!static_cast<Handle *>(h)->scene.save_cache(path, &err)) {
    g_err = err;
    return -1;
  }
  return 0;
}

// meshes as concatenated arrays: vert_off[n+1], verts[3*nv], tri_off[n+1], tris[3*nt] (ids local
// to the mesh), mesh_mat[n]; materials kd[3*m], ke[3*m]
void *pth_scene_from_arrays(uint32_t n_meshes, const uint32_t *vert_off, const float *verts, const uint32_t *tri_off,
                            const uint32_t *tris, const uint32_t *mesh_mat, uint32_t n_mats, const float *kd,
                            const float *ke) {
  ParsedScene parsed;
  for (uint32_t i = 0; i < n_mats; ++i) {
    Material m;
    m.diffuse = vec(kd[3 * i], kd[3 * i + 1], kd[3 * i + 2]);
    m.emission = vec(ke[3 * i], ke[3 * i + 1], ke[3 * i + 2]);
    parsed.materials.push_back(m);
  }
  parsed.meshes.resize(n_meshes);
  for (uint32_t i = 0; i < n_meshes; ++i) {
    Polygon &p = parsed.meshes[i];
    p.vertices.reserve(vert_off[i + 1] - vert_off[i]);
    for (uint32_t k = vert_off[i]; k < vert_off[i + 1]; ++k) p.vertices.push_back(vec(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]));
    p.triangles.reserve(tri_off[i + 1] - tri_off[i]);
    for (uint32_t k = tri_off[i]; k < tri_off[i + 1]; ++k) p.triangles.push_back({tris[3 * k], tris[3 * k + 1], tris[3 * k + 2]});
    if (mesh_mat[i] >= n_mats) {
      g_err = "material index out of range";
      return nullptr;
    }
    p.material = (int)mesh_mat[i];
    std::string err;
    if (!p.init(&err)) {
      g_err = err;
      return nullptr;
    }
  }
  if (parsed.meshes.empty()) {
    g_err = "no meshes";
    return nullptr;
  }
  Handle *h = new (std::nothrow) Handle();
  if (!h) {
    g_err = "out of memory";
    return nullptr;
  }
  h->scene = Scene(std::move(parsed));
  return h;
}

void pth_scene_free(void *h) { delete static_cast<Handle *>(h); }

// The scene as a PtSceneDesc; the pointers stay valid until pth_scene_free.
const PtSceneDesc *pth_scene_desc(void *h) { return &static_cast<Handle *>(h)->scene.desc(); }

uint32_t pth_scene_lights(void *h, uint32_t *out, uint32_t cap) {
  std::vector<int> l = static_cast<Handle *>(h)->scene.lights();
  for (size_t i = 0; i < l.size() && i < cap; ++i) out[i] = (uint32_t)l[i];
  return (uint32_t)l.size();
}

double pth_mesh_area(void *h, uint32_t mesh) {
  const Scene &sc = static_cast<Handle *>(h)->scene;
  return mesh < sc.desc().n_meshes ? sc.mesh_area(mesh) : 0.0;
}

// Uploads (once per device) and returns the PtScene of include/ptgpu.h, owned by the handle.
PtScene *pth_scene_device(void *h, int device) {
  try {
    return static_cast<Handle *>(h)->scene.device_scene(device);
  } catch (const std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// Camera::point_towards on the host for every pixel.
void pth_point_towards(const PtCamera *c, int W, int H, float *dir_xyz) {
  Camera cam(vec(c->origin[0], c->origin[1], c->origin[2]), vec(c->euler[0], c->euler[1], c->euler[2]));
  cam.OUTW = W; cam.OUTH = H; cam.vw = c->vw; cam.vh = c->vh; cam.fov = c->fov_deg;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      vec d = cam.point_towards(px, py);
      float *o = dir_xyz + ((size_t)py * W + px) * 3;
      o[0] = d.x; o[1] = d.y; o[2] = d.z;
    }
}

// Camera::render into pixels[H*W*3]; cb (may be null) is called after every `every` passes.
int pth_camera_render(void *h, const PtCamera *c, const PtOptions *opt, int W, int H, int samples, uint64_t seed,
                      int device, float *pixels, PtCounters *counters, void (*cb)(void), int every) {
  try {
    Camera cam(vec(c->origin[0], c->origin[1], c->origin[2]), vec(c->euler[0], c->euler[1], c->euler[2]));
    cam.OUTW = W; cam.OUTH = H; cam.vw = c->vw; cam.vh = c->vh; cam.fov = c->fov_deg;
    cam.aperture = c->aperture; cam.focus = c->focus;
    cam.seed = seed; cam.device = device;
    if (opt) {
      cam.options = *opt;
      cam.bgcolor = vec(opt->bgcolor[0], opt->bgcolor[1], opt->bgcolor[2]);
    }
    std::function<void()> f;
    if (cb) f = [&]() {
      memcpy(pixels, cam.pixels.data(), (size_t)W * H * 3 * sizeof(float));
      cb();
    };
    cam.render(static_cast<Handle *>(h)->scene, samples, f, every);
    memcpy(pixels, cam.pixels.data(), (size_t)W * H * 3 * sizeof(float));
    if (counters) *counters = cam.counters;
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// The same render with the display conversion on the device (Camera::display_only): rgb[H*W*3] receives the
// clamped bytes after every callback interval, pixels[H*W*3] (may be null) the float image at the end.
int pth_camera_render_display(void *h, const PtCamera *c, const PtOptions *opt, int W, int H, int samples, uint64_t seed,
                              int device, uint8_t *rgb, float *pixels, void (*cb)(void), int every) {
  try {
    Camera cam(vec(c->origin[0], c->origin[1], c->origin[2]), vec(c->euler[0], c->euler[1], c->euler[2]));
    cam.OUTW = W; cam.OUTH = H; cam.vw = c->vw; cam.vh = c->vh; cam.fov = c->fov_deg;
    cam.aperture = c->aperture; cam.focus = c->focus;
    cam.seed = seed; cam.device = device;
    cam.display_only = true;
    if (opt) {
      cam.options = *opt;
      cam.bgcolor = vec(opt->bgcolor[0], opt->bgcolor[1], opt->bgcolor[2]);
    }
    std::function<void()> f;
    if (cb) f = [&]() {
      memcpy(rgb, cam.display.data(), (size_t)W * H * 3);
      cb();
    };
    cam.render(static_cast<Handle *>(h)->scene, samples, f, every);
    memcpy(rgb, cam.display.data(), (size_t)W * H * 3);
    if (pixels) memcpy(pixels, cam.pixels.data(), (size_t)W * H * 3 * sizeof(float));
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

int pth_save_ppm(const char *path, const float *pixels, int W, int H) {
  Camera cam(vec(0, 0, 0));
  cam.OUTW = W; cam.OUTH = H;
  cam.pixels.resize((size_t)W * H);
  memcpy(static_cast<void *>(cam.pixels.data()), pixels, (size_t)W * H * 3 * sizeof(float));
  cam.save(path);
  return 0;
}

}  // extern "C"

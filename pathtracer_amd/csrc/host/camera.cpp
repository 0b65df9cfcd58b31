// camera.cpp — host-side Camera.
#include "camera.h"

#include <climits>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace pt {

Camera::Camera(const vec &o, const vec &d) : origin(o), direction(d) { pt_options_default(&options); }

vec Camera::point_towards(int px, int py) const {
  return vec((px - (scalar)OUTW / 2) * vw / OUTW,
             (py - (scalar)OUTH / 2) * -vh / OUTH,
             (scalar)(1 / (2 * std::tan(fov / 2 * (M_PI / 180)) / vw)))
      .rotate(direction);
}

PtCamera Camera::abi_camera() const {
  PtCamera c;
  pt_camera_default(&c);
  c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
  c.euler[0] = direction.x; c.euler[1] = direction.y; c.euler[2] = direction.z;
  c.fov_deg = fov;
  c.vw = vw; c.vh = vh;
  c.aperture = aperture; c.focus = focus;
  return c;
}

namespace {
// Device memory of one progressive render, released when render() returns or throws.
struct DeviceFrame {
  int device = 0;
  void *sums = nullptr, *out = nullptr;
  ~DeviceFrame() {
    if (sums) (void)pt_device_free(device, sums);
    if (out) (void)pt_device_free(device, out);
  }
};
void must(int rc, const char *what) {
  if (rc != PT_OK) throw std::runtime_error(std::string(what) + ": " + pt_last_error());
}
}  // namespace

void Camera::render(const Scene &scene, int samples, std::function<void()> cb, int passes_per_callback) {
  if (OUTW <= 0 || OUTH <= 0) throw std::runtime_error("Camera::render: bad image size");
  const size_t np = (size_t)OUTW * OUTH;
  pixels.assign(np, vec(0, 0, 0));  // camera.cpp:177-179
  display.clear();
  memset(&counters, 0, sizeof(counters));
  if (samples <= 0) return;
  PtOptions opt = options;
  opt.bgcolor[0] = bgcolor.x; opt.bgcolor[1] = bgcolor.y; opt.bgcolor[2] = bgcolor.z;
  const PtCamera cam = abi_camera();
  const uint32_t stride = opt.spp_stride ? opt.spp_stride : 1;
  const int step = cb ? (passes_per_callback > 0 ? passes_per_callback : 1) : samples;
  auto add_counters = [&](const PtCounters &c) {
    counters.paths += c.paths; counters.rays_closest += c.rays_closest; counters.rays_shadow += c.rays_shadow;
    counters.node_boxes += c.node_boxes; counters.tri_tests += c.tri_tests; counters.bounce_passes += c.bounce_passes;
  };
  if (devices.size() > 1) {
    // several devices: pt_render_multi merges the per-device sums into a host accumulator
    std::vector<PtScene *> devs;
    for (int d : devices) devs.push_back(scene.device_scene(d));
    std::vector<float> sums(np * 3, 0.0f);
    for (int done = 0; done < samples; done += step) {
      const int now = samples - done < step ? samples - done : step;
      PtCounters c;
      must(pt_render_multi(devs.data(), (int)devs.size(), &cam, OUTW, OUTH, spp_begin + (uint32_t)done * stride, (uint32_t)now, seed,
                           &opt, sums.data(), &c), "pt_render_multi");
      add_counters(c);
      pt_resolve(sums.data(), np * 3, (uint32_t)(done + now), reinterpret_cast<float *>(pixels.data()));
      if (display_only) {  // main.cpp:43: clamp((int)value, 0, 255); the merged sums are on the host here
        display.resize(np * 3);
        const float *px = reinterpret_cast<const float *>(pixels.data());
        for (size_t i = 0; i < np * 3; ++i) {
          const float v = px[i];
          const int t = (v >= 2147483648.0f || v < -2147483648.0f || v != v) ? INT_MIN : (int)v;  // cvttsd2si
          display[i] = (unsigned char)(t < 0 ? 0 : (t > 255 ? 255 : t));
        }
      }
      if (cb) cb();
    }
    std::puts("done");  // camera.cpp:199
    return;
  }
  // One device: the reference's accumulator itself, `pixels` as a running mean (camera.cpp:188-190), lives in
  // device memory over all passes and is updated there with the reference's arithmetic; after every callback
  // interval only what the callback reads is copied back: `pixels` (12 bytes a pixel), or with display_only the
  // clamped bytes of the GUI callback (main.cpp:43-47, 3 bytes a pixel), converted by a kernel.
  if (stride != 1) throw std::runtime_error("Camera::render: a shard of a frame (spp_stride > 1) has no running mean; use pt_render_dev");
  DeviceFrame f;
  f.device = devices.size() == 1 ? devices[0] : device;
  PtScene *ds = scene.device_scene(f.device);
  must(pt_device_alloc(f.device, np * 3 * sizeof(float), &f.sums), "pt_device_alloc");
  if (display_only) {
    must(pt_device_alloc(f.device, np * 3, &f.out), "pt_device_alloc");
    display.assign(np * 3, 0);
  }
  for (int done = 0; done < samples; done += step) {
    const int now = samples - done < step ? samples - done : step;
    const bool last = done + now >= samples;
    PtCounters c;
    must(pt_render_mean_dev(ds, &cam, OUTW, OUTH, spp_begin + (uint32_t)done, (uint32_t)now, seed, &opt,
                            static_cast<float *>(f.sums), &c, nullptr), "pt_render_mean_dev");
    add_counters(c);
    if (display_only) {
      must(pt_display_dev(static_cast<const float *>(f.sums), np, static_cast<uint8_t *>(f.out), f.device, nullptr), "pt_display_dev");
      must(pt_device_read(f.device, display.data(), f.out, np * 3, nullptr), "pt_device_read");
    }
    if (!display_only || last) must(pt_device_read(f.device, pixels.data(), f.sums, np * 3 * sizeof(float), nullptr), "pt_device_read");
    if (cb) cb();
  }
  std::puts("done");  // camera.cpp:199
}

void Camera::save(const std::string &filename) const {
  std::ofstream ofs(filename);
  ofs << "P3\n" << OUTW << " " << OUTH << "\n" << 255 << "\n";
  for (int y = 0; y < OUTH; ++y)
    for (int x = 0; x < OUTW; ++x) {
      const vec &p = pixels[(size_t)y * OUTW + x];
      ofs << (int)p.x << " " << (int)p.y << " " << (int)p.z << "\n";
    }
}

}  // namespace pt

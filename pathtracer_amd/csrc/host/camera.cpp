// camera.cpp — host-side Camera.
#include "camera.h"

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

void Camera::render(const Scene &scene, int samples, std::function<void()> cb, int passes_per_callback) {
  if (OUTW <= 0 || OUTH <= 0) throw std::runtime_error("Camera::render: bad image size");
  const size_t np = (size_t)OUTW * OUTH;
  pixels.assign(np, vec(0, 0, 0));  // camera.cpp:177-179
  memset(&counters, 0, sizeof(counters));
  if (samples <= 0) return;
  std::vector<PtScene *> devs;
  if (devices.size() > 1)
    for (int d : devices) devs.push_back(scene.device_scene(d));
  else
    devs.push_back(scene.device_scene(devices.size() == 1 ? devices[0] : device));
  PtOptions opt = options;
  opt.bgcolor[0] = bgcolor.x; opt.bgcolor[1] = bgcolor.y; opt.bgcolor[2] = bgcolor.z;
  const PtCamera cam = abi_camera();
  std::vector<float> sums(np * 3, 0.0f);
  const uint32_t stride = opt.spp_stride ? opt.spp_stride : 1;
  const int step = cb ? (passes_per_callback > 0 ? passes_per_callback : 1) : samples;
  for (int done = 0; done < samples; done += step) {
    const int now = samples - done < step ? samples - done : step;
    PtCounters c;
    if (pt_render_multi(devs.data(), (int)devs.size(), &cam, OUTW, OUTH, spp_begin + (uint32_t)done * stride, (uint32_t)now, seed, &opt,
                        sums.data(), &c) != PT_OK)
      throw std::runtime_error(std::string("pt_render_multi: ") + pt_last_error());
    counters.paths += c.paths; counters.rays_closest += c.rays_closest; counters.rays_shadow += c.rays_shadow;
    counters.node_boxes += c.node_boxes; counters.tri_tests += c.tri_tests; counters.bounce_passes += c.bounce_passes;
    // the reference keeps a running mean of radiance*255 (camera.cpp:188-190); sum/n*255 is
    // the same quantity up to float rounding
    pt_resolve(sums.data(), np * 3, (uint32_t)(done + now), reinterpret_cast<float *>(pixels.data()));
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

// camera.h — host-side Camera: the reference's `camera` module (camera.cpp) with
// pathtrace/render running on the GPU.
#ifndef PT_HOST_CAMERA_H
#define PT_HOST_CAMERA_H

#include <functional>
#include <string>
#include <vector>

#include "scene.h"

namespace pt {

class Camera {
 public:
  int OUTW = 256, OUTH = 256;      // camera.cpp:51 (fixed there; settable here)
  vec bgcolor = vec(0, 0, 0);      // camera.cpp:52
  std::vector<vec> pixels;         // OUTH*OUTW, row-major: pixels[py*OUTW+px] (camera.cpp:53)
  int vw = 1, vh = 1;              // camera.cpp:55
  vec origin, direction;           // camera.cpp:56 (direction = Euler angles in radians)
  scalar fov = 65;                 // camera.cpp:58
  // not in the reference: thin lens, integrator switches, sample sharding
  scalar aperture = 0, focus = 1;
  PtOptions options;
  uint64_t seed = 0;
  int device = 0;
  std::vector<int> devices;        // more than one entry: the samples are dealt over these devices (pt_render_multi)
  uint32_t spp_begin = 0;          // global index of this render's first sample (a shard of a larger frame)
  PtCounters counters{};           // work done by the last render()
  // Progressive display (main.cpp:40-52: the GUI's callback turns `pixels` into clamped 8-bit colours after every
  // pass).  With display_only the conversion runs on the device and every callback interval copies back only
  // these bytes, display[(py*OUTW+px)*3 + c]; `pixels` is then filled once, after the last pass.
  bool display_only = false;
  std::vector<unsigned char> display;

  explicit Camera(const vec &origin, const vec &direction = vec(0, 0, 0));

  // Camera::point_towards (camera.cpp:161-169), evaluated on the host.
  vec point_towards(int px, int py) const;

  // Camera::render (camera.cpp:171-200).  Clears `pixels`, renders `samples` samples per
  // pixel on the GPU and leaves radiance * 255 averaged over the samples in `pixels`.
  // With a callback the image is refreshed and cb() called after every `passes_per_callback`
  // samples (1 = after every pass, as the reference does).  On one device the radiance sums live in
  // device memory for the whole call; nothing is allocated or uploaded per pass.  Throws
  // std::runtime_error on a device failure.
  void render(const Scene &scene, int samples = 32, std::function<void()> cb = nullptr,
              int passes_per_callback = 1);

  // Camera::save (camera.cpp:204-218): plain PPM, integer truncation, no clamp.
  void save(const std::string &filename) const;

  PtCamera abi_camera() const;
};

}  // namespace pt
#endif

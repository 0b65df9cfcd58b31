// main.cpp — headless render driver: what RenderThread::run does in the reference's GUI
// (main.cpp:40-52: render 16 samples with a per-pass callback that converts `pixels` to an
// image), without Qt.  Scene path, image size, samples and camera come from the command
// line; the defaults are the reference's literals (main.cpp:28,42; camera.cpp:51,55,58).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "camera.h"

using namespace pt;

static void usage(const char *argv0) {
  std::fprintf(stderr,
               "usage: %s scene.obj [-o out.ppm] [--spp N] [--width W] [--height H] [--vw A --vh B]\n"
               "          [--origin x,y,z] [--euler a,b,c] [--fov deg] [--aperture r --focus d]\n"
               "          [--max-depth k] [--cosine] [--exact] [--seed s] [--device i] [--devices n] [--progress]\n"
               "          [--save-cache scene.ptscene]   (a .ptscene file is also accepted as the scene)\n"
               "          [--heatmap]   (instead of rendering: HitData::hits of every primary ray as a grey image)\n"
               "          [--display]   (write what the reference's window shows: colours clamped to 0..255, binary PPM)\n",
               argv0);
}

static bool parse3(const char *s, vec &v) { return std::sscanf(s, "%f,%f,%f", &v.x, &v.y, &v.z) == 3; }

// The display image of the GUI callback (main.cpp:43-47: clamp((int)value, 0, 255)), converted on the device
// by Camera::render (display_only) and written as a binary PPM.
static void write_display_ppm(const Camera &cam, const std::string &path) {
  std::ofstream ofs(path, std::ios::binary);
  ofs << "P6\n" << cam.OUTW << " " << cam.OUTH << "\n255\n";
  ofs.write(reinterpret_cast<const char *>(cam.display.data()), (std::streamsize)cam.display.size());
}

int main(int argc, char **argv) {
  if (argc < 2) {
    usage(argv[0]);
    return 2;
  }
  std::string scene_path = argv[1], out_path = "out.ppm", cache_out;
  bool heatmap = false;
  int spp = 16;
  bool progress = false, display = false;
  Camera cam(vec(0, 0, 0));
  for (int i = 2; i < argc; ++i) {
    auto need = [&](const char *flag) -> const char * {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "%s needs a value\n", flag);
        std::exit(2);
      }
      return argv[++i];
    };
    const std::string a = argv[i];
    if (a == "-o") out_path = need("-o");
    else if (a == "--spp") spp = std::atoi(need("--spp"));
    else if (a == "--width") cam.OUTW = std::atoi(need("--width"));
    else if (a == "--height") cam.OUTH = std::atoi(need("--height"));
    else if (a == "--vw") cam.vw = std::atoi(need("--vw"));
    else if (a == "--vh") cam.vh = std::atoi(need("--vh"));
    else if (a == "--fov") cam.fov = (scalar)std::atof(need("--fov"));
    else if (a == "--aperture") cam.aperture = (scalar)std::atof(need("--aperture"));
    else if (a == "--focus") cam.focus = (scalar)std::atof(need("--focus"));
    else if (a == "--max-depth") cam.options.max_depth = std::atoi(need("--max-depth"));
    else if (a == "--seed") cam.seed = std::strtoull(need("--seed"), nullptr, 10);
    else if (a == "--device") cam.device = std::atoi(need("--device"));
    else if (a == "--devices") {  // the samples dealt over devices 0..n-1, one all-reduce per callback
      const int nd = std::atoi(need("--devices"));
      cam.devices.clear();
      for (int d = 0; d < nd; ++d) cam.devices.push_back(d);
    }
    else if (a == "--cosine") cam.options.sampling = PT_SAMPLE_COSINE;
    else if (a == "--exact") cam.options.trace_mode = PT_TRACE_EXACT;
    else if (a == "--progress") progress = true;
    else if (a == "--save-cache") cache_out = need("--save-cache");
    else if (a == "--heatmap") heatmap = true;
    else if (a == "--display") display = true;
    else if (a == "--origin") { if (!parse3(need("--origin"), cam.origin)) { usage(argv[0]); return 2; } }
    else if (a == "--euler") { if (!parse3(need("--euler"), cam.direction)) { usage(argv[0]); return 2; } }
    else { usage(argv[0]); return 2; }
  }
  pt::verbose = true;
  try {
    Scene scene;
    std::string err;
    const bool is_cache = scene_path.size() > 8 && scene_path.compare(scene_path.size() - 8, 8, ".ptscene") == 0;
    if (is_cache) {
      if (!Scene::load_cache(scene_path, scene, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
      }
    } else {
      ParsedScene parsed;
      if (!parseobj(scene_path, parsed, err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
      }
      scene = Scene(std::move(parsed));
    }
    if (!cache_out.empty() && !scene.save_cache(cache_out, &err)) {
      std::fprintf(stderr, "%s\n", err.c_str());
      return 1;
    }
    std::printf("%u meshes, %zu triangles, %zu lights\n", scene.desc().n_meshes, scene.triangle_count(), scene.lights().size());
    if (heatmap) {
      // the reference keeps HitData::hits "for special visualizations" (object.cpp:203): how many
      // internal nodes lie above what each primary ray found
      const size_t n = (size_t)cam.OUTW * cam.OUTH;
      std::vector<float> col(6 * n);
      for (int y = 0; y < cam.OUTH; ++y)
        for (int x = 0; x < cam.OUTW; ++x) {
          const size_t i = (size_t)y * cam.OUTW + x;
          const vec d = cam.point_towards(x, y);
          col[i] = cam.origin.x; col[n + i] = cam.origin.y; col[2 * n + i] = cam.origin.z;
          col[3 * n + i] = d.x; col[4 * n + i] = d.y; col[5 * n + i] = d.z;
        }
      std::vector<int32_t> hits(n);
      if (pt_trace_hits(scene.device_scene(cam.device), n, &col[0], &col[n], &col[2 * n], &col[3 * n], &col[4 * n],
                        &col[5 * n], hits.data(), cam.options.trace_mode) != PT_OK)
        throw std::runtime_error(std::string("pt_trace_hits: ") + pt_last_error());
      int32_t mx = 1;
      for (int32_t h : hits) mx = std::max(mx, h);
      std::ofstream ofs(out_path, std::ios::binary);
      ofs << "P5\n" << cam.OUTW << " " << cam.OUTH << "\n255\n";
      for (int32_t h : hits) ofs.put((char)(unsigned char)(h * 255 / mx));
      std::printf("hits: max %d, written to %s\n", mx, out_path.c_str());
      return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int pass = 0;
    cam.display_only = display;
    if (progress)
      cam.render(scene, spp, [&]() { std::printf("pass %d/%d\n", ++pass, spp); });
    else
      cam.render(scene, spp);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double rays = (double)(cam.counters.rays_closest + cam.counters.rays_shadow);
    std::printf("%.3f s, %.2f M samples/s, %.2f Mrays/s\n", secs, cam.counters.paths / secs * 1e-6, rays / secs * 1e-6);
    if (display) write_display_ppm(cam, out_path);
    else cam.save(out_path);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}

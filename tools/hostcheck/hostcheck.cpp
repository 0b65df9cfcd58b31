// Robustness run of the host-side parsers under AddressSanitizer / UBSan (host code only):
// mutated PNG / PPM / OBJ / scene-cache files and mutated scene descriptions must be rejected
// or accepted, never crash.  build + run: make -C tools/hostcheck
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../pathtracer_amd/csrc/host/image.h"
#include "../../pathtracer_amd/csrc/host/object.h"
#include "../../pathtracer_amd/csrc/host/scene.h"
#include "../../pathtracer_amd/csrc/pt_flatten.h"

using namespace pt;

static std::vector<uint8_t> slurp(const std::string &p) {
  std::ifstream in(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::vector<uint8_t> &b) {
  std::ofstream out(p, std::ios::binary);
  out.write(reinterpret_cast<const char *>(b.data()), (std::streamsize)b.size());
}

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: hostcheck <dir with seed files> <iterations>\n");
    return 2;
  }
  const std::string dir = argv[1];
  const int iters = std::atoi(argv[2]);
  std::mt19937 rng(12345);
  const char *seeds[] = {"seed.png", "seed_pal.png", "seed.ppm", "seed.obj", "seed.ptscene"};
  size_t accepted = 0, rejected = 0;
  for (const char *name : seeds) {
    const std::vector<uint8_t> orig = slurp(dir + "/" + name);
    if (orig.empty()) {
      std::fprintf(stderr, "missing seed %s\n", name);
      return 2;
    }
    const std::string ext = std::string(name).substr(std::string(name).find('.'));
    const std::string tmp = dir + "/mut" + ext;
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> b = orig;
      const int kind = (int)(rng() % 4);
      if (kind == 0) {  // flip a few bytes
        for (int k = 0, n = 1 + (int)(rng() % 4); k < n; ++k) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8));
      } else if (kind == 1) {  // truncate
        b.resize(rng() % b.size());
      } else if (kind == 2) {  // overwrite a run with random bytes
        size_t at = rng() % b.size(), len = 1 + rng() % 16;
        for (size_t k = at; k < b.size() && k < at + len; ++k) b[k] = (uint8_t)rng();
      } else {  // duplicate a slice in the middle
        size_t at = rng() % b.size(), len = 1 + rng() % 64;
        std::vector<uint8_t> ins(b.begin() + (long)at, b.begin() + (long)std::min(b.size(), at + len));
        b.insert(b.begin() + (long)at, ins.begin(), ins.end());
      }
      spit(tmp, b);
      std::string err;
      bool ok = false;
      if (ext == ".png" || ext == ".ppm") {
        Image img;
        ok = load_image(tmp, img, err);
        if (ok && img.rgb.size() != (size_t)img.width * img.height * 3) return 1;
      } else if (ext == ".obj") {
        ParsedScene ps;
        ok = parseobj(tmp, ps, err);
        if (ok) {
          Scene sc(std::move(ps));
          ptl::FlatScene flat;
          std::string e2;
          (void)ptl::flatten_scene(sc.desc(), flat, e2);
        }
      } else {
        Scene sc;
        ok = Scene::load_cache(tmp, sc, &err);
        if (ok) {  // the checksum makes this rare; the tree check is the flattener's
          ptl::FlatScene flat;
          std::string e2;
          (void)ptl::flatten_scene(sc.desc(), flat, e2);
        }
      }
      (ok ? accepted : rejected)++;
    }
  }
  // scene descriptions with damaged trees / permutations / indices straight into the flattener
  {
    ParsedScene ps;
    std::string err;
    if (!parseobj(dir + "/seed.obj", ps, err)) return 2;
    Scene sc(std::move(ps));
    const PtSceneDesc &d0 = sc.desc();
    for (int it = 0; it < iters * 4; ++it) {
      std::vector<PtNode> tl(d0.tlas_nodes, d0.tlas_nodes + d0.n_tlas_nodes);
      std::vector<PtNode> bl(d0.blas_nodes, d0.blas_nodes + d0.blas_node_offset[d0.n_meshes]);
      std::vector<uint32_t> to(d0.tlas_order, d0.tlas_order + d0.n_meshes), bo(d0.blas_order, d0.blas_order + d0.n_tris);
      std::vector<uint32_t> mm(d0.mesh_material, d0.mesh_material + d0.n_meshes);
      std::vector<int32_t> mt(d0.material_texture ? d0.material_texture : nullptr,
                              d0.material_texture ? d0.material_texture + d0.n_materials : nullptr);
      PtSceneDesc d = d0;
      d.tlas_nodes = tl.data(); d.blas_nodes = bl.data(); d.tlas_order = to.data(); d.blas_order = bo.data();
      d.mesh_material = mm.data();
      if (!mt.empty()) d.material_texture = mt.data();
      switch (rng() % 6) {
        case 0: bl[rng() % bl.size()].index_or_first = (uint32_t)rng(); break;
        case 1: bl[rng() % bl.size()].count = (uint32_t)(rng() % 20); break;
        case 2: tl[rng() % tl.size()].index_or_first = (uint32_t)(rng() % 50); break;
        case 3: bo[rng() % bo.size()] = (uint32_t)rng(); break;
        case 4: mm[rng() % mm.size()] = (uint32_t)(rng() % 9); break;
        default: if (!mt.empty()) mt[rng() % mt.size()] = (int32_t)(rng() % 7) - 2; break;
      }
      ptl::FlatScene flat;
      std::string e2;
      (void)ptl::flatten_scene(d, flat, e2);
    }
  }
  std::printf("hostcheck: %zu mutated files accepted, %zu rejected, no crash\n", accepted, rejected);
  return 0;
}

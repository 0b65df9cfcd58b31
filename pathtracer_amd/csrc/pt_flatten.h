// pt_flatten.h — host-side conversion of a PtSceneDesc into the device layout.
#ifndef PT_FLATTEN_H
#define PT_FLATTEN_H

#include <string>
#include <vector>

#include "../../include/ptgpu.h"
#include "pt_layout.h"

namespace ptl {

struct FlatScene {
  std::vector<WideNode> nodes;       // node 0 is the super-root
  std::vector<TriRec> tris;          // in tie-rank order, plus one degenerate triangle at the end
  std::vector<uint8_t> tri_levels;   // per triangle record: internal nodes of the reference between the root and its leaf
  std::vector<MatRec> mats;
  std::vector<TexRec> textures;      // empty when no material has a diffuse map
  std::vector<uint8_t> texels;       // RGB bytes of all textures
  std::vector<float> tri_uv;         // 6 floats per triangle record (u,v of vt0, vt1, vt2); empty without textures
  std::vector<LightRec> lights;
  std::vector<LightTri> light_tris;
  uint32_t max_depth = 0;            // stack entries the deepest descent needs
  uint32_t n_meshes = 0;
  float max_abs_coord = 0;           // largest |coordinate| of any vertex
};

// Returns PT_OK or a negative PtStatus with `err` filled in.
int flatten_scene(const PtSceneDesc &d, FlatScene &out, std::string &err);

}  // namespace ptl
#endif

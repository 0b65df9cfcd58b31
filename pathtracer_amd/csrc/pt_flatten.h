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
  // The fast traversal's own tree over the same triangles (see pt_fasttree.cpp)
  std::vector<Quad4Node> nodes4;     // node 0 is the root
  uint32_t n_ill = 0;                // ill-conditioned triangles in the scene
  std::vector<TriRec4> tris4;        // triangle records in that tree's leaf order, plus the degenerate one at the end
  std::vector<uint32_t> tri_id4;     // per record of tris4: global triangle id in file order (0xffffffff: the degenerate one)
  std::vector<GateBox> gates;        // boxes of the reference's internal nodes
  std::vector<float> tri_uv4;        // texture coordinates in tris4 order; empty without textures
  std::vector<Oct8Node> nodes8;      // the same tree collapsed to eight children per node, boxes on a grid; empty when the
                                     // scene cannot use it (an ill-conditioned triangle or a box that is not finite)
  uint32_t max_depth8 = 0;           // stack entries the deepest descent of that one can need
  std::vector<uint32_t> tri_gate;    // per record of tris: the same gate index (input of the fast-tree builder)
  uint32_t max_depth4 = 0;           // stack entries the deepest descent of the fast tree can need
  float margin_rel = 0;              // boxes of the fast tree are widened by margin_rel * max_abs_coord + 2e-5
  uint32_t max_depth = 0;            // stack entries the deepest descent needs
  uint32_t n_meshes = 0;
  float max_abs_coord = 0;           // largest |coordinate| of any vertex
  uint32_t n_unnested = 0;           // internal boxes and triangle vertices found outside the internal box above them
};

// Builds nodes4 / tris4 / gate4 / tri_uv4 from tris, tri_gate and the description's vertices.
// A triangle (nine floats: v0, v1, v2) whose hit test can report hits away from the triangle itself: the
// reference's test (object.cpp:406-429) divides by e1 . (d x e2); when e1 and e2 are parallel, or nearly (the
// sine of their angle below 2^-9: from there on the error of the computed hit, about 2^-24 of its distance
// divided by that sine, outgrows the margin the fast tree's leaf boxes carry), or not finite, that determinant is rounding noise for every ray, and u, v, t
// come out anywhere, inside [0, 1] often enough to count as a hit far from the triangle's bounds (measured: a
// zero-area triangle "hit" 12 units away from itself).  The reference reports such hits.  A traversal that
// prunes by the triangle's bounds or by distance would lose them, so the fast tree bounds these triangles by
// their gate alone and never skips them for being far (REF_NOCULL), and the pruning walk of the
// reference-shaped tree is not used on a scene that has any.
bool triangle_is_ill_conditioned(const float *p9);

void build_fast_tree(const PtSceneDesc &d, FlatScene &out);

// Verifies the invariants of the fast tree (used by pt_scene_check).
int check_fast_tree(const PtSceneDesc &d, const FlatScene &f, std::string &err);
// The same for its eight-child form (nothing to do when the scene has none).
int check_oct_tree(const PtSceneDesc &d, const FlatScene &f, std::string &err);
// A child plane of an eight-child node as the kernels see it: base + index * 2^(e - 127), in double (exact).
double oct_plane(const Oct8Node &n, int axis, uint32_t index);

// Returns PT_OK or a negative PtStatus with `err` filled in.
int flatten_scene(const PtSceneDesc &d, FlatScene &out, std::string &err);

}  // namespace ptl
#endif

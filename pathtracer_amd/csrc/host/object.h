// object.h — host-side scene objects: the drop-in surface of the reference's `object`
// module (object.cpp) minus everything the GPU now does.  Meshes are index-based
// (no per-triangle heap objects); the BVH builder reproduces BVH::subdivide
// (object.cpp:313-361) node for node because the device consumes this tree as is.
#ifndef PT_HOST_OBJECT_H
#define PT_HOST_OBJECT_H

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/ptgpu.h"
#include "image.h"
#include "math.h"

namespace pt {

struct Material {  // object.cpp:144-156; only the fields the integrator reads
  vec diffuse = vec(.5f, .5f, .5f);
  vec emission = vec(0, 0, 0);
  std::string diffuse_map;  // path named by map_Kd
  int texture = -1;         // index into ParsedScene::textures / Scene::textures; -1 = diffuse_map == nullptr
};

struct AABB {  // object.cpp:160-196
  vec min, max;
};

struct HitData {  // object.cpp:199-204; `obj` becomes a (mesh, triangle) pair
  scalar t = INF;
  int mesh = -1;       // index into Scene::meshes, -1 on a miss
  int triangle = -1;   // index into Polygon::triangles
  vec barycentric;
};

// Flat BVH over `n` items given by their boxes.  Node layout is PtNode, which is
// the reference's BVH::Node (object.cpp:228-238).
class BVH {
 public:
  static constexpr unsigned max_leaf_size = 8;  // object.cpp:226
  std::vector<PtNode> nodes;
  std::vector<uint32_t> objects;  // item ids, permuted in place by the build
  BVH() = default;
  explicit BVH(const std::vector<AABB> &item_boxes);
  AABB bbox() const;  // object.cpp:287-291
};

// One OBJ `o` group (object.cpp:451-527).
class Polygon {
 public:
  std::vector<vec> vertices;
  std::vector<std::array<uint32_t, 3>> triangles;  // vertex ids local to this mesh, creation order
  std::vector<vec> vtextures;                       // object.cpp:453
  std::vector<std::array<uint32_t, 3>> tri_vt;      // Triangle::vt per triangle; 0xffffffff where a corner has none
  int material = 0;                                 // index into the owning material table
  double area_ = 0;                                 // sum of XY-projected triangle areas, object.cpp:493-495
  BVH bvh;

  // Triangulated faces -> areas -> private BVH (object.cpp:460-503).  Returns false and
  // fills `err` when the mesh has no triangle or an index is out of range.
  bool init(std::string *err = nullptr);
  AABB bbox() const { return bvh.bbox(); }          // object.cpp:507
  double area() const { return area_; }             // object.cpp:511
  vec v0(size_t t) const { return vertices[triangles[t][0]]; }
  vec v1(size_t t) const { return vertices[triangles[t][1]]; }
  vec v2(size_t t) const { return vertices[triangles[t][2]]; }
  vec normal(size_t t) const { return (v1(t) - v0(t)).cross(v2(t) - v0(t)); }  // object.cpp:393
  double triangle_area(size_t t) const;                                         // object.cpp:395-404
  AABB triangle_bbox(size_t t) const;                                           // object.cpp:377-382
};

struct ParsedScene {
  std::vector<Polygon> meshes;
  std::vector<Material> materials;  // [0] is the default material of Object::Object (object.cpp:209)
  std::vector<Image> textures;      // decoded diffuse maps
};

// parseobj (object.cpp:595-710) for the dialect its regular expressions accept.
// Returns false and fills `err` where the reference would crash (vertex before `o`,
// unknown usemtl, a number std::stof rejects, a face with fewer than 3 corners).
bool parseobj(const std::string &filename, ParsedScene &out, std::string &err);

extern bool verbose;  // print the per-mesh counts and timings the reference prints

}  // namespace pt
#endif

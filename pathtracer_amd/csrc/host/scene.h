// scene.h — host-side Scene: the reference's `scene` module (scene.cpp) with the ray
// queries answered by the GPU library.
#ifndef PT_HOST_SCENE_H
#define PT_HOST_SCENE_H

#include <string>
#include <vector>

#include "object.h"

namespace pt {

class Scene {
 public:
  std::vector<Polygon> meshes;      // creation ("file") order
  std::vector<Material> materials;
  std::vector<Image> textures;      // decoded diffuse maps; Material::texture indexes this
  BVH objects;                      // top-level tree over the meshes (scene.cpp:13); objects.objects
                                    // holds mesh ids in the order the build left them

  Scene() = default;
  // Scene::Scene (scene.cpp:15): takes the parsed meshes and builds the top-level tree.
  explicit Scene(ParsedScene &&parsed);
  Scene(const Scene &) = delete;
  Scene &operator=(const Scene &) = delete;
  Scene(Scene &&o) noexcept;
  Scene &operator=(Scene &&o) noexcept;
  ~Scene();

  // Scene::castray (scene.cpp:18-20) for one ray, answered by the device.  Uploads the
  // scene on first use.  Throws std::runtime_error when the device call fails.
  HitData castray(vec origin, vec direction) const;
  // The same for n rays at once (structure-of-arrays results).
  void castrays(size_t n, const vec *origin, const vec *direction, HitData *out, int mode = PT_TRACE_FAST) const;

  // Scene::lights (scene.cpp:22-30): ids of emissive meshes in top-level object order.
  std::vector<int> lights() const;

  size_t triangle_count() const;
  // Polygon::area of mesh `mesh` (object.cpp:493-495,511); from the stored vertices for a cached scene.
  double mesh_area(size_t mesh) const;

  // Binary cache of the *built* scene (both trees, their permutations, vertices, materials):
  // loading it skips OBJ parsing and BVH construction, which dominate start-up for scenes of
  // millions of triangles, and uploads exactly what a fresh build would upload.  A loaded
  // scene has no `meshes` (vertex sharing is not stored); ray queries, lights(),
  // triangle_count() and rendering work as usual.  Both return false and fill `err`.
  bool save_cache(const std::string &path, std::string *err = nullptr) const;
  static bool load_cache(const std::string &path, Scene &out, std::string *err = nullptr);
  bool from_cache() const { return cached_; }
  // Fills a PtSceneDesc whose pointers refer to storage owned by this Scene.
  const PtSceneDesc &desc() const;
  // Device copy (created on first use on `device`; copies on several devices are kept side by side).
  PtScene *device_scene(int device = 0) const;

 private:
  void build_desc() const;
  void point_desc() const;  // desc_ <- the storage vectors below
  bool cached_ = false;  // desc_ storage was read from a cache file, not built from `meshes`
  mutable bool desc_ready_ = false;
  mutable PtSceneDesc desc_{};
  mutable std::vector<uint32_t> blas_node_offset_, tri_offset_, blas_order_, mesh_material_;
  mutable std::vector<PtNode> blas_nodes_;
  mutable std::vector<float> tri_verts_;
  mutable std::vector<PtMaterial> mats_;
  mutable std::vector<float> tri_uv_;            // only when a texture exists
  mutable std::vector<int32_t> material_texture_;
  mutable std::vector<PtTexture> tex_desc_;
  mutable std::vector<std::pair<int, PtScene *>> dev_;  // (device ordinal, copy); the first one serves castrays()
};

}  // namespace pt
#endif

// scene.cpp — host-side Scene.
#include "scene.h"

#include <stdexcept>

namespace pt {

Scene::Scene(ParsedScene &&parsed) : meshes(std::move(parsed.meshes)), materials(std::move(parsed.materials)) {
  std::vector<AABB> boxes(meshes.size());
  for (size_t i = 0; i < meshes.size(); ++i) boxes[i] = meshes[i].bbox();
  objects = BVH(boxes);
}

Scene::Scene(Scene &&o) noexcept { *this = std::move(o); }

Scene &Scene::operator=(Scene &&o) noexcept {
  if (this != &o) {
    if (dev_) pt_scene_free(dev_);
    meshes = std::move(o.meshes);
    materials = std::move(o.materials);
    objects = std::move(o.objects);
    desc_ready_ = false;  // pointers in desc_ refer to the vectors below; rebuild lazily
    dev_ = o.dev_;
    dev_ordinal_ = o.dev_ordinal_;
    o.dev_ = nullptr;
    o.desc_ready_ = false;
  }
  return *this;
}

Scene::~Scene() {
  if (dev_) pt_scene_free(dev_);
}

std::vector<int> Scene::lights() const {
  std::vector<int> res;
  for (uint32_t id : objects.objects)
    if (materials[meshes[id].material].emission != vec(0, 0, 0)) res.push_back((int)id);
  return res;
}

size_t Scene::triangle_count() const {
  size_t n = 0;
  for (const Polygon &m : meshes) n += m.triangles.size();
  return n;
}

void Scene::build_desc() const {
  const size_t nm = meshes.size();
  blas_node_offset_.assign(nm + 1, 0);
  tri_offset_.assign(nm + 1, 0);
  mesh_material_.resize(nm);
  for (size_t i = 0; i < nm; ++i) {
    blas_node_offset_[i + 1] = blas_node_offset_[i] + (uint32_t)meshes[i].bvh.nodes.size();
    tri_offset_[i + 1] = tri_offset_[i] + (uint32_t)meshes[i].triangles.size();
    mesh_material_[i] = (uint32_t)meshes[i].material;
  }
  blas_nodes_.clear();
  blas_nodes_.reserve(blas_node_offset_[nm]);
  blas_order_.clear();
  blas_order_.reserve(tri_offset_[nm]);
  tri_verts_.clear();
  tri_verts_.reserve((size_t)tri_offset_[nm] * 9);
  for (const Polygon &m : meshes) {
    blas_nodes_.insert(blas_nodes_.end(), m.bvh.nodes.begin(), m.bvh.nodes.end());
    blas_order_.insert(blas_order_.end(), m.bvh.objects.begin(), m.bvh.objects.end());
    for (size_t t = 0; t < m.triangles.size(); ++t)
      for (int c = 0; c < 3; ++c) {
        const vec &p = m.vertices[m.triangles[t][c]];
        tri_verts_.push_back(p.x);
        tri_verts_.push_back(p.y);
        tri_verts_.push_back(p.z);
      }
  }
  mats_.resize(materials.size());
  for (size_t i = 0; i < materials.size(); ++i) {
    mats_[i].kd[0] = materials[i].diffuse.x; mats_[i].kd[1] = materials[i].diffuse.y; mats_[i].kd[2] = materials[i].diffuse.z;
    mats_[i].ke[0] = materials[i].emission.x; mats_[i].ke[1] = materials[i].emission.y; mats_[i].ke[2] = materials[i].emission.z;
  }
  desc_.n_meshes = (uint32_t)nm;
  desc_.n_tris = tri_offset_[nm];
  desc_.n_materials = (uint32_t)mats_.size();
  desc_.n_tlas_nodes = (uint32_t)objects.nodes.size();
  desc_.tlas_nodes = objects.nodes.data();
  desc_.tlas_order = objects.objects.data();
  desc_.blas_node_offset = blas_node_offset_.data();
  desc_.blas_nodes = blas_nodes_.data();
  desc_.tri_offset = tri_offset_.data();
  desc_.blas_order = blas_order_.data();
  desc_.tri_verts = tri_verts_.data();
  desc_.mesh_material = mesh_material_.data();
  desc_.materials = mats_.data();
  desc_ready_ = true;
}

const PtSceneDesc &Scene::desc() const {
  if (!desc_ready_) build_desc();
  return desc_;
}

PtScene *Scene::device_scene(int device) const {
  if (dev_ && dev_ordinal_ == device) return dev_;
  if (dev_) {
    pt_scene_free(dev_);
    dev_ = nullptr;
  }
  PtScene *s = nullptr;
  if (pt_scene_upload(&desc(), device, &s) != PT_OK) throw std::runtime_error(std::string("pt_scene_upload: ") + pt_last_error());
  dev_ = s;
  dev_ordinal_ = device;
  return dev_;
}

void Scene::castrays(size_t n, const vec *origin, const vec *direction, HitData *out, int mode) const {
  if (n == 0) return;
  PtScene *s = device_scene(dev_ordinal_ < 0 ? 0 : dev_ordinal_);
  std::vector<float> in(6 * n), t(n), u(n), v(n);
  std::vector<int32_t> tri(n);
  for (size_t i = 0; i < n; ++i) {
    in[i] = origin[i].x; in[n + i] = origin[i].y; in[2 * n + i] = origin[i].z;
    in[3 * n + i] = direction[i].x; in[4 * n + i] = direction[i].y; in[5 * n + i] = direction[i].z;
  }
  if (pt_trace_closest(s, n, &in[0], &in[n], &in[2 * n], &in[3 * n], &in[4 * n], &in[5 * n], t.data(), tri.data(),
                       u.data(), v.data(), mode) != PT_OK)
    throw std::runtime_error(std::string("pt_trace_closest: ") + pt_last_error());
  desc();
  for (size_t i = 0; i < n; ++i) {
    HitData h;
    h.t = t[i];
    if (tri[i] >= 0) {
      // global triangle id -> (mesh, local triangle)
      size_t lo = 0, hi = meshes.size();
      while (hi - lo > 1) {
        size_t mid = (lo + hi) / 2;
        if (tri_offset_[mid] <= (uint32_t)tri[i]) lo = mid; else hi = mid;
      }
      h.mesh = (int)lo;
      h.triangle = (int)((uint32_t)tri[i] - tri_offset_[lo]);
      h.barycentric = vec(u[i], v[i], 1 - u[i] - v[i]);  // object.cpp:428
    }
    out[i] = h;
  }
}

HitData Scene::castray(vec origin, vec direction) const {
  HitData h;
  castrays(1, &origin, &direction, &h, PT_TRACE_EXACT);
  return h;
}

}  // namespace pt

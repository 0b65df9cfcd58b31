// scene.cpp — host-side Scene.
#include "scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pt {

Scene::Scene(ParsedScene &&parsed)
    : meshes(std::move(parsed.meshes)), materials(std::move(parsed.materials)), textures(std::move(parsed.textures)) {
  std::vector<AABB> boxes(meshes.size());
  for (size_t i = 0; i < meshes.size(); ++i) boxes[i] = meshes[i].bbox();
  objects = BVH(boxes);
}

Scene::Scene(Scene &&o) noexcept { *this = std::move(o); }

Scene &Scene::operator=(Scene &&o) noexcept {
  if (this != &o) {
    for (auto &d : dev_) pt_scene_free(d.second);
    dev_.clear();
    meshes = std::move(o.meshes);
    materials = std::move(o.materials);
    textures = std::move(o.textures);
    objects = std::move(o.objects);
    desc_ready_ = false;  // pointers in desc_ refer to the vectors below; rebuild lazily
    cached_ = o.cached_;
    if (cached_) {  // no meshes to rebuild from: take the arrays themselves
      blas_node_offset_ = std::move(o.blas_node_offset_);
      tri_offset_ = std::move(o.tri_offset_);
      blas_order_ = std::move(o.blas_order_);
      mesh_material_ = std::move(o.mesh_material_);
      blas_nodes_ = std::move(o.blas_nodes_);
      tri_verts_ = std::move(o.tri_verts_);
      mats_ = std::move(o.mats_);
      tri_uv_ = std::move(o.tri_uv_);
      material_texture_ = std::move(o.material_texture_);
      point_desc();
    }
    dev_ = std::move(o.dev_);
    o.dev_.clear();
    o.desc_ready_ = false;
  }
  return *this;
}

Scene::~Scene() {
  for (auto &d : dev_) pt_scene_free(d.second);
}

std::vector<int> Scene::lights() const {
  std::vector<int> res;
  for (uint32_t id : objects.objects) {
    const int mat = cached_ ? (int)mesh_material_[id] : meshes[id].material;
    if (materials[mat].emission != vec(0, 0, 0)) res.push_back((int)id);
  }
  return res;
}

size_t Scene::triangle_count() const {
  if (cached_) return tri_offset_.empty() ? 0 : tri_offset_.back();
  size_t n = 0;
  for (const Polygon &m : meshes) n += m.triangles.size();
  return n;
}

double Scene::mesh_area(size_t mesh) const {
  if (!cached_) return meshes[mesh].area();
  double area = 0;
  for (uint32_t t = tri_offset_[mesh]; t < tri_offset_[mesh + 1]; ++t) {
    const float *a = &tri_verts_[(size_t)t * 9], *b = a + 3, *c = a + 6;
    area += (double)(std::abs(a[0] * (b[1] - c[1]) + b[0] * (c[1] - a[1]) + c[0] * (a[1] - b[1])) / 2);  // as Polygon::triangle_area
  }
  return area;
}

void Scene::point_desc() const {
  const size_t nm = mesh_material_.size();
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
  tex_desc_.resize(textures.size());
  for (size_t i = 0; i < textures.size(); ++i) {
    tex_desc_[i].width = textures[i].width;
    tex_desc_[i].height = textures[i].height;
    tex_desc_[i].rgb = textures[i].rgb.data();
  }
  desc_.n_textures = (uint32_t)textures.size();
  desc_.textures = textures.empty() ? nullptr : tex_desc_.data();
  desc_.material_texture = textures.empty() ? nullptr : material_texture_.data();
  desc_.tri_uv = textures.empty() ? nullptr : tri_uv_.data();
  desc_ready_ = true;
}

void Scene::build_desc() const {
  if (cached_) {
    point_desc();
    return;
  }
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
  tri_uv_.clear();
  material_texture_.clear();
  if (!textures.empty()) {
    material_texture_.resize(materials.size());
    for (size_t i = 0; i < materials.size(); ++i) material_texture_[i] = materials[i].texture;
    tri_uv_.reserve((size_t)tri_offset_[nm] * 6);
    for (const Polygon &m : meshes)
      for (size_t t = 0; t < m.triangles.size(); ++t)
        for (int c = 0; c < 3; ++c) {
          const uint32_t id = m.tri_vt.size() == m.triangles.size() ? m.tri_vt[t][c] : 0xffffffffu;
          const bool has = id < m.vtextures.size();  // untextured objects may lack coordinates
          tri_uv_.push_back(has ? m.vtextures[id].x : 0.0f);
          tri_uv_.push_back(has ? m.vtextures[id].y : 0.0f);
        }
  }
  point_desc();
}

const PtSceneDesc &Scene::desc() const {
  if (!desc_ready_) build_desc();
  return desc_;
}

PtScene *Scene::device_scene(int device) const {
  for (auto &d : dev_)
    if (d.first == device) return d.second;
  PtScene *s = nullptr;
  if (pt_scene_upload(&desc(), device, &s) != PT_OK) throw std::runtime_error(std::string("pt_scene_upload: ") + pt_last_error());
  dev_.emplace_back(device, s);
  return s;
}

void Scene::castrays(size_t n, const vec *origin, const vec *direction, HitData *out, int mode) const {
  if (n == 0) return;
  PtScene *s = device_scene(dev_.empty() ? 0 : dev_.front().first);
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
      size_t lo = 0, hi = mesh_material_.size();
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

// ---------------------------------------------------------------- cache file
// Layout (little endian): magic "PTSCENE2", seven uint32 counts (meshes, triangles, materials,
// top-level nodes, mesh nodes, sizeof(PtNode), sizeof(PtMaterial)), a 64-bit FNV-1a hash of the
// payload, then the payload: the arrays of PtSceneDesc in declaration order; after them a
// uint32 texture count and, when it is not 0, width and height of every texture, the
// material_texture and tri_uv arrays and the texels.
namespace {

const char kMagic[8] = {'P', 'T', 'S', 'C', 'E', 'N', 'E', '2'};

struct Fnv {
  uint64_t h = 1469598103934665603ull;
  void add(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  }
};

bool set_err(std::string *err, const std::string &msg) {
  if (err) *err = msg;
  return false;
}

}  // namespace

bool Scene::save_cache(const std::string &path, std::string *err) const {
  const PtSceneDesc &d = desc();
  const uint32_t n_blas = d.blas_node_offset[d.n_meshes];
  const uint32_t head[7] = {d.n_meshes, d.n_tris, d.n_materials, d.n_tlas_nodes, n_blas, (uint32_t)sizeof(PtNode),
                            (uint32_t)sizeof(PtMaterial)};
  struct Part { const void *p; size_t n; };
  const Part parts[] = {
      {d.tlas_nodes, (size_t)d.n_tlas_nodes * sizeof(PtNode)},
      {d.tlas_order, (size_t)d.n_meshes * 4},
      {d.blas_node_offset, ((size_t)d.n_meshes + 1) * 4},
      {d.blas_nodes, (size_t)n_blas * sizeof(PtNode)},
      {d.tri_offset, ((size_t)d.n_meshes + 1) * 4},
      {d.blas_order, (size_t)d.n_tris * 4},
      {d.tri_verts, (size_t)d.n_tris * 36},
      {d.mesh_material, (size_t)d.n_meshes * 4},
      {d.materials, (size_t)d.n_materials * sizeof(PtMaterial)},
  };
  std::vector<Part> all(parts, parts + 9);
  const uint32_t n_tex = d.n_textures;
  std::vector<uint32_t> dims;
  all.push_back({&n_tex, 4});
  if (n_tex) {
    for (uint32_t t = 0; t < n_tex; ++t) {
      dims.push_back(d.textures[t].width);
      dims.push_back(d.textures[t].height);
    }
    all.push_back({dims.data(), dims.size() * 4});
    all.push_back({d.material_texture, (size_t)d.n_materials * 4});
    all.push_back({d.tri_uv, (size_t)d.n_tris * 24});
    for (uint32_t t = 0; t < n_tex; ++t) all.push_back({d.textures[t].rgb, (size_t)d.textures[t].width * d.textures[t].height * 3});
  }
  Fnv f;
  for (const Part &q : all) f.add(q.p, q.n);
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) return set_err(err, "cannot open " + path + " for writing");
  bool ok = std::fwrite(kMagic, 1, 8, fp) == 8 && std::fwrite(head, 4, 7, fp) == 7 && std::fwrite(&f.h, 8, 1, fp) == 1;
  for (const Part &q : all) ok = ok && (q.n == 0 || std::fwrite(q.p, 1, q.n, fp) == q.n);
  ok = (std::fclose(fp) == 0) && ok;
  if (!ok) return set_err(err, "short write to " + path);
  return true;
}

bool Scene::load_cache(const std::string &path, Scene &out, std::string *err) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return set_err(err, "cannot open " + path);
  char magic[8];
  uint32_t head[7];
  uint64_t want = 0;
  bool ok = std::fread(magic, 1, 8, fp) == 8 && std::fread(head, 4, 7, fp) == 7 && std::fread(&want, 8, 1, fp) == 1;
  if (!ok || std::memcmp(magic, kMagic, 8) != 0 || head[5] != sizeof(PtNode) || head[6] != sizeof(PtMaterial)) {
    std::fclose(fp);
    return set_err(err, path + " is not a scene cache of this version");
  }
  const uint32_t nm = head[0], nt = head[1], nmat = head[2], ntl = head[3], nbl = head[4];
  // the counts must account for the file's size exactly before anything is allocated from them
  const uint64_t payload = (uint64_t)ntl * sizeof(PtNode) + (uint64_t)nm * 4 + ((uint64_t)nm + 1) * 4 +
                           (uint64_t)nbl * sizeof(PtNode) + ((uint64_t)nm + 1) * 4 + (uint64_t)nt * 4 +
                           (uint64_t)nt * 36 + (uint64_t)nm * 4 + (uint64_t)nmat * sizeof(PtMaterial);
  std::fseek(fp, 0, SEEK_END);
  const long size = std::ftell(fp);
  std::fseek(fp, 8 + 28 + 8, SEEK_SET);
  if (size < 0 || (uint64_t)size < 8 + 28 + 8 + payload + 4 || nm == 0) {
    std::fclose(fp);
    return set_err(err, path + ": size does not match its header");
  }
  Scene s;
  s.cached_ = true;
  s.objects.nodes.resize(ntl);
  s.objects.objects.resize(nm);
  s.blas_node_offset_.resize((size_t)nm + 1);
  s.blas_nodes_.resize(nbl);
  s.tri_offset_.resize((size_t)nm + 1);
  s.blas_order_.resize(nt);
  s.tri_verts_.resize((size_t)nt * 9);
  s.mesh_material_.resize(nm);
  s.mats_.resize(nmat);
  struct Part { void *p; size_t n; };
  const Part parts[] = {
      {s.objects.nodes.data(), (size_t)ntl * sizeof(PtNode)}, {s.objects.objects.data(), (size_t)nm * 4},
      {s.blas_node_offset_.data(), ((size_t)nm + 1) * 4},     {s.blas_nodes_.data(), (size_t)nbl * sizeof(PtNode)},
      {s.tri_offset_.data(), ((size_t)nm + 1) * 4},           {s.blas_order_.data(), (size_t)nt * 4},
      {s.tri_verts_.data(), (size_t)nt * 36},                 {s.mesh_material_.data(), (size_t)nm * 4},
      {s.mats_.data(), (size_t)nmat * sizeof(PtMaterial)},
  };
  Fnv f;
  for (const Part &q : parts) {
    ok = ok && (q.n == 0 || std::fread(q.p, 1, q.n, fp) == q.n);
    if (ok) f.add(q.p, q.n);
  }
  // texture section: every count is checked against what is left of the file before it is used
  uint64_t left = (uint64_t)size - (8 + 28 + 8 + payload);
  uint32_t n_tex = 0;
  ok = ok && std::fread(&n_tex, 4, 1, fp) == 1;
  if (ok) f.add(&n_tex, 4);
  left -= 4;
  if (ok && n_tex) {
    std::vector<uint32_t> dims((size_t)2 * (n_tex <= left / 8 ? n_tex : 0));
    ok = !dims.empty() && std::fread(dims.data(), 4, dims.size(), fp) == dims.size();
    uint64_t need = (uint64_t)nmat * 4 + (uint64_t)nt * 24;
    if (ok) {
      f.add(dims.data(), dims.size() * 4);
      left -= dims.size() * 4;
      for (uint32_t t = 0; t < n_tex && ok; ++t) {
        ok = dims[2 * t] != 0 && dims[2 * t + 1] != 0 && (uint64_t)dims[2 * t] * dims[2 * t + 1] <= (1ull << 28);
        need += (uint64_t)dims[2 * t] * dims[2 * t + 1] * 3;
      }
    }
    ok = ok && need == left;
    if (ok) {
      s.material_texture_.resize(nmat);
      s.tri_uv_.resize((size_t)nt * 6);
      ok = std::fread(s.material_texture_.data(), 4, nmat, fp) == nmat && std::fread(s.tri_uv_.data(), 4, (size_t)nt * 6, fp) == (size_t)nt * 6;
      if (ok) {
        f.add(s.material_texture_.data(), (size_t)nmat * 4);
        f.add(s.tri_uv_.data(), (size_t)nt * 24);
      }
      s.textures.resize(n_tex);
      for (uint32_t t = 0; t < n_tex && ok; ++t) {
        Image &img = s.textures[t];
        img.width = dims[2 * t];
        img.height = dims[2 * t + 1];
        img.rgb.resize((size_t)img.width * img.height * 3);
        ok = std::fread(img.rgb.data(), 1, img.rgb.size(), fp) == img.rgb.size();
        if (ok) f.add(img.rgb.data(), img.rgb.size());
      }
    }
  } else if (ok && left != 0) {
    ok = false;
  }
  std::fclose(fp);
  if (!ok) return set_err(err, path + ": short or inconsistent file");
  if (f.h != want) return set_err(err, path + ": checksum mismatch");
  if (s.tri_offset_[nm] != nt || s.blas_node_offset_[nm] != nbl) return set_err(err, path + ": offsets do not end at the totals");
  for (uint32_t i = 0; i < nm; ++i)
    if (s.mesh_material_[i] >= nmat) return set_err(err, path + ": material index out of range");
  s.materials.resize(nmat);
  for (uint32_t i = 0; i < nmat; ++i) {
    s.materials[i].diffuse = vec(s.mats_[i].kd[0], s.mats_[i].kd[1], s.mats_[i].kd[2]);
    s.materials[i].emission = vec(s.mats_[i].ke[0], s.mats_[i].ke[1], s.mats_[i].ke[2]);
    if (n_tex) {
      if (s.material_texture_[i] < -1 || s.material_texture_[i] >= (int32_t)n_tex) return set_err(err, path + ": texture index out of range");
      s.materials[i].texture = s.material_texture_[i];
    }
  }
  out = std::move(s);  // the tree structure itself is validated by pt_scene_upload
  return true;
}

}  // namespace pt

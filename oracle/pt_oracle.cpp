// pt_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by or
// called from the product path; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may use it, and there only as the checker.
//
// A CPU restatement of the render hot path of kejcao/pathtracer @ 2024-10-08,
// written from the reference's source text.  Each function names the reference
// lines it follows.  float32 throughout, one rounding per operation (build with
// -ffp-contract=off and without -march, like the reference's Release build).
//
// PARITY PINNING: the reference itself cannot be built in this environment
// without stand-in headers (it needs <format>, <print>, std::unreachable and
// the FFmpeg headers), so this restatement is pinned only by the known-answer
// values recorded from an earlier run of the reference (closest-hit t bits
// and barycentrics on examples/plane.obj, point_towards at the default camera);
// everything those values do not cover is "parity unpinned".
//
// Two things here are not in the reference and exist so that the GPU path can be
// compared bit for bit: a counter-based random generator (the reference's is an
// unseeded shared mt19937, math.cpp:13-18) and an iterative form of the
// recursive estimator (camera.cpp:63-159) that multiplies a running throughput
// instead of unwinding a recursion.  The reference forms are kept beside them.

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <numeric>
#include <random>
#include <regex>
#include <string>
#include <thread>
#include <vector>

#include "../include/ptgpu.h"

namespace orc {

static const float kInf = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------- math.cpp:20-183
struct V3 {
  float x = 0, y = 0, z = 0;
  V3() = default;
  V3(float a, float b, float c) : x(a), y(b), z(c) {}
  float at(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 add(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 sub(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 mul(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 scale(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }  // math.cpp:168-175
static inline V3 divs(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }   // math.cpp:176-183
static inline V3 neg(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // math.cpp:108
static inline V3 cross(V3 a, V3 b) {                                                 // math.cpp:124-126
  return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // math.cpp:109
static inline V3 normalize(V3 a) {                                                        // math.cpp:111-114
  float n = norm(a);
  return n == 0 ? a : V3(a.x / n, a.y / n, a.z / n);
}
// math.cpp:130-140: every product in double, one narrowing per component.
static inline V3 rotate(V3 p, float a, float b, float y) {
  double ca = std::cos((double)a), sa = std::sin((double)a);
  double cb = std::cos((double)b), sb = std::sin((double)b);
  double cy = std::cos((double)y), sy = std::sin((double)y);
  double px = p.x, py = p.y, pz = p.z;
  double rx = (cb * cy * px) + ((sa * sb * cy - ca * sy) * py) + ((ca * sb * cy + sa * sy) * pz);
  double ry = (cb * sy * px) + ((sa * sb * sy + ca * cy) * py) + ((ca * sb * sy - sa * cy) * pz);
  double rz = (-sb * px) + (sa * cb * py) + (ca * cb * pz);
  return V3((float)rx, (float)ry, (float)rz);
}

// ---------------------------------------------------------------- geometry records
struct Box {
  V3 lo, hi;
};
struct Node {  // BVH::Node, object.cpp:228-238
  uint32_t ref;
  uint32_t count;
  Box box;
};
struct Tree {
  std::vector<Node> nodes;
  std::vector<uint32_t> order;  // BVH::objects as item ids
};
struct Mat {  // object.cpp:144-156
  V3 kd = V3(.5f, .5f, .5f);
  V3 ke = V3(0, 0, 0);
  int tex = -1;  // diffuse_map, object.cpp:149: index into World::textures, -1 = nullptr
};
struct Tex {  // Image, object.cpp:31-142, after decoding to RGB24
  int w = 0, h = 0;
  std::vector<uint8_t> rgb;
};
struct Tri {
  uint32_t a, b, c;
};
struct Mesh {  // Polygon, object.cpp:451-527
  std::vector<V3> verts;
  std::vector<V3> vts;       // vtextures, object.cpp:453
  std::vector<Tri> tris;
  std::vector<Tri> tri_vt;   // Triangle::vt per triangle (empty when the faces carry no vt index)
  int mat = -1;
  double area = 0;
  Tree blas;
  uint32_t tri_base = 0;
};
struct World {
  std::vector<Mesh> meshes;
  std::vector<Mat> mats;
  std::vector<Tex> textures;
  Tree tlas;
  std::vector<uint32_t> lights;  // Scene::lights(), scene.cpp:22-30
  std::string error;
};

struct Counters {
  uint64_t rays_closest = 0, rays_shadow = 0, nodes = 0, tris = 0, paths = 0;
};
static thread_local Counters tl_cnt;

// AABB::intersect, object.cpp:179-195
static inline bool box_hit(const Box &b, V3 o, V3 d) {
  float tmin = (float)1e-5, tmax = kInf;
  for (int i = 0; i < 3; ++i) {
    float di = d.at(i);
    if (di != 0) {
      float n = 1 / di;
      float t0 = (b.lo.at(i) - o.at(i)) * n;
      float t1 = (b.hi.at(i) - o.at(i)) * n;
      if (n < 0) std::swap(t0, t1);
      if (t0 > tmin) tmin = t0;
      if (t1 < tmax) tmax = t1;
    }
  }
  return tmax >= tmin;
}

struct Hit {  // HitData, object.cpp:199-204
  float t = kInf;
  int mesh = -1;
  int tri = -1;
  float u = 0, v = 0, w = 0;
  int hits = 0;
};

// Triangle::intersect, object.cpp:406-429
static inline Hit tri_hit(const Mesh &m, int mi, int ti, V3 o, V3 d) {
  ++tl_cnt.tris;
  const Tri &T = m.tris[ti];
  V3 v0 = m.verts[T.a];
  V3 e1 = sub(m.verts[T.b], v0);
  V3 e2 = sub(m.verts[T.c], v0);
  V3 p = cross(d, e2);
  float n = dot(p, e1);
  Hit miss;
  if (n == 0) return miss;
  V3 tv = sub(o, v0);
  float u = 1 / n * dot(p, tv);
  if (u < 0 || u > 1) return miss;
  V3 q = scale(cross(tv, e1), 1 / n);
  float v = dot(q, d);
  if (v < 0 || u + v > 1) return miss;
  float t = dot(q, e2);
  if (t < 1e-6) return miss;  // compared in double, object.cpp:426
  Hit h;
  h.t = t;
  h.mesh = mi;
  h.tri = ti;
  h.u = u;
  h.v = v;
  h.w = 1 - u - v;
  return h;
}

// BVH::intersect, object.cpp:265-285, for a mesh's private tree.
static Hit blas_hit(const Mesh &m, int mi, V3 o, V3 d, uint32_t i) {
  ++tl_cnt.nodes;
  const Node &nd = m.blas.nodes[i];
  if (nd.count > 0) {
    Hit best = tri_hit(m, mi, (int)m.blas.order[nd.ref], o, d);
    for (uint32_t k = 1; k < nd.count; ++k) {
      Hit h = tri_hit(m, mi, (int)m.blas.order[nd.ref + k], o, d);
      if (h.t < best.t) best = h;  // leftmost minimum
    }
    return best;
  }
  if (!box_hit(nd.box, o, d)) return Hit();
  Hit l = blas_hit(m, mi, o, d, nd.ref);
  Hit r = blas_hit(m, mi, o, d, nd.ref + 1);
  ++l.hits;
  ++r.hits;
  return l.t < r.t ? l : r;
}

// The same routine over the top-level tree, whose leaf items are meshes
// (Polygon::intersect, object.cpp:508-510).
static Hit tlas_hit(const World &w, V3 o, V3 d, uint32_t i) {
  ++tl_cnt.nodes;
  const Node &nd = w.tlas.nodes[i];
  if (nd.count > 0) {
    uint32_t m0 = w.tlas.order[nd.ref];
    Hit best = blas_hit(w.meshes[m0], (int)m0, o, d, 0);
    for (uint32_t k = 1; k < nd.count; ++k) {
      uint32_t mk = w.tlas.order[nd.ref + k];
      Hit h = blas_hit(w.meshes[mk], (int)mk, o, d, 0);
      if (h.t < best.t) best = h;
    }
    return best;
  }
  if (!box_hit(nd.box, o, d)) return Hit();
  Hit l = tlas_hit(w, o, d, nd.ref);
  Hit r = tlas_hit(w, o, d, nd.ref + 1);
  ++l.hits;
  ++r.hits;
  return l.t < r.t ? l : r;
}

// Scene::castray, scene.cpp:18-20
static inline Hit castray(const World &w, V3 o, V3 d) { return tlas_hit(w, o, d, 0); }

// ---------------------------------------------------------------- build (object.cpp:255-263,296-361)
static inline float fmin2(float a, float b) { return b < a ? b : a; }  // std::min
static inline float fmax2(float a, float b) { return a < b ? b : a; }  // std::max

struct Builder {
  Tree &T;
  const std::vector<Box> &ib;   // item boxes (Object::bbox)
  std::vector<V3> cen;          // Object::centroid, object.cpp:219
  Builder(Tree &t, const std::vector<Box> &b) : T(t), ib(b) {
    cen.resize(b.size());
    for (size_t i = 0; i < b.size(); ++i) cen[i] = add(scale(b[i].lo, .5f), scale(b[i].hi, .5f));
  }
  void fit(uint32_t i) {  // update_box, object.cpp:296-308
    Box bx;
    bx.lo = V3(kInf, kInf, kInf);
    bx.hi = V3(-kInf, -kInf, -kInf);
    uint32_t f = T.nodes[i].ref, c = T.nodes[i].count;
    for (uint32_t k = f; k < f + c; ++k) {
      const Box &o = ib[T.order[k]];
      bx.lo = V3(fmin2(bx.lo.x, o.lo.x), fmin2(bx.lo.y, o.lo.y), fmin2(bx.lo.z, o.lo.z));
      bx.hi = V3(fmax2(bx.hi.x, o.hi.x), fmax2(bx.hi.y, o.hi.y), fmax2(bx.hi.z, o.hi.z));
    }
    T.nodes[i].box = bx;
  }
  void split(uint32_t i) {  // subdivide, object.cpp:313-361
    uint32_t count = T.nodes[i].count;
    if (count <= 8) return;
    uint32_t first = T.nodes[i].ref;
    V3 ext = sub(T.nodes[i].box.hi, T.nodes[i].box.lo);
    int axis = 0;
    if (ext.y > ext.x) axis = 1;
    if (ext.z > ext.at(axis)) axis = 2;
    float mid = T.nodes[i].box.lo.at(axis) + ext.at(axis) / 2;
    auto b = T.order.begin() + first;
    auto it = std::partition(b, b + count, [&](uint32_t id) { return cen[id].at(axis) < mid; });
    uint32_t midpoint = (uint32_t)std::distance(b, it);
    if (midpoint == 0 || midpoint >= count - 1) {
      std::nth_element(b, b + count / 2, b + count,
                       [&](uint32_t p, uint32_t q) { return cen[p].at(axis) < cen[q].at(axis); });
      midpoint = count / 2;
    }
    uint32_t lhs = (uint32_t)T.nodes.size();
    Node L{first, midpoint + 1, Box()};  // the +1 is the reference's, object.cpp:348
    Node R{first + midpoint + 1, count - (midpoint + 1), Box()};
    T.nodes.push_back(L);
    T.nodes.push_back(R);
    T.nodes[i].ref = lhs;
    T.nodes[i].count = 0;
    fit(lhs);
    fit(lhs + 1);
    split(lhs);
    split(lhs + 1);
  }
  void run() {
    T.nodes.clear();
    T.order.resize(ib.size());
    std::iota(T.order.begin(), T.order.end(), 0u);
    T.nodes.push_back(Node{0, (uint32_t)ib.size(), Box()});
    fit(0);
    split(0);
  }
};

static inline float min3(float a, float b, float c) {  // std::min({a,b,c})
  float m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  return m;
}
static inline float max3(float a, float b, float c) {  // std::max({a,b,c})
  float m = a;
  if (m < b) m = b;
  if (m < c) m = c;
  return m;
}

// Triangle::area, object.cpp:395-404 (XY projection only; float, then widened)
static inline double tri_area(const Mesh &m, const Tri &t) {
  V3 a = m.verts[t.a], b = m.verts[t.b], c = m.verts[t.c];
  float s = std::abs(a.x * (b.y - c.y) + b.x * (c.y - a.y) + c.x * (a.y - b.y)) / 2;
  return (double)s;
}
static inline V3 tri_normal(const Mesh &m, const Tri &t) {  // object.cpp:393
  return cross(sub(m.verts[t.b], m.verts[t.a]), sub(m.verts[t.c], m.verts[t.a]));
}

// Polygon::init, object.cpp:460-503 (triangulation is done by the caller)
static bool finish_mesh(World &w, Mesh &m) {
  if (m.tris.empty()) {
    w.error = "mesh without faces";
    return false;
  }
  for (const Tri &t : m.tris)
    if (t.a >= m.verts.size() || t.b >= m.verts.size() || t.c >= m.verts.size()) {
      w.error = "face index outside its own object's vertices";
      return false;
    }
  if (m.mat >= 0 && w.mats[m.mat].tex >= 0) {  // Triangle::vt0/vt1/vt2 assert this, object.cpp:542-553
    if (m.tri_vt.size() != m.tris.size()) {
      w.error = "textured object without texture coordinates";
      return false;
    }
    for (const Tri &t : m.tri_vt)
      if (t.a >= m.vts.size() || t.b >= m.vts.size() || t.c >= m.vts.size()) {
        w.error = "vt index outside its own object's texture coordinates";
        return false;
      }
  }
  std::vector<Box> tb(m.tris.size());
  for (size_t i = 0; i < m.tris.size(); ++i) {  // Triangle ctor, object.cpp:377-382
    V3 a = m.verts[m.tris[i].a], b = m.verts[m.tris[i].b], c = m.verts[m.tris[i].c];
    tb[i].lo = V3(min3(a.x, b.x, c.x), min3(a.y, b.y, c.y), min3(a.z, b.z, c.z));
    tb[i].hi = V3(max3(a.x, b.x, c.x), max3(a.y, b.y, c.y), max3(a.z, b.z, c.z));
  }
  m.area = 0;
  for (const Tri &t : m.tris) m.area += tri_area(m, t);
  Builder(m.blas, tb).run();
  return true;
}

static bool finish_world(World &w) {  // Scene::Scene, scene.cpp:15
  if (w.meshes.empty()) {
    w.error = "no objects";
    return false;
  }
  uint32_t base = 0;
  std::vector<Box> mb(w.meshes.size());
  for (size_t i = 0; i < w.meshes.size(); ++i) {
    w.meshes[i].tri_base = base;
    base += (uint32_t)w.meshes[i].tris.size();
    mb[i] = w.meshes[i].blas.nodes[0].box;  // Polygon::bbox, object.cpp:507
  }
  Builder(w.tlas, mb).run();
  w.lights.clear();
  for (uint32_t id : w.tlas.order) {  // scene.cpp:24-28
    const Mat &mt = w.mats[w.meshes[id].mat];
    if (mt.ke.x != 0 || mt.ke.y != 0 || mt.ke.z != 0) w.lights.push_back(id);
  }
  return true;
}

// ---------------------------------------------------------------- OBJ / MTL (object.cpp:555-710)
static std::string dir_of(const std::string &p);

// The reference decodes map_Kd files with FFmpeg (object.cpp:47-121), which this image does not
// have; binary PPM (P6, maxval 255) is what this restatement reads.
static bool load_ppm(const std::string &path, Tex &t) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::string magic;
  in >> magic;
  if (magic != "P6") return false;
  int vals[3], got = 0;
  while (got < 3) {
    int c = in.peek();
    if (c == '#') { std::string skip; std::getline(in, skip); continue; }
    if (c == EOF) return false;
    if (std::isspace(c)) { in.get(); continue; }
    if (!(in >> vals[got])) return false;
    ++got;
  }
  in.get();  // the single whitespace after maxval
  if (vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255) return false;
  t.w = vals[0];
  t.h = vals[1];
  t.rgb.resize((size_t)t.w * t.h * 3);
  in.read(reinterpret_cast<char *>(t.rgb.data()), (std::streamsize)t.rgb.size());
  return (size_t)in.gcount() == t.rgb.size();
}

static void load_mtl(const std::string &path, World &w, std::map<std::string, int> &names) {
  static const std::regex r_map("^\\s*map_Kd\\s+(.*?)\\s*$");
  static const std::regex r_new("^\\s*newmtl\\s+(.*?)\\s*$");
  static const std::regex r_kd("^\\s*Kd\\s+([0-9.]+)\\s+([0-9.]+)\\s+([0-9.]+)\\s*$");
  static const std::regex r_ke("^\\s*Ke\\s+([0-9.]+)\\s+([0-9.]+)\\s+([0-9.]+)\\s*$");
  std::ifstream in(path);
  std::string line;
  int cur = -1;
  std::smatch m;
  while (std::getline(in, line)) {
    if (std::regex_match(line, m, r_new)) {
      w.mats.push_back(Mat());
      cur = (int)w.mats.size() - 1;
      names[m[1].str()] = cur;
    } else if (std::regex_match(line, m, r_kd)) {
      if (cur >= 0) w.mats[cur].kd = V3(std::stof(m[1].str()), std::stof(m[2].str()), std::stof(m[3].str()));
    } else if (std::regex_match(line, m, r_ke)) {
      if (cur >= 0) w.mats[cur].ke = V3(std::stof(m[1].str()), std::stof(m[2].str()), std::stof(m[3].str()));
    } else if (std::regex_match(line, m, r_map)) {  // object.cpp:576-579: relative to the .mtl file
      if (cur >= 0) {
        Tex t;
        if (!load_ppm(dir_of(path) + m[1].str(), t)) {
          w.error = "cannot read texture " + m[1].str() + " (binary PPM expected)";
          return;
        }
        w.textures.push_back(std::move(t));
        w.mats[cur].tex = (int)w.textures.size() - 1;
      }
    }
  }
}

static std::string dir_of(const std::string &p) {
  size_t k = p.find_last_of('/');
  return k == std::string::npos ? std::string() : p.substr(0, k + 1);
}

static World *load_obj(const std::string &path) {
  static const std::regex r_v("^\\s*v\\s+(-?[0-9.]+)\\s+(-?[0-9.]+)\\s+(-?[0-9.]+)\\s*$");
  static const std::regex r_vn("^\\s*vn\\s+(-?[0-9.]+)\\s+(-?[0-9.]+)\\s+(-?[0-9.]+)\\s*$");
  static const std::regex r_vt("^\\s*vt\\s+(-?[0-9.]+)(\\s+-?[0-9.]+)?(\\s+-?[0-9.]+)?\\s*$");
  static const std::regex r_f3("^\\s*f(\\s+([0-9]+)\\/([0-9]+)\\/([0-9]+))+\\s*$");
  static const std::regex r_f2("^\\s*f(\\s+([0-9]+)\\/\\/([0-9]+))+\\s*$");
  static const std::regex c_f3("\\s+([0-9]+)\\/([0-9]+)\\/([0-9]+)");
  static const std::regex c_f2("\\s+([0-9]+)\\/\\/([0-9]+)");
  static const std::regex r_o("^o\\s(.*)$");
  static const std::regex r_lib("^\\s*mtllib\\s+(.*?)\\s*$");
  static const std::regex r_use("^\\s*usemtl\\s+(.*?)\\s*$");

  World *w = new World();
  w->mats.push_back(Mat());  // the default `new Material()` of Object::Object, object.cpp:209
  std::map<std::string, int> names;
  std::ifstream in(path);
  if (!in) {
    w->error = "cannot open " + path;
    return w;
  }
  struct Pending {
    Mesh m;
    std::vector<std::vector<uint32_t>> faces;
    std::vector<std::vector<uint32_t>> faces_vt;  // the vt index of every corner (0xffffffff: the a//c form has none)
    size_t nvn = 0;
  };
  Pending *cur = nullptr;
  float off_v = 0, off_vt = 0;  // veci stores floats, math.cpp:22,191
  bool ok = true;
  auto close = [&]() {
    if (!cur) return;
    for (size_t k = 0; k < cur->faces.size(); ++k) {  // fan about corner 0, object.cpp:478-490
      const auto &f = cur->faces[k];
      const auto &ft = cur->faces_vt[k];
      for (size_t i = 1; i + 1 < f.size(); ++i) {
        cur->m.tris.push_back(Tri{f[0], f[i], f[i + 1]});
        cur->m.tri_vt.push_back(Tri{ft[0], ft[i], ft[i + 1]});
      }
    }
    if (cur->m.mat < 0) cur->m.mat = 0;
    if (!finish_mesh(*w, cur->m)) ok = false;
    off_v += (float)cur->m.verts.size();
    off_vt += (float)cur->m.vts.size();  // object.cpp:679-680
    w->meshes.push_back(std::move(cur->m));
    delete cur;
    cur = nullptr;
  };
  std::string line;
  std::smatch m;
  while (ok && std::getline(in, line)) {
    if (std::regex_match(line, m, r_v)) {
      if (!cur) { w->error = "v before o"; ok = false; break; }
      cur->m.verts.push_back(V3(std::stof(m[1].str()), std::stof(m[2].str()), std::stof(m[3].str())));
    } else if (std::regex_match(line, m, r_vn)) {
      if (!cur) { w->error = "vn before o"; ok = false; break; }
      ++cur->nvn;
    } else if (std::regex_match(line, m, r_vt)) {
      if (!cur) { w->error = "vt before o"; ok = false; break; }
      cur->m.vts.push_back(V3(std::stof(m[1].str()), m[2].matched ? std::stof(m[2].str()) : 0,
                              m[3].matched ? std::stof(m[3].str()) : 0));  // object.cpp:662-668
    } else if (std::regex_match(line, m, r_f3) || std::regex_match(line, m, r_f2)) {
      if (!cur) { w->error = "f before o"; ok = false; break; }
      bool three = std::regex_match(line, r_f3);
      const std::regex &cr = three ? c_f3 : c_f2;
      std::vector<uint32_t> face, face_vt;
      for (std::sregex_iterator it(line.begin(), line.end(), cr), e; it != e; ++it) {
        float idx = (float)std::stoi((*it)[1].str());  // object.cpp:620,629-635
        idx -= 1;
        idx -= off_v;
        face.push_back((uint32_t)idx);
        if (three) {
          float ti = (float)std::stoi((*it)[2].str());
          ti -= 1;
          ti -= off_vt;
          face_vt.push_back(ti >= 0 ? (uint32_t)ti : 0xffffffffu);
        } else {
          face_vt.push_back(0xffffffffu);  // veci(l[0], 0, l[1]) - 1 is negative there, object.cpp:672
        }
      }
      cur->faces.push_back(face);
      cur->faces_vt.push_back(face_vt);
    } else if (std::regex_match(line, m, r_o)) {
      close();
      cur = new Pending();
    } else if (std::regex_match(line, m, r_lib)) {
      load_mtl(dir_of(path) + m[1].str(), *w, names);
      if (!w->error.empty()) { ok = false; break; }
    } else if (std::regex_match(line, m, r_use)) {
      if (!cur) { w->error = "usemtl before o"; ok = false; break; }
      auto f = names.find(m[1].str());
      if (f == names.end()) { w->error = "usemtl names an unknown material"; ok = false; break; }
      cur->m.mat = f->second;
    }
  }
  if (ok) close();
  if (cur) delete cur;
  if (ok && w->error.empty()) finish_world(*w);
  return w;
}

// ---------------------------------------------------------------- random numbers
static inline void philox(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
  uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
  uint32_t k0 = k_in[0], k1 = k_in[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One source of uniform doubles in [0,1) per path.
struct Rng {
  int kind = 0;  // 0 counter, 1 mt19937
  uint32_t key[2] = {0, 0};
  uint32_t pixel = 0, sample = 0;
  std::mt19937 *mt = nullptr;
  // draw number `dim` of this path
  double u01(uint32_t dim) {
    if (kind == 1) return std::generate_canonical<double, 53>(*mt);  // what uniform_real_distribution<> draws
    uint32_t c[4] = {pixel, sample, dim >> 1, 0}, o[4];
    philox(c, key, o);
    uint32_t hi = o[2 * (dim & 1)], lo = o[2 * (dim & 1) + 1];
    uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)m * (1.0 / 9007199254740992.0);
  }
  // randreal(x, y), math.cpp:13-18: the distribution works in double, the result is narrowed.
  float real(float x, float y, uint32_t dim) { return (float)(u01(dim) * ((double)y - (double)x) + (double)x); }
};

// ---------------------------------------------------------------- integrator
struct Frame {
  const World *w;
  PtCamera cam;
  PtOptions opt;
  int W, H;
  float pin_z;         // third component of the pinhole direction
  uint32_t dims_per_vertex;
};

static inline uint32_t vertex_dims(size_t n_lights) {
  uint32_t d = 3u * (uint32_t)n_lights + 3u;
  return (d + 1u) & ~1u;
}

// Camera::point_towards before the rotation, camera.cpp:161-167
static inline V3 pinhole_dir(const Frame &f, int px, int py) {
  float x = (px - (float)f.W / 2) * f.cam.vw / f.W;
  float y = (py - (float)f.H / 2) * -f.cam.vh / f.H;
  return V3(x, y, f.pin_z);
}
static inline float pinhole_z(const PtCamera &c) {
  return (float)(1 / (2 * std::tan(c.fov_deg / 2 * (M_PI / 180)) / c.vw));
}

static void camera_ray(const Frame &f, int px, int py, Rng &rng, V3 &o, V3 &d) {
  V3 dc = pinhole_dir(f, px, py);
  if (f.cam.aperture > 0) {  // thin lens: an extension, absent from the reference
    float u1 = rng.real(0, 1, 0), u2 = rng.real(0, 1, 1);
    float r = f.cam.aperture * std::sqrt(u1);
    V3 lens((float)(std::cos(2 * M_PI * u2) * r), (float)(std::sin(2 * M_PI * u2) * r), 0);
    float s = f.cam.focus / dc.z;
    V3 pf = scale(dc, s);
    float inv_s = dc.z / f.cam.focus;
    V3 dl = scale(sub(pf, lens), inv_s);
    o = add(V3(f.cam.origin[0], f.cam.origin[1], f.cam.origin[2]),
            rotate(lens, f.cam.euler[0], f.cam.euler[1], f.cam.euler[2]));
    d = rotate(dl, f.cam.euler[0], f.cam.euler[1], f.cam.euler[2]);
  } else {
    o = V3(f.cam.origin[0], f.cam.origin[1], f.cam.origin[2]);
    d = rotate(dc, f.cam.euler[0], f.cam.euler[1], f.cam.euler[2]);  // camera.cpp:168
  }
}

// Polygon::sample + Triangle::sample, object.cpp:513-523,432-442
static void sample_light(const Mesh &m, Rng &rng, uint32_t base, V3 &point, V3 &nrm) {
  double r = (double)rng.real(0, (float)m.area, base);
  double accum = 0;
  size_t pick = m.tris.size() - 1;  // the reference reaches std::unreachable() here; take the last
  for (size_t i = 0; i < m.tris.size(); ++i) {
    accum += tri_area(m, m.tris[i]);
    if (r <= accum) {
      pick = i;
      break;
    }
  }
  const Tri &t = m.tris[pick];
  V3 u = sub(m.verts[t.b], m.verts[t.a]);
  V3 v = sub(m.verts[t.c], m.verts[t.a]);
  double s = (double)rng.real(0, 1, base + 1);
  double tt = (double)rng.real(0, 1, base + 2);
  if (s + tt <= 1)
    point = add(scale(u, (float)s), scale(v, (float)tt));  // v0 is not added, object.cpp:439
  else
    point = add(scale(u, (float)(1 - s)), scale(v, (float)(1 - tt)));
  nrm = tri_normal(m, t);
}

// random_direction, camera.cpp:77-90
static V3 uniform_dir(V3 n, float x, float y) {
  float phi = 1 - 2 * x;
  float theta = (float)std::fmax((double)0, (double)(1 - phi * phi));
  float st = std::sqrt(theta);
  V3 v((float)(std::cos(2 * M_PI * y) * st), (float)(std::sin(2 * M_PI * y) * st), phi);
  v = normalize(v);
  return dot(v, n) < 0 ? neg(v) : v;
}
// Cosine-weighted direction about n: an extension, absent from the reference.
static V3 cosine_dir(V3 n, float x, float y) {
  float r = std::sqrt(x);
  float lz = std::sqrt(std::fmax(0.0f, 1 - x));
  float lx = (float)(std::cos(2 * M_PI * y) * r), ly = (float)(std::sin(2 * M_PI * y) * r);
  float s = std::copysign(1.0f, n.z);
  float a = -1 / (s + n.z);
  float b = n.x * n.y * a;
  V3 t1(1 + s * n.x * n.x * a, s * b, -s * n.x);
  V3 t2(b, s + n.y * n.y * a, -n.y);
  return add(add(scale(t1, lx), scale(t2, ly)), scale(n, lz));
}

// The light loop of camera.cpp:96-115 at one path vertex.
static V3 direct_light(const Frame &f, V3 x, V3 n, Rng &rng, uint32_t base) {
  const World &w = *f.w;
  V3 color(0, 0, 0);
  for (size_t l = 0; l < w.lights.size(); ++l) {
    const Mesh &lm = w.meshes[w.lights[l]];
    V3 point, ln;
    sample_light(lm, rng, base + 3u * (uint32_t)l, point, ln);
    V3 to = normalize(sub(point, x));
    ++tl_cnt.rays_shadow;
    if (castray(w, x, to).t == kInf) continue;
    float c = dot(to, n);
    if (c < 0) c = 0;
    float y = dot(to, ln);
    if (y < 0) y = 0;
    double nn = (double)norm(to);
    y = (float)((double)y / std::pow(nn, 2));
    V3 ke = w.mats[lm.mat].ke;
    float inv_area = (float)(1 / lm.area);
    color = add(color, scale(scale(scale(ke, c), y), inv_area));
  }
  return color;
}

struct Vertex {
  V3 x, n;
  const Mat *mat;
  bool textured;
  V3 texel;  // the brdf of a textured vertex, camera.cpp:124-146
};

// static_cast<int>(scalar) of camera.cpp:142-143 as x86-64 performs it (cvttss2si): INT_MIN when
// the value does not fit.
static inline int cvt_int(float f) {
  return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? (int)0x80000000 : (int)f;
}
static inline V3 texel_brdf(const World &w, const Mesh &m, const Hit &h) {
  const float beta = h.u, gamma = h.v, alpha = h.w;  // auto [beta,gamma,alpha] = hit.barycentric, camera.cpp:124
  const Tri &vt = m.tri_vt[h.tri];
  const V3 t0 = m.vts[vt.a], t1 = m.vts[vt.b], t2 = m.vts[vt.c];
  const float u = alpha * t0.x + beta * t1.x + gamma * t2.x;  // camera.cpp:137-138
  const float v = alpha * t0.y + beta * t1.y + gamma * t2.y;
  const Tex &img = w.textures[w.mats[m.mat].tex];
  const int x = std::clamp(cvt_int(u * img.w), 0, img.w - 1);  // camera.cpp:141-145
  const int y = std::clamp(cvt_int(v * img.h), 0, img.h - 1);
  const uint8_t *p = &img.rgb[((size_t)y * img.w + x) * 3];
  return divs(V3(p[0], p[1], p[2]), 255);  // Image::at, object.cpp:128-134
}
static inline Vertex shade_setup(const World &w, const Hit &h, V3 o, V3 d) {
  Vertex v;
  v.x = add(o, scale(d, h.t));  // camera.cpp:68
  const Mesh &m = w.meshes[h.mesh];
  v.n = normalize(tri_normal(m, m.tris[h.tri]));  // camera.cpp:69
  if (dot(v.n, normalize(d)) > 0) v.n = neg(v.n);  // camera.cpp:72-73
  v.mat = &w.mats[m.mat];
  v.textured = v.mat->tex >= 0;
  if (v.textured) v.texel = texel_brdf(w, m, h);
  return v;
}
// camera.cpp:118-119: `randreal(0, 1) < rr_prob` with `constexpr double rr_prob = .7`: the float draw is widened and
// compared with the double, so a draw equal to 0.7f (just below .7) continues.
static inline bool roulette_pass(double rr_prob, float draw) { return (double)draw < rr_prob; }
static inline bool keep_going(const Frame &f, int depth, Rng &rng, uint32_t base) {
  uint32_t L = (uint32_t)f.w->lights.size();
  bool cont = depth <= f.opt.rr_min_depth || roulette_pass(f.opt.rr_prob, rng.real(0, 1, base + 3 * L));
  if (f.opt.max_depth > 0 && depth >= f.opt.max_depth) cont = false;
  return cont;
}
static inline V3 next_dir(const Frame &f, V3 n, Rng &rng, uint32_t base) {
  uint32_t L = (uint32_t)f.w->lights.size();
  float x = rng.real(0, 1, base + 3 * L + 1);
  float y = rng.real(0, 1, base + 3 * L + 2);
  return f.opt.sampling == PT_SAMPLE_COSINE ? cosine_dir(n, x, y) : uniform_dir(n, x, y);
}
// Divisor of the bounce weight: camera.cpp:152 narrowed by operator/(vec, scalar).
static inline float pdf_f(const Frame &f) { return (float)(1 / (2 * M_PI) * f.opt.rr_prob); }

// Camera::pathtrace as written, camera.cpp:63-159.
static V3 trace_recursive(const Frame &f, V3 o, V3 d, int depth, Rng &rng) {
  ++tl_cnt.rays_closest;
  Hit h = castray(*f.w, o, d);
  if (h.t == kInf) return V3(f.opt.bgcolor[0], f.opt.bgcolor[1], f.opt.bgcolor[2]);
  Vertex v = shade_setup(*f.w, h, o, d);
  uint32_t base = 2u + (uint32_t)depth * f.dims_per_vertex;
  V3 color = direct_light(f, v.x, v.n, rng, base);
  if (keep_going(f, depth, rng, base)) {
    V3 w = next_dir(f, v.n, rng, base);
    color = add(color, v.mat->ke);
    V3 child = trace_recursive(f, v.x, w, depth + 1, rng);
    if (f.opt.sampling == PT_SAMPLE_COSINE) {
      color = add(color, divs(mul(child, v.textured ? scale(v.texel, (float)M_PI) : v.mat->kd), (float)f.opt.rr_prob));
    } else {
      V3 brdf = v.textured ? v.texel : divs(v.mat->kd, (float)M_PI);  // camera.cpp:121-146
      color = add(color, divs(scale(mul(child, brdf), std::abs(dot(v.n, w))), pdf_f(f)));
    }
  }
  return color;
}

// The same estimator carried forward with a running throughput; this is the
// operation order the GPU kernels use.
static V3 trace_iterative(const Frame &f, V3 o, V3 d, Rng &rng) {
  V3 T(1, 1, 1), L(0, 0, 0);
  for (int depth = 0;; ++depth) {
    ++tl_cnt.rays_closest;
    Hit h = castray(*f.w, o, d);
    if (h.t == kInf) {
      L = add(L, mul(T, V3(f.opt.bgcolor[0], f.opt.bgcolor[1], f.opt.bgcolor[2])));
      break;
    }
    Vertex v = shade_setup(*f.w, h, o, d);
    uint32_t base = 2u + (uint32_t)depth * f.dims_per_vertex;
    V3 local = direct_light(f, v.x, v.n, rng, base);
    bool cont = keep_going(f, depth, rng, base);
    if (cont) local = add(local, v.mat->ke);
    L = add(L, mul(T, local));
    if (!cont) break;
    V3 w = next_dir(f, v.n, rng, base);
    if (f.opt.sampling == PT_SAMPLE_COSINE) {
      T = divs(mul(T, v.textured ? scale(v.texel, (float)M_PI) : v.mat->kd), (float)f.opt.rr_prob);
    } else {
      V3 brdf = v.textured ? v.texel : divs(v.mat->kd, (float)M_PI);
      T = divs(scale(mul(T, brdf), std::abs(dot(v.n, w))), pdf_f(f));
    }
    o = v.x;
    d = w;
  }
  return L;
}

static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace orc

using namespace orc;

// ================================================================ C interface for ctypes
extern "C" {

struct OrcRenderArgs {
  int32_t W, H;
  uint32_t spp_begin, spp_count;
  uint64_t seed;
  int32_t rng;      // 0 counter-based, 1 mt19937 seeded per (sample,row)
  int32_t form;     // 0 recursive (camera.cpp:63-159 as written), 1 iterative throughput
  int32_t threads;  // 0 = std::thread::hardware_concurrency(), camera.cpp:49
  int32_t running_mean;  // 1: out = Camera::pixels (camera.cpp:188-190); 0: out += radiance sums
};

const char *orc_error(const World *w) { return w->error.empty() ? nullptr : w->error.c_str(); }
World *orc_load_obj(const char *path) { return load_obj(path); }
void orc_free(World *w) { delete w; }

World *orc_from_arrays(uint32_t n_meshes, const uint32_t *vert_off, const float *verts,
                       const uint32_t *tri_off, const uint32_t *tris, const uint32_t *mesh_mat,
                       uint32_t n_mats, const float *kd, const float *ke) {
  World *w = new World();
  for (uint32_t i = 0; i < n_mats; ++i) {
    Mat m;
    m.kd = V3(kd[3 * i], kd[3 * i + 1], kd[3 * i + 2]);
    m.ke = V3(ke[3 * i], ke[3 * i + 1], ke[3 * i + 2]);
    w->mats.push_back(m);
  }
  w->meshes.resize(n_meshes);
  for (uint32_t i = 0; i < n_meshes; ++i) {
    Mesh &m = w->meshes[i];
    for (uint32_t k = vert_off[i]; k < vert_off[i + 1]; ++k)
      m.verts.push_back(V3(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]));
    for (uint32_t k = tri_off[i]; k < tri_off[i + 1]; ++k)
      m.tris.push_back(Tri{tris[3 * k], tris[3 * k + 1], tris[3 * k + 2]});
    m.mat = (int)mesh_mat[i];
    if (m.mat < 0 || m.mat >= (int)n_mats) {
      w->error = "material index out of range";
      return w;
    }
    if (!finish_mesh(*w, m)) return w;
  }
  finish_world(*w);
  return w;
}

void orc_counts(const World *w, uint32_t *n_meshes, uint32_t *n_tris, uint32_t *n_tlas_nodes,
                uint32_t *n_blas_nodes, uint32_t *n_lights, uint32_t *n_mats) {
  uint32_t t = 0, b = 0;
  for (const Mesh &m : w->meshes) {
    t += (uint32_t)m.tris.size();
    b += (uint32_t)m.blas.nodes.size();
  }
  *n_meshes = (uint32_t)w->meshes.size();
  *n_tris = t;
  *n_tlas_nodes = (uint32_t)w->tlas.nodes.size();
  *n_blas_nodes = b;
  *n_lights = (uint32_t)w->lights.size();
  *n_mats = (uint32_t)w->mats.size();
}

// Everything a PtSceneDesc needs, copied into caller-allocated arrays.
// Textures and texture coordinates, in the shape PtSceneDesc wants them.
uint32_t orc_n_textures(const World *w) { return (uint32_t)w->textures.size(); }
void orc_texture_size(const World *w, uint32_t i, uint32_t *width, uint32_t *height) {
  *width = (uint32_t)w->textures[i].w;
  *height = (uint32_t)w->textures[i].h;
}
void orc_texture_data(const World *w, uint32_t i, uint8_t *rgb) {
  std::memcpy(rgb, w->textures[i].rgb.data(), w->textures[i].rgb.size());
}
// The brdf camera.cpp:124-146 takes from the diffuse map for a hit on triangle `tri` of `mesh`
// with barycentrics (u, v); returns 0 when the mesh's material has no map.
int orc_texel(const World *w, uint32_t mesh, uint32_t tri, float u, float v, float *rgb) {
  const Mesh &m = w->meshes[mesh];
  if (w->mats[m.mat].tex < 0) return 0;
  Hit h;
  h.mesh = (int)mesh; h.tri = (int)tri; h.u = u; h.v = v; h.w = 1 - u - v;
  const V3 c = texel_brdf(*w, m, h);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
  return 1;
}

void orc_export_textured(const World *w, int32_t *material_texture, float *tri_uv) {
  for (size_t i = 0; i < w->mats.size(); ++i) material_texture[i] = w->mats[i].tex;
  size_t t = 0;
  for (const Mesh &m : w->meshes)
    for (size_t k = 0; k < m.tris.size(); ++k, ++t) {
      const bool has = m.tri_vt.size() == m.tris.size() && m.tri_vt[k].a < m.vts.size() && m.tri_vt[k].b < m.vts.size() &&
                       m.tri_vt[k].c < m.vts.size();
      const uint32_t id[3] = {has ? m.tri_vt[k].a : 0, has ? m.tri_vt[k].b : 0, has ? m.tri_vt[k].c : 0};
      for (int c = 0; c < 3; ++c) {
        tri_uv[t * 6 + c * 2 + 0] = has ? m.vts[id[c]].x : 0.0f;
        tri_uv[t * 6 + c * 2 + 1] = has ? m.vts[id[c]].y : 0.0f;
      }
    }
}

void orc_export(const World *w, PtNode *tlas_nodes, uint32_t *tlas_order, uint32_t *blas_node_offset,
                PtNode *blas_nodes, uint32_t *tri_offset, uint32_t *blas_order, float *tri_verts,
                uint32_t *mesh_material, PtMaterial *mats, uint32_t *lights, double *mesh_area) {
  auto put = [](PtNode &o, const Node &n) {
    o.index_or_first = n.ref;
    o.count = n.count;
    o.bmin[0] = n.box.lo.x; o.bmin[1] = n.box.lo.y; o.bmin[2] = n.box.lo.z;
    o.bmax[0] = n.box.hi.x; o.bmax[1] = n.box.hi.y; o.bmax[2] = n.box.hi.z;
  };
  for (size_t i = 0; i < w->tlas.nodes.size(); ++i) put(tlas_nodes[i], w->tlas.nodes[i]);
  for (size_t i = 0; i < w->tlas.order.size(); ++i) tlas_order[i] = w->tlas.order[i];
  uint32_t nb = 0, nt = 0;
  for (size_t i = 0; i < w->meshes.size(); ++i) {
    const Mesh &m = w->meshes[i];
    blas_node_offset[i] = nb;
    tri_offset[i] = nt;
    for (size_t k = 0; k < m.blas.nodes.size(); ++k) put(blas_nodes[nb + k], m.blas.nodes[k]);
    for (size_t k = 0; k < m.tris.size(); ++k) {
      blas_order[nt + k] = m.blas.order[k];
      const V3 p[3] = {m.verts[m.tris[k].a], m.verts[m.tris[k].b], m.verts[m.tris[k].c]};
      for (int c = 0; c < 3; ++c) {
        tri_verts[(size_t)(nt + k) * 9 + c * 3 + 0] = p[c].x;
        tri_verts[(size_t)(nt + k) * 9 + c * 3 + 1] = p[c].y;
        tri_verts[(size_t)(nt + k) * 9 + c * 3 + 2] = p[c].z;
      }
    }
    mesh_material[i] = (uint32_t)m.mat;
    if (mesh_area) mesh_area[i] = m.area;
    nb += (uint32_t)m.blas.nodes.size();
    nt += (uint32_t)m.tris.size();
  }
  blas_node_offset[w->meshes.size()] = nb;
  tri_offset[w->meshes.size()] = nt;
  for (size_t i = 0; i < w->mats.size(); ++i) {
    mats[i].kd[0] = w->mats[i].kd.x; mats[i].kd[1] = w->mats[i].kd.y; mats[i].kd[2] = w->mats[i].kd.z;
    mats[i].ke[0] = w->mats[i].ke.x; mats[i].ke[1] = w->mats[i].ke.y; mats[i].ke[2] = w->mats[i].ke.z;
  }
  if (lights)
    for (size_t i = 0; i < w->lights.size(); ++i) lights[i] = w->lights[i];
}

// Scene::castray over n rays.  tri is the global triangle id in file order.
// counters (may be null): [0] node records read, [1] triangles tested.
void orc_castray(const World *w, uint64_t n, const float *ox, const float *oy, const float *oz,
                 const float *dx, const float *dy, const float *dz, float *t, int32_t *tri,
                 float *u, float *v, int32_t *hits, uint64_t *counters) {
  tl_cnt = Counters();
  for (uint64_t i = 0; i < n; ++i) {
    Hit h = castray(*w, V3(ox[i], oy[i], oz[i]), V3(dx[i], dy[i], dz[i]));
    t[i] = h.t;
    tri[i] = h.mesh < 0 ? -1 : (int32_t)(w->meshes[h.mesh].tri_base + (uint32_t)h.tri);
    if (u) u[i] = h.u;
    if (v) v[i] = h.v;
    if (hits) hits[i] = h.hits;
  }
  if (counters) {
    counters[0] = tl_cnt.nodes;
    counters[1] = tl_cnt.tris;
  }
}

// Camera::point_towards for all pixels, camera.cpp:161-169.
void orc_camera_rays(const PtCamera *cam, int32_t W, int32_t H, float *dir_xyz) {
  Frame f;
  f.w = nullptr;
  f.cam = *cam;
  f.W = W;
  f.H = H;
  f.pin_z = pinhole_z(*cam);
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      V3 d = rotate(pinhole_dir(f, px, py), cam->euler[0], cam->euler[1], cam->euler[2]);
      float *o = dir_xyz + ((size_t)py * W + px) * 3;
      o[0] = d.x; o[1] = d.y; o[2] = d.z;
    }
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox(ctr, key, out); }

// Camera::render, camera.cpp:171-200, with rows handed to threads from a shared
// counter as parallel_for does (camera.cpp:14-39).
int orc_render(const World *w, const PtCamera *cam, const PtOptions *opt, const OrcRenderArgs *a,
               float *out_rgb, PtCounters *counters) {
  if (!w->error.empty()) return -1;
  Frame f;
  f.w = w;
  f.cam = *cam;
  f.opt = *opt;
  f.W = a->W;
  f.H = a->H;
  f.pin_z = pinhole_z(*cam);
  f.dims_per_vertex = vertex_dims(w->lights.size());
  const uint32_t stride = opt->spp_stride ? opt->spp_stride : 1;
  int nthreads = a->threads > 0 ? a->threads : (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  if (a->running_mean)
    for (size_t i = 0; i < (size_t)a->W * a->H * 3; ++i) out_rgb[i] = 0;  // camera.cpp:177-179
  std::vector<Counters> tc((size_t)nthreads);
  for (uint32_t k = 0; k < a->spp_count; ++k) {
    const uint32_t sample = a->spp_begin + k * stride;
    std::atomic<int> next_row{0};
    auto runner = [&](int tid) {
      tl_cnt = Counters();
      for (;;) {
        int py = next_row.fetch_add(1);
        if (py >= a->H) break;
        std::mt19937 mt;
        Rng rng;
        rng.kind = a->rng;
        rng.key[0] = (uint32_t)a->seed;
        rng.key[1] = (uint32_t)(a->seed >> 32);
        rng.sample = sample;
        if (a->rng == 1) {
          mt.seed((uint32_t)mix64(a->seed ^ mix64(((uint64_t)sample << 32) | (uint32_t)py)));
          rng.mt = &mt;
        }
        for (int px = 0; px < a->W; ++px) {
          rng.pixel = (uint32_t)py * (uint32_t)a->W + (uint32_t)px;
          V3 o, d;
          camera_ray(f, px, py, rng, o, d);
          ++tl_cnt.paths;
          V3 L = a->form == 0 ? trace_recursive(f, o, d, 0, rng) : trace_iterative(f, o, d, rng);
          float *p = out_rgb + ((size_t)py * a->W + px) * 3;
          if (a->running_mean) {  // camera.cpp:188-190
            float i = (float)k;   // int i promoted by operator*(scalar, vec)
            p[0] = (i * p[0] + L.x * 255) / (float)(k + 1);
            p[1] = (i * p[1] + L.y * 255) / (float)(k + 1);
            p[2] = (i * p[2] + L.z * 255) / (float)(k + 1);
          } else {
            p[0] += L.x; p[1] += L.y; p[2] += L.z;
          }
        }
      }
      Counters &c = tc[(size_t)tid];
      c.paths += tl_cnt.paths; c.rays_closest += tl_cnt.rays_closest; c.rays_shadow += tl_cnt.rays_shadow;
      c.nodes += tl_cnt.nodes; c.tris += tl_cnt.tris;
    };
    std::vector<std::thread> th;
    int extra = std::min(nthreads - 1, a->H - 1);  // camera.cpp:32
    for (int i = 0; i < extra; ++i) th.emplace_back(runner, i + 1);
    runner(0);
    for (auto &t : th) t.join();
  }
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (const Counters &c : tc) {
      counters->paths += c.paths;
      counters->rays_closest += c.rays_closest;
      counters->rays_shadow += c.rays_shadow;
      counters->node_boxes += c.nodes;
      counters->tri_tests += c.tris;
    }
  }
  return 0;
}

// Camera::save, camera.cpp:204-218: P3, integer truncation, no clamp.
int orc_save_ppm(const char *path, const float *pixels, int32_t W, int32_t H) {
  std::ofstream ofs(path);
  if (!ofs) return -1;
  ofs << "P3\n" << W << " " << H << "\n" << 255 << "\n";
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float *p = pixels + ((size_t)y * W + x) * 3;
      ofs << (int)p[0] << " " << (int)p[1] << " " << (int)p[2] << "\n";
    }
  return 0;
}

int orc_roulette_pass(double rr_prob, float draw) { return roulette_pass(rr_prob, draw) ? 1 : 0; }

int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"

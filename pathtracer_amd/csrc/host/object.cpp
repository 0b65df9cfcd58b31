// object.cpp — host-side BVH construction and OBJ/MTL loading.
#include "object.h"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <numeric>
#include <sstream>

namespace pt {

bool verbose = false;

// ------------------------------------------------------------------ BVH
namespace {

inline scalar lesser(scalar a, scalar b) { return b < a ? b : a; }    // std::min
inline scalar greater(scalar a, scalar b) { return a < b ? b : a; }   // std::max

struct BuildCtx {
  const std::vector<AABB> &boxes;
  std::vector<vec> centroid;  // Object::centroid, object.cpp:219
  std::vector<PtNode> &nodes;
  std::vector<uint32_t> &ids;

  BuildCtx(const std::vector<AABB> &b, std::vector<PtNode> &n, std::vector<uint32_t> &i)
      : boxes(b), nodes(n), ids(i) {
    centroid.resize(b.size());
    for (size_t k = 0; k < b.size(); ++k) centroid[k] = .5f * b[k].min + .5f * b[k].max;
  }

  // update_box, object.cpp:296-308
  void refit(uint32_t i) {
    scalar lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
    const uint32_t first = nodes[i].index_or_first, count = nodes[i].count;
    for (uint32_t k = first; k < first + count; ++k) {
      const AABB &b = boxes[ids[k]];
      lo[0] = lesser(lo[0], b.min.x); lo[1] = lesser(lo[1], b.min.y); lo[2] = lesser(lo[2], b.min.z);
      hi[0] = greater(hi[0], b.max.x); hi[1] = greater(hi[1], b.max.y); hi[2] = greater(hi[2], b.max.z);
    }
    for (int a = 0; a < 3; ++a) {
      nodes[i].bmin[a] = lo[a];
      nodes[i].bmax[a] = hi[a];
    }
  }

  // One step of subdivide (object.cpp:313-361).  Returns the index of the left child, or 0
  // when node i stays a leaf.
  uint32_t split(uint32_t i) {
    const uint32_t count = nodes[i].count, first = nodes[i].index_or_first;
    if (count <= BVH::max_leaf_size) return 0;
    const scalar ex = nodes[i].bmax[0] - nodes[i].bmin[0];
    const scalar ey = nodes[i].bmax[1] - nodes[i].bmin[1];
    const scalar ez = nodes[i].bmax[2] - nodes[i].bmin[2];
    const scalar ext[3] = {ex, ey, ez};
    int axis = 0;
    if (ey > ex) axis = 1;
    if (ez > ext[axis]) axis = 2;
    const scalar mid = nodes[i].bmin[axis] + ext[axis] / 2;
    auto begin = ids.begin() + first, end = begin + count;
    auto cut = std::partition(begin, end, [&](uint32_t id) { return centroid[id][axis] < mid; });
    uint32_t m = (uint32_t)(cut - begin);
    if (m == 0 || m >= count - 1) {
      std::nth_element(begin, begin + count / 2, end,
                       [&](uint32_t p, uint32_t q) { return centroid[p][axis] < centroid[q][axis]; });
      m = count / 2;
    }
    const uint32_t left = (uint32_t)nodes.size();
    PtNode a{}, b{};
    a.index_or_first = first;
    a.count = m + 1;  // the left child takes one item more than the partition point (object.cpp:348)
    b.index_or_first = first + m + 1;
    b.count = count - (m + 1);
    nodes.push_back(a);
    nodes.push_back(b);
    nodes[i].index_or_first = left;
    nodes[i].count = 0;
    refit(left);
    refit(left + 1);
    return left;
  }
};

}  // namespace

BVH::BVH(const std::vector<AABB> &item_boxes) {
  objects.resize(item_boxes.size());
  std::iota(objects.begin(), objects.end(), 0u);
  if (item_boxes.empty()) return;
  BuildCtx ctx(item_boxes, nodes, objects);
  PtNode root{};
  root.index_or_first = 0;
  root.count = (uint32_t)item_boxes.size();
  nodes.push_back(root);
  ctx.refit(0);
  // The reference recurses left child first (object.cpp:359-360); an explicit stack that
  // pops the left child first numbers the nodes identically without deep recursion.
  std::vector<uint32_t> todo{0};
  while (!todo.empty()) {
    const uint32_t i = todo.back();
    todo.pop_back();
    const uint32_t left = ctx.split(i);
    if (left) {
      todo.push_back(left + 1);
      todo.push_back(left);
    }
  }
}

AABB BVH::bbox() const {
  AABB b;
  if (objects.empty() || nodes.empty()) return b;
  b.min = vec(nodes[0].bmin[0], nodes[0].bmin[1], nodes[0].bmin[2]);
  b.max = vec(nodes[0].bmax[0], nodes[0].bmax[1], nodes[0].bmax[2]);
  return b;
}

// ------------------------------------------------------------------ Polygon
double Polygon::triangle_area(size_t t) const {
  const vec a = v0(t), b = v1(t), c = v2(t);
  return (double)(std::abs(a.x * (b.y - c.y) + b.x * (c.y - a.y) + c.x * (a.y - b.y)) / 2);
}

AABB Polygon::triangle_bbox(size_t t) const {
  const vec a = v0(t), b = v1(t), c = v2(t);
  auto lo3 = [](scalar p, scalar q, scalar r) { scalar m = p; if (q < m) m = q; if (r < m) m = r; return m; };
  auto hi3 = [](scalar p, scalar q, scalar r) { scalar m = p; if (m < q) m = q; if (m < r) m = r; return m; };
  AABB box;
  box.min = vec(lo3(a.x, b.x, c.x), lo3(a.y, b.y, c.y), lo3(a.z, b.z, c.z));
  box.max = vec(hi3(a.x, b.x, c.x), hi3(a.y, b.y, c.y), hi3(a.z, b.z, c.z));
  return box;
}

bool Polygon::init(std::string *err) {
  if (triangles.empty()) {
    if (err) *err = "object without faces";
    return false;
  }
  for (const auto &t : triangles)
    for (uint32_t i : t)
      if (i >= vertices.size()) {
        if (err) *err = "face refers to a vertex outside its own object";
        return false;
      }
  if (verbose) std::printf("v:%zu f:%zu\n", vertices.size(), triangles.size());
  area_ = 0;
  std::vector<AABB> boxes(triangles.size());
  for (size_t t = 0; t < triangles.size(); ++t) {
    area_ += triangle_area(t);
    boxes[t] = triangle_bbox(t);
  }
  bvh = BVH(boxes);
  return true;
}

// ------------------------------------------------------------------ OBJ / MTL
namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

// Whitespace-separated tokens of a line as (begin, end) offsets.
void tokens_of(const std::string &s, std::vector<std::pair<size_t, size_t>> &out) {
  out.clear();
  size_t i = 0, n = s.size();
  while (i < n) {
    while (i < n && is_space(s[i])) ++i;
    if (i >= n) break;
    size_t b = i;
    while (i < n && !is_space(s[i])) ++i;
    out.emplace_back(b, i);
  }
}

// "-?[0-9.]+" (or "[0-9.]+" when `sign` is false)
bool number_syntax(const std::string &s, size_t b, size_t e, bool sign) {
  if (sign && b < e && s[b] == '-') ++b;
  if (b >= e) return false;
  for (size_t i = b; i < e; ++i)
    if (!((s[i] >= '0' && s[i] <= '9') || s[i] == '.')) return false;
  return true;
}

// std::stof on a token: longest prefix strtof accepts; failure where stof would throw
bool to_float(const std::string &s, size_t b, size_t e, float &out) {
  char buf[64];
  std::string big;
  const char *p;
  if (e - b < sizeof(buf)) {
    memcpy(buf, s.data() + b, e - b);
    buf[e - b] = 0;
    p = buf;
  } else {
    big.assign(s, b, e - b);
    p = big.c_str();
  }
  char *end = nullptr;
  errno = 0;
  out = std::strtof(p, &end);
  return end != p && errno != ERANGE;
}

// one face corner "a/b/c" or "a//c"; returns the first integer and, for a/b/c, the second
bool corner_index(const std::string &s, size_t b, size_t e, bool &three, long &idx, long &vt) {
  size_t i = b;
  auto digits = [&](long *val) {
    size_t st = i;
    long v = 0;
    bool over = false;
    while (i < e && s[i] >= '0' && s[i] <= '9') {
      v = v * 10 + (s[i] - '0');
      if (v > 2147483647L) over = true;
      ++i;
    }
    if (val) *val = over ? -1 : v;
    return i > st;
  };
  if (!digits(&idx)) return false;
  if (i >= e || s[i] != '/') return false;
  ++i;
  if (i < e && s[i] == '/') {  // a//c
    ++i;
    three = false;
    if (!digits(nullptr)) return false;
    return i == e;
  }
  three = true;
  if (!digits(&vt)) return false;
  if (i >= e || s[i] != '/') return false;
  ++i;
  if (!digits(nullptr)) return false;
  return i == e;
}

std::string directory_of(const std::string &p) {
  size_t k = p.find_last_of('/');
  return k == std::string::npos ? std::string() : p.substr(0, k + 1);
}

// text after the keyword with surrounding whitespace removed: "(.*?)\s*$"
bool rest_after_keyword(const std::string &line, const std::vector<std::pair<size_t, size_t>> &tk, std::string &out) {
  // `.` does not match '\r' or '\n' in an ECMAScript regex: such a line matches nothing
  if (tk.size() < 2) {
    out.clear();
    // "^\s*key\s+(.*?)\s*$" needs at least one whitespace after the keyword
    return tk.size() == 1 && tk[0].second < line.size();
  }
  const size_t b = tk[1].first, e = tk.back().second;
  for (size_t i = b; i < e; ++i)
    if (line[i] == '\r' || line[i] == '\n') return false;
  out.assign(line, b, e - b);
  return true;
}

bool parsemtl(const std::string &filename, std::vector<Material> &mats, std::vector<Image> &textures,
              std::map<std::string, int> &names, std::string &err) {
  std::ifstream in(filename);  // a missing file is silently empty (object.cpp:581)
  std::string line;
  std::vector<std::pair<size_t, size_t>> tk;
  int cur = -1;
  while (std::getline(in, line)) {
    tokens_of(line, tk);
    if (tk.empty()) continue;
    const std::string key(line, tk[0].first, tk[0].second - tk[0].first);
    if (key == "newmtl") {
      std::string name;
      if (!rest_after_keyword(line, tk, name)) continue;
      mats.push_back(Material());
      cur = (int)mats.size() - 1;
      names[name] = cur;
    } else if ((key == "Kd" || key == "Ke") && tk.size() == 4) {
      float f[3];
      bool ok = cur >= 0;
      for (int k = 0; k < 3 && ok; ++k)
        ok = number_syntax(line, tk[k + 1].first, tk[k + 1].second, false) &&
             to_float(line, tk[k + 1].first, tk[k + 1].second, f[k]);
      if (!ok) continue;
      (key == "Kd" ? mats[cur].diffuse : mats[cur].emission) = vec(f[0], f[1], f[2]);
    } else if (key == "map_Kd") {
      std::string name;
      if (cur >= 0 && rest_after_keyword(line, tk, name)) {  // object.cpp:576-579
        mats[cur].diffuse_map = directory_of(filename) + name;
        Image img;
        if (!load_image(mats[cur].diffuse_map, img, err)) return false;  // the reference aborts, object.cpp:40-52
        textures.push_back(std::move(img));
        mats[cur].texture = (int)textures.size() - 1;
      }
    }
  }
  return true;
}

}  // namespace

bool parseobj(const std::string &filename, ParsedScene &out, std::string &err) {
  const auto t_start = std::chrono::steady_clock::now();
  out = ParsedScene();
  out.materials.push_back(Material());
  std::ifstream in(filename);
  if (!in) {
    err = "cannot open " + filename;
    return false;
  }
  std::map<std::string, int> names;
  Polygon *cur = nullptr;
  Polygon pending;
  float offset_v = 0, offset_vt = 0;  // the reference keeps index offsets in a float vector (math.cpp:22, object.cpp:599)
  auto close_current = [&]() -> bool {
    if (!cur) return true;
    if (!cur->init(&err)) return false;
    if (out.materials[cur->material].texture >= 0)  // Triangle::vt0/vt1/vt2 assert this, object.cpp:542-553
      for (const auto &t : cur->tri_vt)
        for (uint32_t id : t)
          if (id >= cur->vtextures.size()) {
            err = filename + ": textured object with a face whose texture coordinate is missing or outside its own object";
            return false;
          }
    offset_v += (float)cur->vertices.size();
    offset_vt += (float)cur->vtextures.size();
    out.meshes.push_back(std::move(*cur));
    pending = Polygon();
    cur = nullptr;
    return true;
  };
  std::string line;
  std::vector<std::pair<size_t, size_t>> tk;
  size_t lineno = 0;
  auto at = [&](const char *what) {
    err = filename + ":" + std::to_string(lineno) + ": " + what;
    return false;
  };
  while (std::getline(in, line)) {
    ++lineno;
    tokens_of(line, tk);
    if (tk.empty()) continue;
    const size_t kb = tk[0].first, ke = tk[0].second, kl = ke - kb;
    const char *k = line.data() + kb;
    const bool is_v = kl == 1 && k[0] == 'v';
    const bool is_vn = kl == 2 && k[0] == 'v' && k[1] == 'n';
    const bool is_vt = kl == 2 && k[0] == 'v' && k[1] == 't';
    if ((is_v || is_vn) && tk.size() == 4) {
      bool syn = true;
      for (int c = 1; c < 4; ++c) syn = syn && number_syntax(line, tk[c].first, tk[c].second, true);
      if (syn) {
        if (!cur) return at("vertex data before the first 'o' line");
        float f[3];
        for (int c = 0; c < 3; ++c)
          if (!to_float(line, tk[c + 1].first, tk[c + 1].second, f[c])) return at("number that std::stof rejects");
        if (is_v) cur->vertices.push_back(vec(f[0], f[1], f[2]));
        continue;
      }
    }
    if (is_vt && tk.size() >= 2 && tk.size() <= 4) {
      bool syn = true;
      for (size_t c = 1; c < tk.size(); ++c) syn = syn && number_syntax(line, tk[c].first, tk[c].second, true);
      if (syn) {
        if (!cur) return at("vertex data before the first 'o' line");
        float f[3] = {0, 0, 0};  // missing components are 0, object.cpp:662-668
        for (size_t c = 1; c < tk.size(); ++c)
          if (!to_float(line, tk[c].first, tk[c].second, f[c - 1])) return at("number that std::stof rejects");
        cur->vtextures.push_back(vec(f[0], f[1], f[2]));
        continue;
      }
    }
    if (kl == 1 && k[0] == 'f' && tk.size() >= 2) {
      bool syn = true, three0 = false;
      std::vector<long> idx(tk.size() - 1), vti(tk.size() - 1, -1);
      for (size_t c = 1; c < tk.size() && syn; ++c) {
        bool three = false;
        syn = corner_index(line, tk[c].first, tk[c].second, three, idx[c - 1], vti[c - 1]);
        if (c == 1) three0 = three;
        else if (three != three0) syn = false;  // one line matches one of the two face patterns
      }
      if (syn) {
        if (!cur) return at("face before the first 'o' line");
        if (idx.size() < 3) return at("face with fewer than three corners");
        std::vector<uint32_t> corner(idx.size()), corner_vt(idx.size(), 0xffffffffu);
        for (size_t c = 0; c < idx.size(); ++c) {
          if (three0 && vti[c] >= 0) {  // the a//c form has no texture coordinate (veci(l[0], 0, l[1]), object.cpp:672)
            float t = (float)(int)vti[c];
            t -= 1;
            t -= offset_vt;
            if (t >= 0) corner_vt[c] = (uint32_t)t;
          }
          if (idx[c] < 0) return at("face index that std::stoi rejects");
          float v = (float)(int)idx[c];  // object.cpp:620,629-635
          v -= 1;
          v -= offset_v;
          if (!(v >= 0)) return at("face index below its own object's first vertex");
          corner[c] = (uint32_t)v;
        }
        for (size_t c = 1; c + 1 < corner.size(); ++c)  // fan about corner 0, object.cpp:478-490
        {
          cur->triangles.push_back({corner[0], corner[c], corner[c + 1]});
          cur->tri_vt.push_back({corner_vt[0], corner_vt[c], corner_vt[c + 1]});
        }
        continue;
      }
    }
    if (line.size() >= 2 && line[0] == 'o' && is_space(line[1])) {  // "^o\s(.*)$"
      bool dot_ok = true;
      for (size_t i = 2; i < line.size(); ++i)
        if (line[i] == '\r' || line[i] == '\n') dot_ok = false;
      if (dot_ok) {
        if (!close_current()) return false;
        cur = &pending;
        continue;
      }
    }
    if (kl == 6 && memcmp(k, "mtllib", 6) == 0) {
      std::string name;
      if (rest_after_keyword(line, tk, name)) {
        if (!parsemtl(directory_of(filename) + name, out.materials, out.textures, names, err)) return false;
        continue;
      }
    }
    if (kl == 6 && memcmp(k, "usemtl", 6) == 0) {
      std::string name;
      if (rest_after_keyword(line, tk, name)) {
        if (!cur) return at("usemtl before the first 'o' line");
        auto f = names.find(name);
        if (f == names.end()) return at("usemtl names a material no mtllib defined");
        cur->material = f->second;
        continue;
      }
    }
    // anything else is ignored, as an unmatched line is in the reference
  }
  if (!close_current()) return false;
  if (out.meshes.empty()) {
    err = filename + ": no objects";
    return false;
  }
  if (verbose) {
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::printf("took %.2f secs to parse %s\n", secs, filename.c_str());
  }
  return true;
}

}  // namespace pt

// pt_flatten.cpp — turns the reference-shaped two-level BVH of a PtSceneDesc into
// the single device tree the gfx950 kernels walk.
//
// What the reference's recursive BVH::intersect (object.cpp:265-285) returns is
// fully described by (a) the set of leaves it reaches — a leaf is reached when
// every INTERNAL ancestor's own box test passes; leaves themselves and whole
// meshes inside a top-level leaf are never box-tested — and (b) a tie rule among
// equal t: inside a leaf the leftmost minimum wins (std::ranges::min), between
// siblings the right child wins (`lhit.t < rhit.t ? lhit : rhit`).  Rule (b) is a
// total order over triangle positions, so every triangle carries its position in
// that order as `rank` and "larger rank wins a tie" reproduces it for any visiting
// order.  Rule (a) is kept by tagging each child slot with its kind (pt_layout.h).
//
// A reference leaf holds up to 8 triangles.  Here it becomes a small subtree of
// GROUP nodes over leaves of at most `leaf_size` triangles, split at the spatial
// median: the exact mode visits all of it (as the reference visits the whole leaf),
// the fast mode box-tests its way down and so tests a few triangles instead of 8.
#include "pt_flatten.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>

namespace ptl {
namespace {

struct Entry {
  float lo[3], hi[3];
  uint32_t ref;
  uint32_t kind;
  uint32_t depth;
};

const uint32_t kTopNodes = 2047;   // nodes numbered breadth-first from the root
const uint32_t kMaxRecursion = 4096;

struct Flattener {
  const PtSceneDesc &d;
  FlatScene &out;
  std::string &err;
  std::vector<WideNode> nodes;       // creation order; renumbered at the end
  std::vector<uint8_t> tri_seen;     // per global triangle
  std::vector<uint8_t> mesh_seen;
  std::vector<uint8_t> tlas_seen;
  std::vector<uint8_t> blas_seen;    // per global blas node
  uint32_t next_slot = 0;
  uint32_t leaf_size = 2;            // triangles per device leaf (1..8)
  bool failed = false;

  Flattener(const PtSceneDesc &dd, FlatScene &o, std::string &e) : d(dd), out(o), err(e) {}

  Entry fail(const char *msg) {
    if (!failed) err = msg;
    failed = true;
    Entry e;
    memset(&e, 0, sizeof(e));
    e.ref = REF_LEAF;
    e.kind = KIND_LEAF;
    return e;
  }

  static void set_box(Entry &e, const PtNode &n) {
    for (int k = 0; k < 3; ++k) {
      e.lo[k] = n.bmin[k];
      e.hi[k] = n.bmax[k];
    }
  }

  uint32_t make_node(const Entry &a, const Entry &b) {
    WideNode w;
    for (int k = 0; k < 3; ++k) {
      w.lo0[k] = a.lo[k]; w.hi0[k] = a.hi[k];
      w.lo1[k] = b.lo[k]; w.hi1[k] = b.hi[k];
    }
    w.ref0 = a.ref; w.ref1 = b.ref;
    w.kind0 = a.kind; w.kind1 = b.kind;
    nodes.push_back(w);
    return (uint32_t)nodes.size() - 1;
  }

  // A child slot that leads nowhere: a leaf holding one degenerate triangle (all zeros: its
  // determinant is 0, so Triangle::intersect's first test rejects it for every ray) behind a
  // box no ray enters.  This way the walk needs no "empty" case.
  Entry empty_entry() {
    Entry e;
    for (int k = 0; k < 3; ++k) {
      e.lo[k] = INFINITY;
      e.hi[k] = -INFINITY;
    }
    TriRec r;
    memset(&r, 0, sizeof(r));
    r.tri_id = 0xffffffffu;
    out.tris.push_back(r);
    out.tri_gate.push_back(NO_GATE);
    out.tri_levels.push_back(0);
    e.ref = REF_LEAF | next_slot;
    ++next_slot;
    e.kind = KIND_LEAF;
    e.depth = 0;
    return e;
  }

  // Boxes of leaves and groups are never tested by the reference; the fast traversal mode
  // uses them to skip work, so they are widened by 2^-11 of the largest coordinate plus a
  // small absolute margin: a triangle whose computed hit lies a few rounding errors outside
  // its exact box must not be lost.
  static void widen(Entry &e) {
    float m = 0;
    for (int k = 0; k < 3; ++k) {
      if (std::isfinite(e.lo[k])) m = std::max(m, std::fabs(e.lo[k]));
      if (std::isfinite(e.hi[k])) m = std::max(m, std::fabs(e.hi[k]));
    }
    const float d = m * (1.0f / 2048.0f) + 2.0e-5f;
    for (int k = 0; k < 3; ++k) {
      e.lo[k] -= d;
      e.hi[k] += d;
    }
  }

  // Balanced grouping nodes over es[lo, hi).
  Entry group(const std::vector<Entry> &es, size_t lo, size_t hi) {
    if (hi - lo == 1) return es[lo];
    size_t mid = lo + (hi - lo) / 2;
    Entry a = group(es, lo, mid), b = group(es, mid, hi);
    Entry g;
    for (int k = 0; k < 3; ++k) {
      g.lo[k] = std::min(a.lo[k], b.lo[k]);
      g.hi[k] = std::max(a.hi[k], b.hi[k]);
    }
    g.ref = make_node(a, b);
    g.kind = KIND_GROUP;
    g.depth = 1 + std::max(a.depth, b.depth);
    return g;  // union of already widened boxes
  }

  // Exact bounds of the triangles in slots [lo, hi), from the vertices as given.
  Entry slot_bounds(uint32_t lo, uint32_t hi) const {
    Entry e;
    for (int k = 0; k < 3; ++k) {
      e.lo[k] = INFINITY;
      e.hi[k] = -INFINITY;
    }
    for (uint32_t s = lo; s < hi; ++s) {
      const float *p = d.tri_verts + (size_t)out.tris[s].tri_id * 9;
      for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 3; ++k) {
          e.lo[k] = std::min(e.lo[k], p[3 * v + k]);   // NaN coordinates leave the box alone
          e.hi[k] = std::max(e.hi[k], p[3 * v + k]);
        }
    }
    return e;
  }

  // Device subtree for one reference leaf whose triangles sit in slots [lo, hi): records are
  // reordered in place (their ranks travel with them), sorted along the axis of largest centroid
  // extent and cut so the left part is a whole number of device leaves.
  Entry split_leaf(uint32_t lo, uint32_t hi) {
    const uint32_t n = hi - lo;
    if (n <= leaf_size) {
      Entry c = slot_bounds(lo, hi);
      c.ref = REF_LEAF | ((n - 1) << 28) | lo;
      c.kind = KIND_LEAF;
      c.depth = 0;
      widen(c);
      return c;
    }
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto centroid = [&](const TriRec &r, int k) {
      const float *p = d.tri_verts + (size_t)r.tri_id * 9;
      return p[k] + p[3 + k] + p[6 + k];
    };
    for (uint32_t s = lo; s < hi; ++s)
      for (int k = 0; k < 3; ++k) {
        const float c = centroid(out.tris[s], k);
        cmin[k] = std::min(cmin[k], c);
        cmax[k] = std::max(cmax[k], c);
      }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
      if (cmax[k] - cmin[k] > cmax[axis] - cmin[axis]) axis = k;
    std::stable_sort(out.tris.begin() + lo, out.tris.begin() + hi, [&](const TriRec &a, const TriRec &b) {
      return centroid(a, axis) < centroid(b, axis);
    });
    const uint32_t leaves = (n + leaf_size - 1) / leaf_size;
    const uint32_t mid = lo + std::min(n - 1, (leaves + 1) / 2 * leaf_size);
    Entry a = split_leaf(lo, mid), b = split_leaf(mid, hi);
    Entry g;
    for (int k = 0; k < 3; ++k) {
      g.lo[k] = std::min(a.lo[k], b.lo[k]);
      g.hi[k] = std::max(a.hi[k], b.hi[k]);
    }
    g.ref = make_node(a, b);
    g.kind = KIND_SPLIT;
    g.depth = 1 + std::max(a.depth, b.depth);
    return g;
  }

  // Registers the box of an internal node of the reference as a gate and returns its index.
  uint32_t add_gate(const PtNode &n) {
    GateBox g;
    for (int k = 0; k < 3; ++k) {
      g.lo[k] = n.bmin[k];
      g.hi[k] = n.bmax[k];
    }
    g.pad0 = g.pad1 = 0;
    out.gates.push_back(g);
    return (uint32_t)out.gates.size() - 1;
  }

  // The fast traversal tests one box per triangle, the lowest internal box above it (its gate), in place of
  // every internal box above it.  The two agree when each internal box lies inside the one above it and the
  // triangles inside the lowest (the reference's builder makes boxes the exact unions of what is below them,
  // object.cpp:296-308).  A description that breaks this is counted here and then walked exactly.
  void check_inside(const float lo[3], const float hi[3], uint32_t gate) {
    if (gate == NO_GATE) return;
    const GateBox &g = out.gates[gate];
    for (int k = 0; k < 3; ++k)
      if (lo[k] < g.lo[k] || hi[k] > g.hi[k]) {  // NaN planes compare false: they restrict no ray either
        ++out.n_unnested;
        return;
      }
  }

  void put_triangle(uint32_t mesh, uint32_t local, uint32_t levels, uint32_t gate) {
    const uint32_t gid = d.tri_offset[mesh] + local;
    const float *p = d.tri_verts + (size_t)gid * 9;
    for (int v = 0; v < 3; ++v) check_inside(p + 3 * v, p + 3 * v, gate);
    TriRec r;
    for (int k = 0; k < 3; ++k) {
      r.v0[k] = p[k];
      r.e1[k] = p[3 + k] - p[k];
      r.e2[k] = p[6 + k] - p[k];
    }
    for (int k = 0; k < 9; ++k)
      if (std::isfinite(p[k])) out.max_abs_coord = std::max(out.max_abs_coord, std::fabs(p[k]));
    r.tri_id = gid;
    r.mat_id = d.mesh_material[mesh];
    r.rank = next_slot;
    out.tris.push_back(r);
    out.tri_gate.push_back(gate);
    out.tri_levels.push_back((uint8_t)std::min<uint32_t>(levels, 255u));
    ++next_slot;
  }

  // `inner` counts the internal nodes of the reference above this node (HitData::hits, object.cpp:281-282);
  // `gate` is the lowest of them (NO_GATE: none).
  Entry mesh_node(uint32_t mesh, uint32_t local, uint32_t level, uint32_t inner, uint32_t gate) {
    const uint32_t nb = d.blas_node_offset[mesh], ne = d.blas_node_offset[mesh + 1];
    const uint32_t ntri = d.tri_offset[mesh + 1] - d.tri_offset[mesh];
    if (level > kMaxRecursion) return fail("mesh tree deeper than 4096 levels");
    if (local >= ne - nb) return fail("mesh node index out of range");
    if (blas_seen[nb + local]) return fail("mesh node referenced twice");
    blas_seen[nb + local] = 1;
    const PtNode &n = d.blas_nodes[nb + local];
    Entry e;
    set_box(e, n);
    if (n.count > 0) {
      if ((uint64_t)n.index_or_first + n.count > ntri) return fail("leaf range outside its mesh");
      const uint32_t first_slot = next_slot;
      for (uint32_t pos = n.count; pos-- > 0;) {  // leftmost position gets the largest slot
        const uint32_t local_tri = d.blas_order[d.tri_offset[mesh] + n.index_or_first + pos];
        if (local_tri >= ntri) return fail("blas_order entry out of range");
        if (tri_seen[d.tri_offset[mesh] + local_tri]) return fail("triangle referenced twice");
        tri_seen[d.tri_offset[mesh] + local_tri] = 1;
        put_triangle(mesh, local_tri, inner, gate);
      }
      if (next_slot > MAX_TRI_SLOTS) return fail("more than 2^28 triangles");
      return split_leaf(first_slot, next_slot);
    }
    check_inside(n.bmin, n.bmax, gate);
    const uint32_t own = add_gate(n);
    Entry l = mesh_node(mesh, n.index_or_first, level + 1, inner + 1, own);
    Entry r = mesh_node(mesh, n.index_or_first + 1, level + 1, inner + 1, own);
    if (failed) return e;
    e.ref = make_node(l, r);
    e.kind = KIND_INTERNAL;
    e.depth = 1 + std::max(l.depth, r.depth);
    return e;
  }

  Entry tlas_node(uint32_t i, uint32_t level, uint32_t inner, uint32_t gate) {
    if (level > kMaxRecursion) return fail("top-level tree deeper than 4096 levels");
    if (i >= d.n_tlas_nodes) return fail("top-level node index out of range");
    if (tlas_seen[i]) return fail("top-level node referenced twice");
    tlas_seen[i] = 1;
    const PtNode &n = d.tlas_nodes[i];
    Entry e;
    set_box(e, n);
    if (n.count > 0) {
      if ((uint64_t)n.index_or_first + n.count > d.n_meshes) return fail("top-level leaf range out of range");
      std::vector<Entry> es;
      for (uint32_t pos = n.count; pos-- > 0;) {
        const uint32_t mesh = d.tlas_order[n.index_or_first + pos];
        if (mesh >= d.n_meshes) return fail("tlas_order entry out of range");
        if (mesh_seen[mesh]) return fail("mesh referenced twice");
        mesh_seen[mesh] = 1;
        if (d.blas_node_offset[mesh + 1] <= d.blas_node_offset[mesh]) return fail("mesh without nodes");
        es.push_back(mesh_node(mesh, 0, level + 1, inner, gate));
        if (failed) return e;
      }
      return group(es, 0, es.size());
    }
    check_inside(n.bmin, n.bmax, gate);
    const uint32_t own = add_gate(n);
    Entry l = tlas_node(n.index_or_first, level + 1, inner + 1, own);
    Entry r = tlas_node(n.index_or_first + 1, level + 1, inner + 1, own);
    if (failed) return e;
    e.ref = make_node(l, r);
    e.kind = KIND_INTERNAL;
    e.depth = 1 + std::max(l.depth, r.depth);
    return e;
  }

  // Root first, the top of the tree breadth-first, the rest depth-first.
  void renumber(uint32_t root) {
    const uint32_t n = (uint32_t)nodes.size();
    std::vector<uint32_t> order;
    order.reserve(n);
    std::deque<uint32_t> q;
    q.push_back(root);
    auto kids = [&](uint32_t i, uint32_t k[2]) {
      int c = 0;
      const WideNode &w = nodes[i];
      if (!(w.ref0 & REF_LEAF)) k[c++] = w.ref0;
      if (!(w.ref1 & REF_LEAF)) k[c++] = w.ref1;
      return c;
    };
    while (!q.empty() && order.size() < kTopNodes) {
      uint32_t i = q.front();
      q.pop_front();
      order.push_back(i);
      uint32_t k[2];
      int c = kids(i, k);
      for (int j = 0; j < c; ++j) q.push_back(k[j]);
    }
    std::vector<uint32_t> st;
    for (uint32_t start : q) {
      st.push_back(start);
      while (!st.empty()) {
        uint32_t i = st.back();
        st.pop_back();
        order.push_back(i);
        uint32_t k[2];
        int c = kids(i, k);
        for (int j = c - 1; j >= 0; --j) st.push_back(k[j]);
      }
    }
    std::vector<uint32_t> newidx(n, 0);
    for (uint32_t j = 0; j < order.size(); ++j) newidx[order[j]] = j;
    out.nodes.resize(order.size());
    for (uint32_t j = 0; j < order.size(); ++j) {
      WideNode w = nodes[order[j]];
      if (!(w.ref0 & REF_LEAF)) w.ref0 = newidx[w.ref0];
      if (!(w.ref1 & REF_LEAF)) w.ref1 = newidx[w.ref1];
      out.nodes[j] = w;
    }
  }

  // Scene::lights() in top-level object order with the reference's areas.
  void build_lights() {
    for (uint32_t pos = 0; pos < d.n_meshes; ++pos) {
      const uint32_t mesh = d.tlas_order[pos];
      const PtMaterial &m = d.materials[d.mesh_material[mesh]];
      if (m.ke[0] == 0 && m.ke[1] == 0 && m.ke[2] == 0) continue;  // emission != vec(0,0,0), scene.cpp:25
      LightRec L;
      L.first = (uint32_t)out.light_tris.size();
      L.count = d.tri_offset[mesh + 1] - d.tri_offset[mesh];
      double area = 0;
      for (uint32_t t = d.tri_offset[mesh]; t < d.tri_offset[mesh + 1]; ++t) {
        const float *p = d.tri_verts + (size_t)t * 9;
        const float *a = p, *b = p + 3, *c = p + 6;
        // Triangle::area, object.cpp:395-404: float shoelace on x,y only, widened afterwards
        float s = std::fabs(a[0] * (b[1] - c[1]) + b[0] * (c[1] - a[1]) + c[0] * (a[1] - b[1])) / 2;
        area += (double)s;
        LightTri lt;
        for (int k = 0; k < 3; ++k) {
          lt.u[k] = b[k] - a[k];
          lt.v[k] = c[k] - a[k];
        }
        lt.n[0] = lt.u[1] * lt.v[2] - lt.u[2] * lt.v[1];
        lt.n[1] = lt.u[2] * lt.v[0] - lt.u[0] * lt.v[2];
        lt.n[2] = lt.u[0] * lt.v[1] - lt.u[1] * lt.v[0];
        lt.pad = 0;
        lt.accum = area;
        out.light_tris.push_back(lt);
      }
      L.area_f = (float)area;
      L.inv_area_f = (float)(1 / area);
      for (int k = 0; k < 3; ++k) L.ke[k] = m.ke[k];
      L.mesh_id = mesh;
      out.lights.push_back(L);
    }
  }

  int run() {
    if (!d.tlas_nodes || !d.tlas_order || !d.blas_node_offset || !d.blas_nodes || !d.tri_offset ||
        !d.blas_order || !d.tri_verts || !d.mesh_material || !d.materials) {
      err = "null array in PtSceneDesc";
      return PT_ERR_INVALID;
    }
    if (d.n_meshes == 0 || d.n_tris == 0 || d.n_tlas_nodes == 0 || d.n_materials == 0) {
      err = "empty scene";
      return PT_ERR_INVALID;
    }
    if (d.tri_offset[0] != 0 || d.tri_offset[d.n_meshes] != d.n_tris || d.blas_node_offset[0] != 0) {
      err = "offset arrays do not start at 0 / end at the totals";
      return PT_ERR_INVALID;
    }
    for (uint32_t m = 0; m < d.n_meshes; ++m) {
      if (d.tri_offset[m + 1] < d.tri_offset[m] || d.blas_node_offset[m + 1] < d.blas_node_offset[m]) {
        err = "offset arrays are not monotone";
        return PT_ERR_INVALID;
      }
      if (d.tri_offset[m + 1] == d.tri_offset[m]) {
        err = "mesh without triangles";
        return PT_ERR_INVALID;
      }
      if (d.mesh_material[m] >= d.n_materials) {
        err = "mesh_material out of range";
        return PT_ERR_INVALID;
      }
    }
    if (d.n_tris > MAX_TRI_SLOTS) {
      err = "more than 2^28 triangles";
      return PT_ERR_UNSUPPORTED;
    }
    tri_seen.assign(d.n_tris, 0);
    mesh_seen.assign(d.n_meshes, 0);
    tlas_seen.assign(d.n_tlas_nodes, 0);
    blas_seen.assign(d.blas_node_offset[d.n_meshes], 0);
    if (const char *e = std::getenv("PT_LEAF")) {  // tuning knob: triangles per device leaf
      const int v = std::atoi(e);
      if (v >= 1 && v <= 8) leaf_size = (uint32_t)v;
    }
    out.tris.reserve((size_t)d.n_tris + 1);
    Entry root = tlas_node(0, 0, 0, NO_GATE);
    if (failed) return PT_ERR_INVALID;
    if (next_slot != d.n_tris) {
      err = "tree does not reach every triangle";
      return PT_ERR_INVALID;
    }
    for (uint32_t m = 0; m < d.n_meshes; ++m)
      if (!mesh_seen[m]) {
        err = "tree does not reach every mesh";
        return PT_ERR_INVALID;
      }
    Entry none = empty_entry();
    uint32_t super = make_node(root, none);
    out.max_depth = root.depth + 1;
    renumber(super);
    out.mats.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; ++i) {
      MatRec &r = out.mats[i];
      for (int k = 0; k < 3; ++k) {
        r.kd[k] = d.materials[i].kd[k];
        r.ke[k] = d.materials[i].ke[k];
      }
      r.tex = 0;
      r.pad1 = 0;
    }
    if (d.n_textures > 0) {
      if (!d.textures || !d.material_texture) {
        err = "n_textures > 0 but textures or material_texture is null";
        return PT_ERR_INVALID;
      }
      bool used = false;
      for (uint32_t i = 0; i < d.n_materials; ++i) {
        const int32_t t = d.material_texture[i];
        if (t < -1 || t >= (int32_t)d.n_textures) {
          err = "material_texture out of range";
          return PT_ERR_INVALID;
        }
        out.mats[i].tex = (uint32_t)(t + 1);
        used = used || t >= 0;
      }
      if (used) {
        if (!d.tri_uv) {
          err = "a material has a diffuse map but tri_uv is null";
          return PT_ERR_INVALID;
        }
        out.textures.resize(d.n_textures);
        for (uint32_t t = 0; t < d.n_textures; ++t) {
          const PtTexture &x = d.textures[t];
          if (x.width == 0 || x.height == 0 || !x.rgb || (uint64_t)x.width * x.height > (1ull << 28)) {
            err = "empty or oversized texture";
            return PT_ERR_INVALID;
          }
          const size_t bytes = (size_t)x.width * x.height * 3;
          if (out.texels.size() + bytes > 0xffffffffull) {
            err = "more than 4 GB of texels";
            return PT_ERR_UNSUPPORTED;
          }
          out.textures[t].offset = (uint32_t)out.texels.size();
          out.textures[t].width = x.width;
          out.textures[t].height = x.height;
          out.textures[t].pad = 0;
          out.texels.insert(out.texels.end(), x.rgb, x.rgb + bytes);
        }
        // texture coordinates follow their triangle records (leaf splitting reordered those)
        out.tri_uv.assign(out.tris.size() * 6, 0.0f);
        for (size_t s = 0; s < out.tris.size(); ++s) {
          const uint32_t gid = out.tris[s].tri_id;
          if (gid == 0xffffffffu) continue;
          for (int k = 0; k < 6; ++k) out.tri_uv[s * 6 + k] = d.tri_uv[(size_t)gid * 6 + k];
        }
      }
    }
    build_lights();
    out.n_meshes = d.n_meshes;
    // the fast traversal's own tree; its leaf boxes are widened by 2^-shift of the largest coordinate + 2e-5
    // 2^-17: every parity and stress test (tens of millions of rays, fast against exact) still shows 0 mismatches
    // with 2^-24, i.e. with the absolute 2e-5 alone; the rate stops improving at 2^-17 (7 M triangles: 4.02 / 4.43
    // / 4.55 / 4.54 G rays/s at 2^-14 / 2^-16 / 2^-17 / 2^-18).  The pruning walks are used for ray origins within
    // 8 times the largest coordinate (pt_gpu.hip, near_limit): a margin of half the size covers half the distance.
    int shift = 17;
    if (const char *e = std::getenv("PT_MARGIN_SHIFT")) {  // tuning knob
      const int v = std::atoi(e);
      if (v >= 6 && v <= 24) shift = v;
    }
    out.margin_rel = std::ldexp(1.0f, -shift);
    build_fast_tree(d, out);
    return PT_OK;
  }
};

}  // namespace

// Invariants of the fast tree: every triangle record is referenced by exactly one leaf, every child box
// contains (with the margin) the boxes or triangles below it, references stay in range.
bool triangle_is_ill_conditioned(const float *p) {
  double e1[3], e2[3];
  for (int k = 0; k < 3; ++k) {
    e1[k] = (double)p[3 + k] - (double)p[k];
    e2[k] = (double)p[6 + k] - (double)p[k];
  }
  const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  const double a = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], b = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
  if (!std::isfinite(cc) || !std::isfinite(a) || !std::isfinite(b)) return true;
  static const double limit = [] {
    int shift = 9;  // sine of the angle between the edges at v0 below 2^-shift
    if (const char *e = std::getenv("PT_ILL_SHIFT")) {  // tuning knob; 99: only exactly parallel edges
      const int v = std::atoi(e);
      if (v >= 0 && v <= 100) shift = v;
    }
    return shift == 100 ? -1.0 : (shift == 99 ? 0.0 : std::ldexp(1.0, -2 * shift));  // 100: no triangle counts (for measurements)
  }();
  if (limit < 0) return false;
  return !(cc > a * b * limit);
}

int check_fast_tree(const PtSceneDesc &d, const FlatScene &f, std::string &err) {
  const uint32_t n_tris = (uint32_t)f.tris4.size();
  if (f.nodes4.empty() || n_tris == 0 || f.tri_id4.size() != n_tris) {
    err = "fast tree: empty";
    return PT_ERR_INVALID;
  }
  std::vector<uint8_t> seen(n_tris, 0), node_seen(f.nodes4.size(), 0);
  const float leaf_margin = f.margin_rel * f.max_abs_coord + 2.0e-5f;
  struct Item { uint32_t node; float lo[3], hi[3]; bool nocull; };
  std::vector<Item> todo;
  Item root;
  root.node = 0;
  root.nocull = true;  // the root is entered by every ray
  for (int k = 0; k < 3; ++k) { root.lo[k] = -INFINITY; root.hi[k] = INFINITY; }
  todo.push_back(root);
  while (!todo.empty()) {
    const Item it = todo.back();
    todo.pop_back();
    if (it.node >= f.nodes4.size() || node_seen[it.node]) { err = "fast tree: node index out of range or referenced twice"; return PT_ERR_INVALID; }
    node_seen[it.node] = 1;
    const Quad4Node &q = f.nodes4[it.node];
    for (int c = 0; c < 4; ++c) {
      const float lo[3] = {q.lox[c], q.loy[c], q.loz[c]}, hi[3] = {q.hix[c], q.hiy[c], q.hiz[c]};
      if (lo[0] > hi[0]) continue;  // unused slot
      for (int k = 0; k < 3; ++k)
        if (lo[k] < it.lo[k] - 1e-3f || hi[k] > it.hi[k] + 1e-3f) { err = "fast tree: child box outside its parent's"; return PT_ERR_INVALID; }
      const bool path_nocull = it.nocull && (q.ref[c] & REF_NOCULL) != 0;  // every reference down to here carries the flag
      if (q.ref[c] & REF_LEAF) {
        const uint32_t first = q.ref[c] & REF_FIRST_MASK, cnt = ((q.ref[c] >> 28) & 1u) + 1u;
        if (first + cnt > n_tris - 1) { err = "fast tree: leaf range out of range"; return PT_ERR_INVALID; }
        for (uint32_t t = first; t < first + cnt; ++t) {
          if (seen[t]) { err = "fast tree: triangle in two leaves"; return PT_ERR_INVALID; }
          seen[t] = 1;
          const float *p = d.tri_verts + (size_t)f.tri_id4[t] * 9;
          if (triangle_is_ill_conditioned(p) && !path_nocull) { err = "fast tree: an ill-conditioned triangle can be skipped by distance"; return PT_ERR_INVALID; }
          for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k)
              if (p[3 * v + k] < lo[k] || p[3 * v + k] > hi[k]) { err = "fast tree: triangle outside its leaf box"; return PT_ERR_INVALID; }  // a gate box holds its triangles too
          // a leaf that distance can skip keeps a margin around its triangles on every axis on which its box is
          // not exactly flat (a flat one is a flat gate's plane, in which the triangles lie)
          if (!path_nocull && !triangle_is_ill_conditioned(p))
            for (int k = 0; k < 3; ++k) {
              if (lo[k] == hi[k]) continue;
              float vmin = p[k], vmax = p[k];
              for (int v = 1; v < 3; ++v) { vmin = std::fmin(vmin, p[3 * v + k]); vmax = std::fmax(vmax, p[3 * v + k]); }
              if (!(lo[k] <= vmin - 0.5f * leaf_margin) || !(hi[k] >= vmax + 0.5f * leaf_margin)) { err = "fast tree: a leaf box plane without margin"; return PT_ERR_INVALID; }
            }
          if (f.tris4[t].gate != NO_GATE && f.tris4[t].gate >= f.gates.size()) { err = "fast tree: gate index out of range"; return PT_ERR_INVALID; }
        }
      } else {
        Item ch;
        ch.node = q.ref[c] & ~REF_NOCULL;
        ch.nocull = path_nocull;
        for (int k = 0; k < 3; ++k) { ch.lo[k] = lo[k]; ch.hi[k] = hi[k]; }
        todo.push_back(ch);
      }
    }
  }
  for (uint32_t t = 0; t + 1 < n_tris; ++t)
    if (!seen[t] && f.tri_id4[t] != 0xffffffffu) { err = "fast tree: triangle in no leaf"; return PT_ERR_INVALID; }  // (a record that fills a pair of the eight-child tree is in none)
  for (size_t i = 0; i < f.nodes4.size(); ++i)
    if (!node_seen[i]) { err = "fast tree: unreachable node"; return PT_ERR_INVALID; }
  return PT_OK;
}

double oct_plane(const Oct8Node &n, int axis, uint32_t index) {
  const uint32_t e = (n.exps >> (8 * axis)) & 0xffu;
  return e == 0 ? (double)n.base[axis] : (double)n.base[axis] + (double)index * std::ldexp(1.0, (int)e - 127);
}

// Invariants of the eight-child tree: every node reachable once, every triangle record in exactly one leaf
// (the records that fill a one-triangle leaf's pair aside), and every box on the way down to a leaf (each decoded
// on its own node's grid, so a child's may reach outside its parent's) around that leaf's triangles with the
// margin the fast tree's leaves keep: at least half of it on every axis, except where a plane is exactly flat
// (the node's exponent byte is 0) and the triangles lie in that very plane.  The test is made against the
// intersection of the boxes on the path.
int check_oct_tree(const PtSceneDesc &d, const FlatScene &f, std::string &err) {
  if (f.nodes8.empty()) return PT_OK;
  const uint32_t n_tris = (uint32_t)f.tris4.size();
  std::vector<uint8_t> seen(n_tris, 0), node_seen(f.nodes8.size(), 0);
  const double leaf_margin = (double)(f.margin_rel * f.max_abs_coord + 2.0e-5f);
  struct Item { uint32_t node; double lo[3], hi[3]; bool flat[3]; };
  std::vector<Item> todo;
  Item root;
  root.node = 0;
  for (int k = 0; k < 3; ++k) { root.lo[k] = -INFINITY; root.hi[k] = INFINITY; root.flat[k] = false; }
  todo.push_back(root);
  while (!todo.empty()) {
    const Item it = todo.back();
    todo.pop_back();
    if (it.node >= f.nodes8.size() || node_seen[it.node]) { err = "eight-child tree: node index out of range or referenced twice"; return PT_ERR_INVALID; }
    node_seen[it.node] = 1;
    const Oct8Node &q = f.nodes8[it.node];
    const uint8_t *qq[3] = {q.qx, q.qy, q.qz};
    const uint32_t imask = q.masks & 0xffu, lmask = (q.masks >> 8) & 0xffu;
    if ((imask & lmask) != 0 || (q.masks >> 16) != 0 || ((q.exps >> 24) & 0xffu) > 2) { err = "eight-child tree: malformed masks"; return PT_ERR_INVALID; }
    uint32_t n_int = 0, n_leaf = 0;
    for (int c = 0; c < 8; ++c) {
      if (!(((imask | lmask) >> c) & 1u)) continue;
      double lo[3], hi[3];
      bool flat[3];
      for (int k = 0; k < 3; ++k) {
        if (qq[k][c] > qq[k][8 + c]) { err = "eight-child tree: inverted box"; return PT_ERR_INVALID; }
        lo[k] = std::fmax(it.lo[k], oct_plane(q, k, qq[k][c]));
        hi[k] = std::fmin(it.hi[k], oct_plane(q, k, qq[k][8 + c]));
        flat[k] = it.flat[k] || ((q.exps >> (8 * k)) & 0xffu) == 0;
      }
      if ((lmask >> c) & 1u) {
        const uint32_t first = q.tri_base + 2u * n_leaf;
        ++n_leaf;
        if ((uint64_t)first + 2 > n_tris - 1) { err = "eight-child tree: leaf range out of range"; return PT_ERR_INVALID; }
        uint32_t real = 0;
        for (uint32_t t = first; t < first + 2; ++t) {
          if (seen[t]) { err = "eight-child tree: triangle record in two leaves"; return PT_ERR_INVALID; }
          seen[t] = 1;
          if (f.tri_id4[t] == 0xffffffffu) continue;  // fills the pair of a one-triangle leaf
          ++real;
          const float *p = d.tri_verts + (size_t)f.tri_id4[t] * 9;
          if (triangle_is_ill_conditioned(p)) { err = "eight-child tree: built over an ill-conditioned triangle"; return PT_ERR_INVALID; }
          for (int k = 0; k < 3; ++k) {
            double vmin = p[k], vmax = p[k];
            for (int v = 1; v < 3; ++v) { vmin = std::fmin(vmin, (double)p[3 * v + k]); vmax = std::fmax(vmax, (double)p[3 * v + k]); }
            if (flat[k]) {
              if (!(vmin == lo[k] && vmax == hi[k])) { err = "eight-child tree: a flat plane that is not its triangles' plane"; return PT_ERR_INVALID; }
            } else if (!(lo[k] <= vmin - 0.5 * leaf_margin) || !(hi[k] >= vmax + 0.5 * leaf_margin)) {
              err = "eight-child tree: a leaf box plane without margin";
              return PT_ERR_INVALID;
            }
          }
          if (f.tris4[t].gate != NO_GATE && f.tris4[t].gate >= f.gates.size()) { err = "eight-child tree: gate index out of range"; return PT_ERR_INVALID; }
        }
        if (real == 0 || f.tri_id4[first] == 0xffffffffu) { err = "eight-child tree: empty leaf"; return PT_ERR_INVALID; }
      } else {
        Item ch;
        ch.node = q.child_base + n_int;
        ++n_int;
        for (int k = 0; k < 3; ++k) { ch.lo[k] = lo[k]; ch.hi[k] = hi[k]; ch.flat[k] = flat[k]; }
        todo.push_back(ch);
      }
    }
  }
  for (uint32_t t = 0; t + 1 < n_tris; ++t)
    if (!seen[t]) { err = "eight-child tree: triangle record in no leaf"; return PT_ERR_INVALID; }
  for (size_t i = 0; i < f.nodes8.size(); ++i)
    if (!node_seen[i]) { err = "eight-child tree: unreachable node"; return PT_ERR_INVALID; }
  return PT_OK;
}

int flatten_scene(const PtSceneDesc &d, FlatScene &out, std::string &err) {
  out = FlatScene();
  Flattener f(d, out, err);
  return f.run();
}

}  // namespace ptl

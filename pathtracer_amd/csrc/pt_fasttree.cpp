// pt_fasttree.cpp — the fast traversal mode's own tree.
//
// The reference's recursive BVH::intersect (object.cpp:265-285) defines WHICH triangle a ray returns:
// the hit of smallest t (ties by position, carried as TriRec::rank) among the triangles of the leaves it
// reaches, and it reaches a leaf exactly when the ray passes the own-box test of every internal node
// above it.  The boxes of a tree are nested (a node's box is the exact min/max of its children's,
// object.cpp:296-308) and AABB::intersect (object.cpp:179-195) is monotone under nesting — a larger box
// gives t0' <= t0 and t1' >= t1 in the same rounded arithmetic — so passing the box of the LOWEST internal
// node above a leaf implies passing all the ones above it.  That lowest box is the triangle's "gate".
//
// Hence the fast mode is free to find candidate triangles with any structure it likes, as long as it
// (a) never misses a triangle whose computed hit could win and (b) accepts a hit only when the ray passes
// the triangle's gate, tested with the reference's arithmetic.  This file builds that structure: one tree
// over all triangles of the scene (no top-level / per-mesh split), binned surface-area heuristic, leaves
// of at most two triangles (the kernels test triangles in packed pairs), collapsed to four children per
// node.  Every box is widened by a margin, because a triangle's computed hit can lie a few rounding
// errors outside its exact bounds.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "pt_flatten.h"

namespace ptl {
namespace {

struct Box3 {
  float lo[3], hi[3];
  void clear() {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  }
  void grow(const Box3 &b) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  float area() const {
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    if (!(x >= 0 && y >= 0 && z >= 0)) return 0;
    return x * y + y * z + z * x;
  }
};

struct BinNode {
  Box3 box;
  uint32_t left = 0, right = 0;   // children (internal) ...
  uint32_t first = 0, count = 0;  // ... or a range of `order` (count > 0: leaf)
};

struct FastBuilder {
  const PtSceneDesc &d;
  FlatScene &out;
  uint32_t n = 0;                 // real triangles (records of out.tris without the degenerate one)
  std::vector<Box3> pb;           // per record of out.tris: exact bounds of its vertices
  std::vector<float> cen;         // 3 per record: box centre
  std::vector<uint32_t> order;    // permutation of records, leaves are ranges of it
  std::vector<BinNode> bn;

  FastBuilder(const PtSceneDesc &dd, FlatScene &o) : d(dd), out(o) {}

  static const int kBins = 32;

  Box3 range_box(uint32_t lo, uint32_t hi) const {
    Box3 b;
    b.clear();
    for (uint32_t i = lo; i < hi; ++i) b.grow(pb[order[i]]);
    return b;
  }

  // Splits order[lo, hi) and returns the first index of the right part.
  uint32_t split(uint32_t lo, uint32_t hi) {
    const uint32_t cnt = hi - lo;
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lo; i < hi; ++i)
      for (int k = 0; k < 3; ++k) {
        const float c = cen[3 * (size_t)order[i] + k];
        if (c < cmin[k]) cmin[k] = c;
        if (c > cmax[k]) cmax[k] = c;
      }
    if (cnt <= 16) {
      // few triangles: exact sweep along every axis; the left part gets an even number of triangles so
      // that leaves come out as pairs
      float best = INFINITY;
      int best_axis = -1;
      uint32_t best_k = 0;
      std::vector<uint32_t> tmp(order.begin() + lo, order.begin() + hi), best_order;
      std::vector<float> right_area(cnt + 1);
      for (int axis = 0; axis < 3; ++axis) {
        if (!(cmax[axis] > cmin[axis])) continue;
        std::stable_sort(tmp.begin(), tmp.end(), [&](uint32_t a, uint32_t b) { return cen[3 * (size_t)a + axis] < cen[3 * (size_t)b + axis]; });
        Box3 b;
        b.clear();
        for (uint32_t k = cnt; k-- > 0;) {
          b.grow(pb[tmp[k]]);
          right_area[k] = b.area();
        }
        b.clear();
        for (uint32_t k = 1; k < cnt; ++k) {
          b.grow(pb[tmp[k - 1]]);
          if (k & 1u) continue;
          const float cost = b.area() * (float)k + right_area[k] * (float)(cnt - k);
          if (cost < best) {
            best = cost;
            best_axis = axis;
            best_k = k;
            best_order = tmp;
          }
        }
      }
      if (best_axis < 0) {  // cnt == 3, or all centres equal
        uint32_t k = cnt >= 4 ? (cnt / 2) & ~1u : (cnt == 3 ? 2u : 1u);
        return lo + k;
      }
      std::copy(best_order.begin(), best_order.end(), order.begin() + lo);
      return lo + best_k;
    }
    float best = INFINITY;
    int best_axis = -1, best_bin = 0;
    for (int axis = 0; axis < 3; ++axis) {
      const float ext = cmax[axis] - cmin[axis];
      if (!(ext > 0) || !std::isfinite(ext)) continue;
      Box3 bb[kBins];
      uint32_t bc[kBins];
      for (int b = 0; b < kBins; ++b) { bb[b].clear(); bc[b] = 0; }
      const float scale = (float)kBins / ext;
      for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t id = order[i];
        int b = (int)((cen[3 * (size_t)id + axis] - cmin[axis]) * scale);
        b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
        bb[b].grow(pb[id]);
        ++bc[b];
      }
      float ra[kBins];
      uint32_t rc[kBins];
      Box3 acc;
      acc.clear();
      uint32_t c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        acc.grow(bb[b]);
        c += bc[b];
        ra[b] = acc.area();
        rc[b] = c;
      }
      acc.clear();
      c = 0;
      for (int b = 0; b + 1 < kBins; ++b) {
        acc.grow(bb[b]);
        c += bc[b];
        if (c == 0 || rc[b + 1] == 0) continue;
        const float cost = acc.area() * (float)c + ra[b + 1] * (float)rc[b + 1];
        if (cost < best) {
          best = cost;
          best_axis = axis;
          best_bin = b;
        }
      }
    }
    if (best_axis < 0) return lo + cnt / 2;  // all centres (nearly) equal, or non-finite: halves
    const float ext = cmax[best_axis] - cmin[best_axis];
    const float scale = (float)kBins / ext;
    auto mid = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t id) {
      int b = (int)((cen[3 * (size_t)id + best_axis] - cmin[best_axis]) * scale);
      b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
      return b <= best_bin;
    });
    uint32_t m = (uint32_t)(mid - order.begin());
    if (m == lo || m == hi) m = lo + cnt / 2;
    return m;
  }

  void build_binary() {
    bn.clear();
    bn.reserve((size_t)n + 16);
    struct Job { uint32_t node, lo, hi; };
    std::vector<Job> jobs;
    bn.push_back(BinNode());
    jobs.push_back(Job{0, 0, n});
    while (!jobs.empty()) {
      const Job j = jobs.back();
      jobs.pop_back();
      bn[j.node].box = range_box(j.lo, j.hi);
      if (j.hi - j.lo <= 2) {
        bn[j.node].first = j.lo;
        bn[j.node].count = j.hi - j.lo;
        continue;
      }
      const uint32_t m = split(j.lo, j.hi);
      const uint32_t l = (uint32_t)bn.size();
      bn.push_back(BinNode());
      bn.push_back(BinNode());
      bn[j.node].left = l;
      bn[j.node].right = l + 1;
      jobs.push_back(Job{l + 1, m, j.hi});
      jobs.push_back(Job{l, j.lo, m});
    }
  }

  // Collapses the binary tree into four-child nodes (creation order; renumbered afterwards) and lays
  // the triangle records out in leaf order.
  struct Wide {
    uint32_t child[8];
    int n_child;
  };
  std::vector<Wide> wide;          // child = index into bn
  std::vector<uint32_t> wide_of;   // per binary node that heads a wide node: its index in `wide`
  std::vector<Wide> wide8;         // the same collapse with up to eight children
  std::vector<uint32_t> wide_of8;

  void collapse(int width, std::vector<Wide> &wide, std::vector<uint32_t> &wide_of) {
    wide.clear();
    wide_of.assign(bn.size(), 0xffffffffu);
    std::vector<uint32_t> todo;
    todo.push_back(0);
    while (!todo.empty()) {
      const uint32_t b = todo.back();
      todo.pop_back();
      Wide w;
      w.n_child = 2;
      w.child[0] = bn[b].left;
      w.child[1] = bn[b].right;
      while (w.n_child < width) {
        int pick = -1;
        float best = -1;
        for (int k = 0; k < w.n_child; ++k) {
          const BinNode &c = bn[w.child[k]];
          if (c.count > 0) continue;
          const float a = c.box.area();
          if (a > best) {
            best = a;
            pick = k;
          }
        }
        if (pick < 0) break;
        const uint32_t c = w.child[pick];
        w.child[pick] = bn[c].left;
        w.child[w.n_child++] = bn[c].right;
      }
      wide_of[b] = (uint32_t)wide.size();
      wide.push_back(w);
      for (int k = 0; k < w.n_child; ++k)
        if (bn[w.child[k]].count == 0) todo.push_back(w.child[k]);
    }
  }

  // The same collapse by dynamic programming over the binary tree: the set of wide nodes that minimises the
  // summed surface area of the wide nodes, i.e. the expected number of node visits of a random ray, for the
  // given width (leaves stay what they are).  cost[n][i] is the least cost of the subtree of binary node n
  // cut into at most i pieces that hang side by side in one wide node; a piece is a leaf or a wide node of its
  // own (its area plus the best cut of its two halves into at most `width` pieces).  The greedy collapse fills the
  // upper node of a small subtree and leaves its children nearly empty (4 of 8 slots in use on average);
  // this one fills the nodes next to the leaves.
  void collapse_optimal(int width, std::vector<Wide> &wide, std::vector<uint32_t> &wide_of) {
    const int W = width;
    const size_t nb = bn.size();
    std::vector<float> cost(nb * (size_t)W, INFINITY);        // [n][i-1], i = 1..W-1 pieces (index W-1 unused)
    std::vector<uint8_t> cut(nb * (size_t)W, 0);              // [n][i-1]: pieces given to the left half (0: same as i-1 pieces)
    std::vector<uint8_t> cut_node(nb, 0);                     // pieces given to the left half when n heads a wide node
    auto C = [&](uint32_t n, int i) -> float & { return cost[(size_t)n * W + (i - 1)]; };
    for (size_t n = nb; n-- > 0;) {  // children were created after their parent
      const BinNode &b = bn[n];
      if (b.count > 0) {
        for (int i = 1; i < W; ++i) C((uint32_t)n, i) = 0.0f;  // a leaf costs the same wherever it hangs
        continue;
      }
      const uint32_t l = b.left, r = b.right;
      // as a wide node of its own: its area + the best cut of the halves into at most W pieces
      float best = INFINITY;
      int best_k = 1;
      for (int k = 1; k < W; ++k) {
        const float c = C(l, k) + C(r, W - k);
        if (c < best) { best = c; best_k = k; }
      }
      cut_node[n] = (uint8_t)best_k;
      C((uint32_t)n, 1) = best + b.box.area();
      for (int i = 2; i < W; ++i) {
        float bi = C((uint32_t)n, i - 1);
        int bk = 0;
        for (int k = 1; k < i; ++k) {
          const float c = C(l, k) + C(r, i - k);
          if (c < bi) { bi = c; bk = k; }
        }
        C((uint32_t)n, i) = bi;
        cut[n * (size_t)W + (i - 1)] = (uint8_t)bk;
      }
    }
    wide.clear();
    wide_of.assign(nb, 0xffffffffu);
    std::vector<uint32_t> todo;
    todo.push_back(0);
    struct Part { uint32_t n; int i; };
    std::vector<Part> parts;
    while (!todo.empty()) {
      const uint32_t head = todo.back();
      todo.pop_back();
      Wide w;
      w.n_child = 0;
      parts.clear();
      parts.push_back(Part{bn[head].left, (int)cut_node[head]});
      parts.push_back(Part{bn[head].right, W - (int)cut_node[head]});
      while (!parts.empty()) {
        Part p = parts.back();
        parts.pop_back();
        if (bn[p.n].count > 0) { w.child[w.n_child++] = p.n; continue; }
        if (p.i >= W) p.i = W - 1;
        while (p.i > 1 && cut[(size_t)p.n * W + (p.i - 1)] == 0) --p.i;  // fewer pieces were as good
        if (p.i == 1) { w.child[w.n_child++] = p.n; continue; }    // hangs here as a wide node of its own
        const int k = cut[(size_t)p.n * W + (p.i - 1)];
        parts.push_back(Part{bn[p.n].right, p.i - k});
        parts.push_back(Part{bn[p.n].left, k});
      }
      wide_of[head] = (uint32_t)wide.size();
      wide.push_back(w);
      for (int k = 0; k < w.n_child; ++k)
        if (bn[w.child[k]].count == 0) todo.push_back(w.child[k]);
    }
  }

  // Boxes the kernels test: a leaf's exact bounds widened by the margin and then clipped to the bounds of its
  // triangles' gate boxes (a ray that misses the gate cannot produce an accepted hit, and the clipped planes
  // are the reference's own, tested with the reference's arithmetic: a flat wall keeps its flat box and a ray
  // leaving that wall does not enter it); an inner node's box is the union of its children's.
  std::vector<Box3> tb;
  std::vector<Box3> tbw;     // the same without the clipping (every plane keeps its margin)
  std::vector<uint8_t> ill;  // per binary node: an ill-conditioned triangle below it
  void test_boxes() {
    const float margin = out.margin_rel * out.max_abs_coord + 2.0e-5f;
    tb.resize(bn.size());
    tbw.resize(bn.size());
    ill.assign(bn.size(), 0);
    out.n_ill = 0;
    for (size_t i = bn.size(); i-- > 0;) {  // children were created after their parent
      const BinNode &b = bn[i];
      if (b.count == 0) {
        tb[i] = tb[b.left];
        tb[i].grow(tb[b.right]);
        tbw[i] = tbw[b.left];
        tbw[i].grow(tbw[b.right]);
        ill[i] = ill[b.left] | ill[b.right];
        continue;
      }
      Box3 w = b.box;
      for (int k = 0; k < 3; ++k) { w.lo[k] -= margin; w.hi[k] += margin; }
      for (uint32_t t = 0; t < b.count; ++t)
        if (triangle_is_ill_conditioned(d.tri_verts + (size_t)out.tris[order[b.first + t]].tri_id * 9)) {
          // its hits are not bounded by its vertices: nothing tighter than the gate (below)
          for (int k = 0; k < 3; ++k) { w.lo[k] = -INFINITY; w.hi[k] = INFINITY; }
          ill[i] = 1;
          ++out.n_ill;
        }
      tbw[i] = w;
      Box3 g;
      g.clear();
      bool all = true;
      for (uint32_t t = 0; t < b.count; ++t) {
        const uint32_t gate = out.tri_gate[order[b.first + t]];
        if (gate == NO_GATE) { all = false; break; }
        Box3 gb;
        for (int k = 0; k < 3; ++k) { gb.lo[k] = out.gates[gate].lo[k]; gb.hi[k] = out.gates[gate].hi[k]; }
        g.grow(gb);
      }
      if (all)
        for (int k = 0; k < 3; ++k) {
          // Clipped planes have no margin, and a plane without margin must not decide a skip by distance: a
          // computed hit point can lie a rounding error outside it, and for a ray nearly parallel to the plane
          // that error is a large difference between the box's entry distance and the hit's t.  Two cases are
          // safe.  A gate that is flat on this axis: its triangles lie in that very plane, so entry distance
          // and t are the same quantity (this is the clipping that pays: a ray leaving a wall does not enter
          // the wall's box).  A leaf that is never skipped by distance (ill-conditioned, REF_NOCULL all the way
          // down): only the pass/miss decision uses its planes, and a ray that passes the gate passes them.
          if (!ill[i] && !(g.lo[k] == g.hi[k])) continue;
          if (g.lo[k] > w.lo[k]) w.lo[k] = g.lo[k];   // NaN planes leave the widened box alone
          if (g.hi[k] < w.hi[k]) w.hi[k] = g.hi[k];
        }
      tb[i] = w;
    }
  }

  // Order of the wide nodes in memory: the top of the tree breadth-first (those are the ones kept in LDS),
  // the rest depth-first.
  std::vector<uint32_t> sequence(const std::vector<Wide> &wide, const std::vector<uint32_t> &wide_of, uint32_t kTop) const {
    std::vector<uint32_t> seq;
    seq.reserve(wide.size());
    std::deque<uint32_t> q;
    q.push_back(0);
    auto kids = [&](uint32_t wi, uint32_t k[8]) {
      int c = 0;
      for (int j = 0; j < wide[wi].n_child; ++j)
        if (bn[wide[wi].child[j]].count == 0) k[c++] = wide_of[wide[wi].child[j]];
      return c;
    };
    while (!q.empty() && seq.size() < kTop) {
      const uint32_t wi = q.front();
      q.pop_front();
      seq.push_back(wi);
      uint32_t k[8];
      const int c = kids(wi, k);
      for (int j = 0; j < c; ++j) q.push_back(k[j]);
    }
    std::vector<uint32_t> st;
    for (uint32_t start : q) {
      st.push_back(start);
      while (!st.empty()) {
        const uint32_t wi = st.back();
        st.pop_back();
        seq.push_back(wi);
        uint32_t k[8];
        const int c = kids(wi, k);
        for (int j = c - 1; j >= 0; --j) st.push_back(k[j]);
      }
    }
    return seq;
  }

  std::vector<uint32_t> leaf_first;  // per binary leaf: its first record in out.tris4
  bool oct_layout = false;           // the eight-child tree has laid the triangle records out (emit8 ran first)

  void put_record(uint32_t rec) {
    const TriRec &r = out.tris[rec];
    TriRec4 q;
    for (int k = 0; k < 3; ++k) { q.v0[k] = r.v0[k]; q.e1[k] = r.e1[k]; q.e2[k] = r.e2[k]; }
    q.gate = out.tri_gate[rec];
    q.mat_id = r.mat_id;
    q.rank = r.rank;
    out.tris4.push_back(q);
    out.tri_id4.push_back(r.tri_id);
  }

  void emit() {
    const std::vector<uint32_t> seq = sequence(wide, wide_of, 1280);
    std::vector<uint32_t> newidx(wide.size(), 0);
    for (uint32_t j = 0; j < seq.size(); ++j) newidx[seq[j]] = j;

    if (!oct_layout) {
      // triangle records in the order the leaves are met
      out.tris4.clear();
      out.tri_id4.clear();
      out.tris4.reserve((size_t)n + 1);
      out.tri_id4.reserve((size_t)n + 1);
      leaf_first.assign(bn.size(), 0);
      for (uint32_t j = 0; j < seq.size(); ++j) {
        const Wide &w = wide[seq[j]];
        for (int k = 0; k < w.n_child; ++k) {
          const BinNode &c = bn[w.child[k]];
          if (c.count == 0) continue;
          leaf_first[w.child[k]] = (uint32_t)out.tris4.size();
          for (uint32_t t = 0; t < c.count; ++t) put_record(order[c.first + t]);
        }
      }
      put_record(n);  // the degenerate triangle no ray hits
    }
    const uint32_t filler = (uint32_t)out.tris4.size() - 1;

    out.nodes4.assign(seq.size(), Quad4Node());
    for (uint32_t j = 0; j < seq.size(); ++j) {
      const Wide &w = wide[seq[j]];
      Quad4Node &o = out.nodes4[j];
      memset(&o, 0, sizeof(o));
      for (int k = 0; k < 4; ++k) {
        if (k >= w.n_child) {
          o.lox[k] = o.loy[k] = o.loz[k] = INFINITY;
          o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
          o.ref[k] = REF_LEAF | filler;
          continue;
        }
        const BinNode &c = bn[w.child[k]];
        const Box3 &cb = tb[w.child[k]];
        o.lox[k] = cb.lo[0]; o.hix[k] = cb.hi[0];
        o.loy[k] = cb.lo[1]; o.hiy[k] = cb.hi[1];
        o.loz[k] = cb.lo[2]; o.hiz[k] = cb.hi[2];
        if (c.count > 0) o.ref[k] = REF_LEAF | ((c.count - 1) << 28) | leaf_first[w.child[k]];
        else o.ref[k] = newidx[wide_of[w.child[k]]];
        if (ill[w.child[k]]) o.ref[k] |= REF_NOCULL;
      }
    }

    // depth of the wide tree -> stack entries a descent can need (up to three siblings left behind per level)
    uint32_t deepest = 1;
    {
      std::vector<std::pair<uint32_t, uint32_t>> walk;  // node, depth
      walk.push_back(std::make_pair(0u, 1u));
      while (!walk.empty()) {
        const std::pair<uint32_t, uint32_t> e = walk.back();
        walk.pop_back();
        if (e.second > deepest) deepest = e.second;
        const Quad4Node &o = out.nodes4[e.first];
        for (int k = 0; k < 4; ++k)
          if (!(o.ref[k] & REF_LEAF)) walk.push_back(std::make_pair(o.ref[k] & ~REF_NOCULL, e.second + 1));
      }
    }
    out.max_depth4 = 3 * deepest + 2;

    if (!out.tri_uv.empty()) {
      out.tri_uv4.assign(out.tris4.size() * 6, 0.0f);
      for (size_t s = 0; s < out.tris4.size(); ++s) {
        const uint32_t gid = out.tri_id4[s];
        if (gid == 0xffffffffu) continue;
        for (int k = 0; k < 6; ++k) out.tri_uv4[s * 6 + k] = d.tri_uv[(size_t)gid * 6 + k];
      }
    }
  }

  // The eight-child form of the same binary tree (pt_layout.h, Oct8Node).  It names children by position, so it
  // decides where nodes and triangle records go: a node's internal children get consecutive node indices, its
  // leaf children consecutive pairs of triangle records (the four-child tree then points into the same records).
  // Built only for a scene that will not sit in LDS as a whole, and only when every box is finite and no
  // triangle is ill-conditioned (boxes bounded by a gate alone, or by nothing, do not go on a grid).
  void emit8() {
    out.nodes8.clear();
    out.max_depth8 = 0;
    oct_layout = false;
    if (out.n_ill != 0) return;
    const size_t lds_bytes = 112 * wide.size() + 48 * ((size_t)n + 1) + 32 * out.gates.size();
    if (lds_bytes <= 88 * 1024 && !std::getenv("PT_OCT_ALWAYS")) return;
    for (const Box3 &b : tb)
      for (int k = 0; k < 3; ++k)
        if (!(b.lo[k] <= b.hi[k]) || !std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) return;
    for (const Box3 &b : tbw)
      for (int k = 0; k < 3; ++k)
        if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) return;
    if (wide8.size() > 0x3fffffffu) return;
    oct_layout = true;
    out.tris4.clear();
    out.tri_id4.clear();
    out.tris4.reserve((size_t)n + n / 4 + 2);
    out.tri_id4.reserve((size_t)n + n / 4 + 2);
    leaf_first.assign(bn.size(), 0);
    out.nodes8.assign(wide8.size(), Oct8Node());
    // nodes are numbered as their parents are laid out: the top of the tree breadth-first (kept in LDS), the rest
    // depth-first, so that a node's children follow it closely
    std::vector<uint32_t> wide_at(wide8.size(), 0);   // nodes8 index -> wide8 index
    std::vector<uint32_t> depth_of(wide8.size(), 1);
    std::deque<uint32_t> work;                         // nodes8 indices whose children are not laid out yet
    uint32_t next_node = 1, deepest = 1;
    work.push_back(0);
    while (!work.empty()) {
      uint32_t j;
      if (next_node < 600) { j = work.front(); work.pop_front(); }
      else { j = work.back(); work.pop_back(); }
      Wide w = wide8[wide_at[j]];
      Oct8Node &o = out.nodes8[j];
      memset(&o, 0, sizeof(o));
      // node box and the axis its children are sorted on
      Box3 nb;
      nb.clear();
      for (int k = 0; k < w.n_child; ++k) nb.grow(tb[w.child[k]]);
      int axis = 0;
      for (int a = 1; a < 3; ++a)
        if (nb.hi[a] - nb.lo[a] > nb.hi[axis] - nb.lo[axis]) axis = a;
      std::stable_sort(w.child, w.child + w.n_child, [&](uint32_t x, uint32_t y) {
        return tb[x].lo[axis] + tb[x].hi[axis] < tb[y].lo[axis] + tb[y].hi[axis];
      });
      o.exps = (uint32_t)axis << 24;
      uint8_t *q[3] = {o.qx, o.qy, o.qz};
      for (int a = 0; a < 3; ++a) {
        // exactly flat node: every child is flat on this axis at one and the same plane
        bool flat = true;
        const float v = tb[w.child[0]].lo[a];
        for (int k = 0; k < w.n_child; ++k) {
          const Box3 &c = tb[w.child[k]];
          if (!(c.lo[a] == c.hi[a] && c.lo[a] == v)) flat = false;
        }
        if (flat) {
          o.base[a] = v;
          continue;  // exponent byte 0, every index 0
        }
        // a child that is flat here keeps its margin instead (its exact plane is not on the grid)
        double lo[8], hi[8], nlo = INFINITY, nhi = -INFINITY;
        for (int k = 0; k < w.n_child; ++k) {
          const Box3 &c = tb[w.child[k]], &cw = tbw[w.child[k]];
          const bool child_flat = c.lo[a] == c.hi[a];
          lo[k] = child_flat ? std::min(cw.lo[a], c.lo[a]) : c.lo[a];
          hi[k] = child_flat ? std::max(cw.hi[a], c.hi[a]) : c.hi[a];
          nlo = std::min(nlo, lo[k]);
          nhi = std::max(nhi, hi[k]);
        }
        const float base = (float)nlo;  // exact: nlo is one of the float planes
        o.base[a] = base;
        int e = -126;
        while (e < 127 && (double)base + 254.0 * std::ldexp(1.0, e) < nhi) ++e;
        const double step = std::ldexp(1.0, e);
        o.exps |= (uint32_t)(e + 127) << (8 * a);
        for (int k = 0; k < w.n_child; ++k) {
          double ql = std::floor((lo[k] - (double)base) / step), qh = std::ceil((hi[k] - (double)base) / step);
          while (ql > 0 && (double)base + ql * step > lo[k]) ql -= 1;
          while ((double)base + qh * step < hi[k]) qh += 1;
          if (ql < 0) ql = 0;
          if (qh > 255) qh = 255;  // cannot happen: 254 steps reach the node's far plane
          q[a][k] = (uint8_t)ql;
          q[a][8 + k] = (uint8_t)qh;
        }
      }
      o.child_base = next_node;
      o.tri_base = (uint32_t)out.tris4.size();
      uint32_t imask = 0, lmask = 0;
      for (int k = 0; k < w.n_child; ++k) {
        const BinNode &c = bn[w.child[k]];
        if (c.count > 0) {
          lmask |= 1u << k;
          leaf_first[w.child[k]] = (uint32_t)out.tris4.size();
          for (uint32_t t = 0; t < c.count; ++t) put_record(order[c.first + t]);
          if (c.count == 1) put_record(n);  // a record no ray hits fills the pair
        } else {
          imask |= 1u << k;
          wide_at[next_node] = wide_of8[w.child[k]];
          depth_of[next_node] = depth_of[j] + 1;
          if (depth_of[next_node] > deepest) deepest = depth_of[next_node];
          ++next_node;
        }
      }
      o.masks = imask | (lmask << 8);
      // the first child is laid out next in the depth-first part (the work list is then taken from its end)
      if (next_node < 600) for (uint32_t c = o.child_base; c < next_node; ++c) work.push_back(c);
      else for (uint32_t c = next_node; c-- > o.child_base;) work.push_back(c);
    }
    put_record(n);  // the degenerate triangle no ray hits, at the end as in the other layout
    out.max_depth8 = 2 * (deepest + 1);  // one entry of two words may wait per level
  }

  void run() {
    n = (uint32_t)out.tris.size() - 1;  // the last record is the degenerate filler
    pb.resize(n);
    cen.resize(3 * (size_t)n);
    order.resize(n);
    for (uint32_t s = 0; s < n; ++s) {
      const float *p = d.tri_verts + (size_t)out.tris[s].tri_id * 9;
      Box3 b;
      b.clear();
      for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 3; ++k) {
          const float x = p[3 * v + k];
          if (x < b.lo[k]) b.lo[k] = x;   // NaN coordinates leave the box alone
          if (x > b.hi[k]) b.hi[k] = x;
        }
      pb[s] = b;
      for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * b.lo[k] + 0.5f * b.hi[k];
        cen[3 * (size_t)s + k] = std::isfinite(c) ? c : 0.0f;
      }
      order[s] = s;
    }
    build_binary();
    if (bn[0].count > 0) {
      // one or two triangles in the whole scene: a root whose only child is that leaf
      const uint32_t leaf = (uint32_t)bn.size();
      bn.push_back(bn[0]);
      bn[0].count = 0;
      bn[0].left = bn[0].right = leaf;
      wide.clear();
      wide_of.assign(bn.size(), 0xffffffffu);
      Wide w;
      w.n_child = 1;
      w.child[0] = leaf;
      wide_of[0] = 0;
      wide.push_back(w);
      wide8 = wide;
      wide_of8 = wide_of;
    } else {
      // measured: the room 17.38 -> 17.63 G rays/s (27.8 -> 27.0 box tests per ray), 7 M triangles 4.02 -> 4.26
      if (std::getenv("PT_COLLAPSE_GREEDY")) collapse(4, wide, wide_of);  // tuning knob: the simple collapse, for comparison
      else collapse_optimal(4, wide, wide_of);
      if (std::getenv("PT_COLLAPSE_GREEDY")) collapse(8, wide8, wide_of8);  // tuning knob: the simple collapse, for comparison
      else collapse_optimal(8, wide8, wide_of8);
    }
    test_boxes();
    emit8();
    emit();
  }
};

}  // namespace

void build_fast_tree(const PtSceneDesc &d, FlatScene &out) {
  FastBuilder b(d, out);
  b.run();
}

}  // namespace ptl

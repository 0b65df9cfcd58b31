// pt_device.h — device-side building blocks for gfx950 (wave64): vector math with
// one rounding per operation, the counter-based random generator, the box and
// triangle tests and the stack-based tree walk.  Compiled with -ffp-contract=off:
// the reference's Release build has no FMA (no -march, README.md:9-14), and bit-equal
// hits need the same roundings.
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptgpu.h"
#include "pt_layout.h"

namespace ptd {

using ptl::KIND_GROUP;
using ptl::KIND_INTERNAL;
using ptl::KIND_LEAF;
using ptl::KIND_SPLIT;

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // math.cpp:108
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                // math.cpp:124-126
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  // math.cpp:109
__device__ __forceinline__ V3 normalize(V3 a) {                                                    // math.cpp:111-114
  float n = norm(a);
  return n == 0.0f ? a : mk(a.x / n, a.y / n, a.z / n);
}

// ------------------------------------------------------------------ Philox-4x32-10
__host__ __device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draw number `dim` of path (pixel, sample): a 53-bit uniform double in [0,1).
// Two draws share one Philox block; draws are consumed in increasing order, so the
// last block is kept.
struct PathRng {
  uint32_t k0, k1, pixel, sample;
  uint32_t block;
  uint32_t w[4];
  __device__ __forceinline__ void init(uint64_t seed, uint32_t px, uint32_t s) {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
    pixel = px;
    sample = s;
    block = 0xffffffffu;
  }
  __device__ __forceinline__ double u01(uint32_t dim) {
    const uint32_t b = dim >> 1;
    if (b != block) {
      philox4x32(pixel, sample, b, 0u, k0, k1, w);
      block = b;
    }
    const uint32_t hi = (dim & 1u) ? w[2] : w[0];
    const uint32_t lo = (dim & 1u) ? w[3] : w[1];
    const uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)m * (1.0 / 9007199254740992.0);
  }
  // randreal(x, y), math.cpp:13-18: uniform_real_distribution<double>, narrowed to float.
  __device__ __forceinline__ float real(float x, float y, uint32_t dim) {
    return (float)(u01(dim) * ((double)y - (double)x) + (double)x);
  }
};

// Russian roulette, camera.cpp:118-119: `randreal(0, 1) < rr_prob` with `constexpr double rr_prob = .7`.
// The float draw is widened and compared with the double, so a draw equal to 0.7f (just below .7) continues.
__host__ __device__ __forceinline__ bool roulette_pass(double rr_prob, float draw) { return (double)draw < rr_prob; }

// ------------------------------------------------------------------ tree walk
struct SceneDev {
  const ptl::WideNode *nodes;
  const ptl::TriRec *tris;
  const ptl::MatRec *mats;
  const ptl::LightRec *lights;
  const ptl::LightTri *light_tris;
  const float *tri_uv;            // 6 floats per triangle record, or null (no textured material)
  const ptl::TexRec *textures;
  const uint8_t *texels;
  uint32_t n_tex;                 // 0 when no material has a diffuse map
  // Where the interval of a box test starts for boxes the reference never tests, set per launch.  A leaf box
  // reaches at least 2e-5 beyond its triangles on every axis on which it is not exactly flat, so a ray with
  // |d| <= 2 is still inside at t = 1e-5 wherever a hit lies in it, and the reference's EPSILON
  // (object.cpp:180) serves (a flat axis is a flat gate's plane, which the reference tests with that EPSILON
  // itself): launches whose rays are all that short (bounce and visibility rays are normalised) use 1e-5.  A
  // longer direction vector can leave the box before t = 1e-5 with a hit at t >= 1e-6 inside: launches that
  // may hold such rays (callers' own rays, camera rays of a wide viewport) use PT_OWN_BOX_TMIN.
  float tmin_own;
  uint32_t n_nodes, n_tris, n_lights, n_mats;
  // the fast mode's own tree (pt_fasttree.cpp): four-child nodes, triangle records in its leaf order,
  // the global triangle id and texture coordinates per such record, the gate boxes
  const ptl::Quad4Node *nodes4;
  const ptl::Oct8Node *nodes8;    // the same tree with eight children per node (scenes outside LDS), or null
  const ptl::TriRec4 *tris4;
  const uint32_t *tri_id4;
  const ptl::GateBox *gates;
  const float *tri_uv4;
  uint32_t n_nodes4, n_tris4, n_gates, n_nodes8;
  uint32_t n_ill;                 // ill-conditioned triangles in the scene (REF_NOCULL references exist iff > 0)
  float margin;  // 2^-11 of the largest vertex coordinate plus 2e-5: how far outside a box counts as on it
  uint32_t quorum;      // lanes of a wave that must hold a parked leaf before triangles are tested (closest hit)
  uint32_t quorum_any;  // any-hit walks: the same in 64ths of the wave's unfinished lanes
  uint32_t vertex_major; // split shading walks its visibility rays vertex by vertex (all lights of a vertex side by side), not light by light
  uint32_t refill;  // idle lanes of a wave at which k_trace_queue retires finished rays and fetches new ones
  uint32_t refill_any;  // the same for visibility walks, whose hand-over is cheaper (a byte out, a ray in)
  uint32_t cohort;      // visibility walks of a scene with many lights: a wave takes whole vertices (see k_shadow_walk)
};

struct HitRec {
  float t;
  uint32_t slot;  // 0xffffffff = miss
  uint32_t rank;  // TriRec::rank of that triangle (tie order)
  float u, v;
  V3 n;           // Triangle::normal() of the hit triangle, un-normalised (only with WANT_N)
  uint32_t mat;   // its material id (only with WANT_N)
};

// Slack of the pruning of the fast mode (a relative 2^-12 plus a small absolute term).
// The reference compares nothing against a running best, so pruning must never drop a
// triangle whose computed t could still win.  (The boxes of leaves and groups, which the
// reference never tests, are widened by the flattener for the same reason.)
#define PT_SLACK_REL 2.44140625e-4f
#define PT_SLACK_ABS 2.0e-5f
// Where the interval of a box test starts for boxes the reference never tests (the fast tree's, and the
// widened leaf and group boxes of the reference-shaped tree) when the ray's direction vector is long: below
// the smallest t a triangle hit can have (1e-6, object.cpp:425), so that no box is left before a hit inside
// it can count.  See SceneDev::tmin_own.
#define PT_OWN_BOX_TMIN 5.0e-7f

// The fast mode over the eight-child form of its tree: what PT_TRACE_FAST runs as on a scene that does not fit
// LDS (chosen by the library, not by callers).
#define PT_TRACE_FAST_OCT 3
// Which tree a traversal mode walks, and whether it prunes.
__host__ __device__ constexpr bool mode_prunes(int mode) { return mode != PT_TRACE_EXACT; }
__host__ __device__ constexpr bool mode_wide(int mode) { return mode == PT_TRACE_FAST || mode == PT_TRACE_FAST_OCT; }
__host__ __device__ constexpr bool mode_oct(int mode) { return mode == PT_TRACE_FAST_OCT; }

// Möller–Trumbore exactly as Triangle::intersect evaluates it (object.cpp:406-429).
// Returns true when the reference would return a hit.
__device__ __forceinline__ bool tri_test(float4 a, float4 b, float4 c, V3 o, V3 d, float &t, float &u, float &v) {
  const V3 v0 = mk(a.x, a.y, a.z);
  const V3 e1 = mk(a.w, b.x, b.y);
  const V3 e2 = mk(b.z, b.w, c.x);
  const V3 pv = cross(d, e2);
  const float n = dot(pv, e1);
  if (n == 0.0f) return false;
  const V3 tv = sub(o, v0);
  const float inv = 1.0f / n;
  u = inv * dot(pv, tv);
  if (u < 0.0f || u > 1.0f) return false;
  const V3 q = scale(cross(tv, e1), inv);
  v = dot(q, d);
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(q, e2);
  if ((double)t < 1e-6) return false;  // object.cpp:426 compares in double
  return true;
}

// Two triangles at once, one in each half of a packed-float pair.  gfx950 executes
// v_pk_mul_f32 / v_pk_add_f32 (two IEEE single operations per lane) in the time of one
// scalar-per-lane operation, and these kernels are bound by vector-instruction issue, so the
// multiplies and adds of Möller–Trumbore are written on float pairs.  Each element still gets
// exactly the operations and roundings of tri_test (no FMA: -ffp-contract=off).  Branch-free:
// in a divergent wave an early return saves nothing unless all 64 lanes take it.
typedef float pt_f2 __attribute__((ext_vector_type(2)));
struct V3x2 {
  pt_f2 x, y, z;
};
__device__ __forceinline__ pt_f2 splat2(float a) { pt_f2 r = {a, a}; return r; }
__device__ __forceinline__ pt_f2 pair2(float a, float b) { pt_f2 r = {a, b}; return r; }
__device__ __forceinline__ V3x2 cross2(V3x2 a, V3x2 b) {
  V3x2 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ pt_f2 dot2(V3x2 a, V3x2 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ void tri_test2(float4 a0, float4 b0, float4 c0, float4 a1, float4 b1, float4 c1, V3 o, V3 d,
                                          bool &hit0, bool &hit1, pt_f2 &t, pt_f2 &u, pt_f2 &v) {
  V3x2 v0, e1, e2, dd, oo;
  v0.x = pair2(a0.x, a1.x); v0.y = pair2(a0.y, a1.y); v0.z = pair2(a0.z, a1.z);
  e1.x = pair2(a0.w, a1.w); e1.y = pair2(b0.x, b1.x); e1.z = pair2(b0.y, b1.y);
  e2.x = pair2(b0.z, b1.z); e2.y = pair2(b0.w, b1.w); e2.z = pair2(c0.x, c1.x);
  dd.x = splat2(d.x); dd.y = splat2(d.y); dd.z = splat2(d.z);
  oo.x = splat2(o.x); oo.y = splat2(o.y); oo.z = splat2(o.z);
  const V3x2 pv = cross2(dd, e2);
  const pt_f2 n = dot2(pv, e1);
  V3x2 tv;
  tv.x = oo.x - v0.x; tv.y = oo.y - v0.y; tv.z = oo.z - v0.z;
  const pt_f2 inv = pair2(1.0f / n.x, 1.0f / n.y);
  u = inv * dot2(pv, tv);
  const V3x2 c = cross2(tv, e1);
  V3x2 q;
  q.x = c.x * inv; q.y = c.y * inv; q.z = c.z * inv;
  v = dot2(q, dd);
  t = dot2(q, e2);
  const pt_f2 uv = u + v;
  // object.cpp:412,417,422,426; (double)t < 1e-6 holds exactly when t <= 1e-6f, the float just below 1e-6
  hit0 = n.x != 0.0f && !(u.x < 0.0f || u.x > 1.0f) && !(v.x < 0.0f || uv.x > 1.0f) && !(t.x <= 1e-6f);
  hit1 = n.y != 0.0f && !(u.y < 0.0f || u.y > 1.0f) && !(v.y < 0.0f || uv.y > 1.0f) && !(t.y <= 1e-6f);
}

// The part of the scene a workgroup keeps in LDS: the first n_nodes nodes (the tree is
// numbered breadth-first from the root, so these are its top levels) and, when the whole
// triangle table fits, all triangles.  Both are stored split by 16-byte piece: piece k of
// node i at nodes[k*n_nodes + i], piece k of triangle i at tris[k*n_tris + i].  Rays of a wave diverge, so every node or triangle
// fetch is 64 separate 16-byte requests; out of LDS such a gather costs a few cycles, out
// of the vector cache at least one cycle per lane.
//
// The copy lives in the kernel's dynamic LDS block and is addressed by offsets into it, not
// through pointers: a pointer that travels through a struct loses its address space and the
// compiler falls back to flat loads (measured: three times the memory instructions).
extern __shared__ float4 pt_smem[];

// Loads and stores with the address space spelled out, so that a value that may come from
// LDS or from global memory is never fetched through a merged (flat) pointer.
typedef float pt_f4 __attribute__((ext_vector_type(4)));
typedef const pt_f4 __attribute__((address_space(3))) *pt_lds_cf4;
typedef uint32_t __attribute__((address_space(3))) *pt_lds_u32;
typedef const pt_f4 __attribute__((address_space(1))) *pt_glb_cf4;
__device__ __forceinline__ float4 lds_f4(uint32_t i) {
  const pt_f4 v = ((pt_lds_cf4)pt_smem)[i];
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 glb_f4(const void *base, size_t i) {
  const pt_f4 v = ((pt_glb_cf4)base)[i];
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t stack_get(uint32_t w) { return ((pt_lds_u32)pt_smem)[w]; }
__device__ __forceinline__ void stack_put(uint32_t w, uint32_t v) { ((pt_lds_u32)pt_smem)[w] = v; }

struct LdsScene {
  uint32_t node_base;  // offset of piece 0 of node 0 in pt_smem, in float4
  uint32_t tri_base;   // offset of piece 0 of triangle 0
  uint32_t gate_base;  // offset of piece 0 of gate box 0 (fast tree only)
  uint32_t n_nodes;
  uint32_t n_tris;
  uint32_t n_gates;
  // Traversal stack: the first stack_lds entries of a lane live in LDS; a tree deeper than
  // that spills the rest to this lane's column of a global array (rarely reached: the stack
  // only holds the far child of nodes where both children were hit).
  uint32_t stack_lds;
  uint32_t *spill;        // this lane's first spill entry
  uint32_t spill_stride;  // distance between a lane's consecutive spill entries, in words
};

__device__ __forceinline__ void push_ref(const LdsScene &C, uint32_t stk, uint32_t stride, int sp, uint32_t v) {
  if ((uint32_t)sp < C.stack_lds) stack_put(stk + (uint32_t)sp * stride, v);
  else C.spill[(size_t)((uint32_t)sp - C.stack_lds) * C.spill_stride] = v;
}
__device__ __forceinline__ uint32_t pop_ref(const LdsScene &C, uint32_t stk, uint32_t stride, int sp) {
  if ((uint32_t)sp < C.stack_lds) return stack_get(stk + (uint32_t)sp * stride);
  return C.spill[(size_t)((uint32_t)sp - C.stack_lds) * C.spill_stride];
}

// The eight-child walk stacks two words per node, always together: in LDS they lie side by side (entry pair
// sp/2 of lane t at word base + sp*stride + 2*t, where the one-word form has entries sp and sp + 1 of lanes t and
// t + ...), written and read as one 64-bit access.  sp and C.stack_lds are even.
typedef unsigned long long __attribute__((address_space(3))) *pt_lds_u64;
__device__ __forceinline__ void push_pair(const LdsScene &C, uint32_t stk, uint32_t stride, int sp, uint32_t a, uint32_t b) {
  if ((uint32_t)sp < C.stack_lds) {
    *(pt_lds_u64)((pt_lds_u32)pt_smem + (stk + threadIdx.x + (uint32_t)sp * stride)) = ((unsigned long long)b << 32) | a;
  } else {
    uint32_t *q = C.spill + (size_t)((uint32_t)sp - C.stack_lds) * C.spill_stride;
    q[0] = a;
    q[C.spill_stride] = b;
  }
}
__device__ __forceinline__ void pop_pair(const LdsScene &C, uint32_t stk, uint32_t stride, int sp, uint32_t &a, uint32_t &b) {
  if ((uint32_t)sp < C.stack_lds) {
    const unsigned long long v = *(pt_lds_u64)((pt_lds_u32)pt_smem + (stk + threadIdx.x + (uint32_t)sp * stride));
    a = (uint32_t)v;
    b = (uint32_t)(v >> 32);
  } else {
    const uint32_t *q = C.spill + (size_t)((uint32_t)sp - C.stack_lds) * C.spill_stride;
    a = q[0];
    b = q[C.spill_stride];
  }
}

// Walks the tree for one ray per lane.  MODE: PT_TRACE_EXACT reaches exactly the leaves the
// reference reaches; PT_TRACE_FAST additionally skips leaves and groups whose box
// the ray misses (with slack) and subtrees that start beyond the best hit (with
// slack), and descends near child first.  ANY: stop at the first accepted triangle.
// `stk` is the offset (in 32-bit words of the kernel's LDS block) of this lane's stack
// column; entries are `stride` words apart.
//
// Shape: the wave alternates between node steps and triangle tests, and a lane that reaches a
// leaf does not test it at once: it parks the leaf in `pend` and goes on with the next node from
// its stack.  The wave tests triangles only when enough of its lanes (S.quorum) hold a parked
// leaf (or when no lane has node work left), so the expensive triangle code runs with many lanes active instead
// of the one or two that happen to be at a leaf in a given step (measured before: 42 % of the
// lanes active per vector instruction).  A lane with a parked leaf that reaches a second one
// waits.  Deferring a leaf only delays the tightening of the cull distance; the result does
// not depend on the order in which leaves are tested (ties go to the larger rank).
// All lanes leave the loop together.
//
// RES: the launch keeps every node, every triangle and the whole stack in LDS (small scenes);
// the tests that choose between LDS and global memory then fall away at compile time.
// Traversal state of one lane's ray, kept across rounds so that a kernel can hand a finished
// lane its next ray while the other lanes of the wave go on (k_trace_queue).
struct RayState {
  V3 o, d;
  float ix, iy, iz;   // 1/d per axis, or the skip value for d == 0
  V3 on, of;          // origins for the entry and exit planes (they differ only on d == 0 axes)
  bool sx, sy, sz;    // 1/d negative
  float cull;         // fast mode: nothing beyond this distance can win
  uint32_t cur;       // node or leaf reference this lane handles next
  uint32_t pend;      // parked leaf reference (leaf references have the top bit set, so 0 = none)
  int sp;
  bool live;          // cur is valid
  bool found;         // ANY: a triangle was hit
  // fast tree only: where this ray finds its near and far planes of a node (pieces 2*axis + sign and
  // 2*axis + 1 - sign): offsets in float4 into the LDS copy (piece * staged nodes) and into a global node
  uint32_t ln[3], lf[3], gn[3], gf[3];
  // fast tree of a scene outside LDS: the same as three sign bits (bit a: 1/d negative on axis a) and three
  // zero bits (bit 3 + a: d == 0 on axis a); the piece offsets and the two plane origins are formed from them
  // at every node.  Such a walk waits on memory, not on the vector unit, and the sixteen registers this saves
  // per lane are a sixth wave per SIMD.
  uint32_t sgn;
  // eight-child tree (walk_round_oct): the group of sibling nodes being visited (first node index; word = order
  // flag << 16 | internal-children mask << 8 | mask of those still to visit), the parked group of sibling leaves
  // (pend = mask of those still to test, pbase = first triangle record, meta bits 0-7 = mask of all leaf
  // children of that node) and a second group of leaves a lane may reach before the first is tested (meta bits
  // 8-15 = its all-leaves mask, bits 16-23 = its mask still to test)
  uint32_t gbase, gw, pbase, qbase, meta;
  HitRec best;
  __device__ __forceinline__ bool done() const { return !live && pend == 0; }
};

template <int MODE, bool WANT_N, bool RES = true>
__device__ __forceinline__ void walk_init(const SceneDev &S, const LdsScene &C, RayState &st, V3 o, V3 d) {
  // 1/d per axis as AABB::intersect forms it (object.cpp:183).  An axis with d == 0 is
  // skipped there without checking that the origin lies inside the slab; a NaN reciprocal
  // reproduces that (NaN never updates tmin/tmax).  Such a ray then passes every box on that
  // axis and walks a large part of the tree for nothing, so the fast mode uses +INF instead:
  // (plane - o) * INF is -INF/+INF on the two sides of the origin and NaN on it, i.e. the
  // slab restricts nothing when the origin is inside or on it and rejects the box otherwise.
  // The box is taken S.margin wider for this check (separate origins for the two planes of
  // the slab): node boxes touch triangle edges, and an edge a rounding error away from the
  // ray's plane can still be hit by the reference's triangle test.
  // The eight-child tree forms its distances as index * (step / d) + (base - o) / d, where an infinite 1/d would
  // turn the sum into NaN; it uses 2^60 instead: the same decision (the sign of plane - o decides), finite sums.
  const float skip = mode_oct(MODE) ? 0x1p60f : (mode_prunes(MODE) ? __builtin_inff() : __builtin_nanf(""));
  st.o = o;
  st.d = d;
  st.ix = d.x != 0.0f ? 1.0f / d.x : skip;
  st.iy = d.y != 0.0f ? 1.0f / d.y : skip;
  st.iz = d.z != 0.0f ? 1.0f / d.z : skip;
  constexpr bool LEAN = mode_wide(MODE) && !RES;
  st.sgn = 0u;
  if constexpr (LEAN) {
    st.sgn = (st.ix < 0.0f ? 1u : 0u) | (st.iy < 0.0f ? 2u : 0u) | (st.iz < 0.0f ? 4u : 0u) |
             (d.x == 0.0f ? 8u : 0u) | (d.y == 0.0f ? 16u : 0u) | (d.z == 0.0f ? 32u : 0u);
  } else {
    const float mx = (mode_prunes(MODE) && d.x == 0.0f) ? S.margin : 0.0f;
    const float my = (mode_prunes(MODE) && d.y == 0.0f) ? S.margin : 0.0f;
    const float mz = (mode_prunes(MODE) && d.z == 0.0f) ? S.margin : 0.0f;
    st.on = mk(o.x + mx, o.y + my, o.z + mz);
    st.of = mk(o.x - mx, o.y - my, o.z - mz);
    st.sx = st.ix < 0.0f;
    st.sy = st.iy < 0.0f;
    st.sz = st.iz < 0.0f;
  }
  if (mode_wide(MODE) && !LEAN) {
    const uint32_t s3[3] = {st.sx ? 1u : 0u, st.sy ? 1u : 0u, st.sz ? 1u : 0u};
    st.sx = st.sy = st.sz = false;  // the fast tree's walk addresses its planes by sign (below) and keeps no sign flags
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      st.gn[a] = 2u * a + s3[a];
      st.gf[a] = 2u * a + 1u - s3[a];
      st.ln[a] = st.gn[a] * C.n_nodes;
      st.lf[a] = st.gf[a] * C.n_nodes;
    }
  }
  st.best.t = __builtin_inff();
  st.best.slot = 0xffffffffu;
  st.best.rank = 0xffffffffu;
  st.best.u = 0.0f;
  st.best.v = 0.0f;
  if (WANT_N) {
    st.best.n = mk(0.0f, 0.0f, 0.0f);
    st.best.mat = 0u;
  }
  st.cull = __builtin_inff();
  st.cur = 0;
  st.pend = 0;
  st.sp = 0;
  st.live = true;
  st.found = false;
  st.gbase = 0u; st.gw = 0u; st.pbase = 0u; st.qbase = 0u; st.meta = 0u;
}

// The reference's box test (AABB::intersect, object.cpp:179-195) for the gate of a triangle record of the
// fast tree: planes as given, the origin as given, an axis with d == 0 skipped (NaN never updates).
template <bool RES>
__device__ __forceinline__ bool gate_pass(const SceneDev &S, const LdsScene &C, const RayState &st, uint32_t gate) {
  if (gate == ptl::NO_GATE) return true;
  float4 g0, g1;
  if (RES || gate < C.n_gates) {
    g0 = lds_f4(C.gate_base + gate);
    g1 = lds_f4(C.gate_base + C.n_gates + gate);
  } else {
    g0 = glb_f4(S.gates, 2u * (size_t)gate);
    g1 = glb_f4(S.gates, 2u * (size_t)gate + 1);
  }
  const float nan = __builtin_nanf("");
  const float ix = st.d.x != 0.0f ? st.ix : nan, iy = st.d.y != 0.0f ? st.iy : nan, iz = st.d.z != 0.0f ? st.iz : nan;
  const bool sx = ix < 0.0f, sy = iy < 0.0f, sz = iz < 0.0f;  // the swap of object.cpp:186-187
  const float ax = ((sx ? g1.x : g0.x) - st.o.x) * ix, bx = ((sx ? g0.x : g1.x) - st.o.x) * ix;
  const float ay = ((sy ? g1.y : g0.y) - st.o.y) * iy, by = ((sy ? g0.y : g1.y) - st.o.y) * iy;
  const float az = ((sz ? g1.z : g0.z) - st.o.z) * iz, bz = ((sz ? g0.z : g1.z) - st.o.z) * iz;
  const float tn = fmaxf(fmaxf(fmaxf(1e-5f, ax), ay), az);
  const float tf = fminf(fminf(fminf(__builtin_inff(), bx), by), bz);
  return tf >= tn;
}

// A triangle hit of a closest-hit walk: it replaces the best one when its t is smaller, or equal with a
// larger rank (the reference's leftmost-in-leaf / right-sibling rule); in the fast tree only if the ray
// also passes the triangle's gate.  c is the third float4 of the triangle's record (gate, material, rank).
template <int MODE, bool COUNT, bool WANT_N, bool RES>
__device__ __forceinline__ void consider_hit(const SceneDev &S, const LdsScene &C, RayState &st, uint32_t &n_boxes, bool hit,
                                             float t, uint32_t slot, float4 c) {
  const uint32_t rank = __float_as_uint(c.w);
  bool wins = hit && (t < st.best.t || (t == st.best.t && (int32_t)rank > (int32_t)st.best.rank));
  if (mode_wide(MODE)) {
    if (wins) {
      if (COUNT) n_boxes += 1;
      wins = gate_pass<RES>(S, C, st, __float_as_uint(c.y));
    }
  }
  if (wins) {
    st.best.t = t;
    st.best.slot = slot;
    st.best.rank = rank;
  }
}

// What a caller wants to know about the closest hit besides t and the triangle: the barycentrics,
// Triangle::normal() (un-normalised, object.cpp:393) and the material id.  The walk itself carries only t,
// slot and rank; these come from the winner's record once the walk is over (u and v by the scalar form of
// the same test: the same operations in the same order, so the same bits as the packed form gave).
template <int MODE, bool RES>
__device__ __forceinline__ void hit_finish(const SceneDev &S, const LdsScene &C, V3 o, V3 d, HitRec &h, bool want_uv, bool want_n) {
  h.u = 0.0f;
  h.v = 0.0f;
  h.n = mk(0.0f, 0.0f, 0.0f);
  h.mat = 0u;
  if (h.slot == 0xffffffffu) return;
  float4 a, b, c;
  if (RES || h.slot < C.n_tris) {
    const uint32_t at = C.tri_base + h.slot;
    a = lds_f4(at); b = lds_f4(at + C.n_tris); c = lds_f4(at + 2u * C.n_tris);
  } else {
    const void *gtris = mode_wide(MODE) ? (const void *)S.tris4 : (const void *)S.tris;
    const size_t at = 3u * (size_t)h.slot;
    a = glb_f4(gtris, at); b = glb_f4(gtris, at + 1); c = glb_f4(gtris, at + 2);
  }
  if (want_uv) {
    float t, u = 0.0f, v = 0.0f;
    (void)tri_test(a, b, c, o, d, t, u, v);
    h.u = u;
    h.v = v;
  }
  if (want_n) {
    h.n = cross(mk(a.w, b.x, b.y), mk(b.z, b.w, c.x));  // (v1-v0) x (v2-v0)
    h.mat = __float_as_uint(c.z);
  }
}

// One round for the wave: a node step for every lane that has one, then — when enough lanes
// hold a parked leaf (`quorum` of them for a closest-hit walk) or no lane has node work — the
// triangle tests of the parked leaves.
// Lanes whose state is done() do nothing.
template <int MODE, bool ANY, bool COUNT, bool WANT_N, bool RES>
__device__ __forceinline__ void walk_round(const SceneDev &S, const LdsScene &C, RayState &st, uint32_t stk,
                                           uint32_t stride, uint32_t &n_boxes, uint32_t &n_tris, uint32_t quorum) {
  constexpr bool PRUNE = mode_prunes(MODE), WIDE = mode_wide(MODE);
  if (st.live && !(st.cur & ptl::REF_LEAF)) {
    float far = (PRUNE && !ANY) ? st.cull : __builtin_inff();
    if constexpr (WIDE) {
      // A node reached through a REF_NOCULL reference has an ill-conditioned triangle below it, which can
      // report a hit at any distance: its children are entered whenever the ray passes their boxes, however
      // far away those begin.  (Scenes without such triangles have no such references.)
      if (PRUNE && !ANY && S.n_ill != 0u && (st.cur & ptl::REF_NOCULL)) far = __builtin_inff();
      const uint32_t node = st.cur & ~ptl::REF_NOCULL;
      // four children; this ray's near planes of all four come as one float4 per axis, its far planes too
      float4 nx, fx, ny, fy, nz, fz, rr;
      pt_f2 onx, ony, onz, ofx, ofy, ofz;
      if constexpr (RES) {
        const uint32_t at = C.node_base + node;
        nx = lds_f4(at + st.ln[0]); fx = lds_f4(at + st.lf[0]);
        ny = lds_f4(at + st.ln[1]); fy = lds_f4(at + st.lf[1]);
        nz = lds_f4(at + st.ln[2]); fz = lds_f4(at + st.lf[2]);
        rr = lds_f4(at + 6u * C.n_nodes);
        onx = splat2(st.on.x); ony = splat2(st.on.y); onz = splat2(st.on.z);
        ofx = splat2(st.of.x); ofy = splat2(st.of.y); ofz = splat2(st.of.z);
      } else {
        // piece 2*axis + sign holds this ray's near planes, 2*axis + 1 - sign its far planes
        const uint32_t px = st.sgn & 1u, py = 2u + ((st.sgn >> 1) & 1u), pz = 4u + ((st.sgn >> 2) & 1u);
        if (node < C.n_nodes) {
          const uint32_t at = C.node_base + node, n = C.n_nodes;
          nx = lds_f4(at + px * n); fx = lds_f4(at + (px ^ 1u) * n);
          ny = lds_f4(at + py * n); fy = lds_f4(at + (py ^ 1u) * n);
          nz = lds_f4(at + pz * n); fz = lds_f4(at + (pz ^ 1u) * n);
          rr = lds_f4(at + 6u * n);
        } else {
          const size_t at = 8u * (size_t)node;
          nx = glb_f4(S.nodes4, at + px); fx = glb_f4(S.nodes4, at + (px ^ 1u));
          ny = glb_f4(S.nodes4, at + py); fy = glb_f4(S.nodes4, at + (py ^ 1u));
          nz = glb_f4(S.nodes4, at + pz); fz = glb_f4(S.nodes4, at + (pz ^ 1u));
          rr = glb_f4(S.nodes4, at + 6);
        }
        // the slab of an axis with d == 0 is taken S.margin wider (see walk_init): the same sums as st.on / st.of
        const float mx = (st.sgn & 8u) ? S.margin : 0.0f, my = (st.sgn & 16u) ? S.margin : 0.0f, mz = (st.sgn & 32u) ? S.margin : 0.0f;
        onx = splat2(st.o.x + mx); ony = splat2(st.o.y + my); onz = splat2(st.o.z + mz);
        ofx = splat2(st.o.x - mx); ofy = splat2(st.o.y - my); ofz = splat2(st.o.z - mz);
      }
      if (COUNT) n_boxes += 4;
      const uint32_t r0 = __float_as_uint(rr.x), r1 = __float_as_uint(rr.y), r2 = __float_as_uint(rr.z), r3 = __float_as_uint(rr.w);
      const pt_f2 ix = splat2(st.ix), iy = splat2(st.iy), iz = splat2(st.iz);
      const pt_f2 ax01 = (pair2(nx.x, nx.y) - onx) * ix, ax23 = (pair2(nx.z, nx.w) - onx) * ix;
      const pt_f2 ay01 = (pair2(ny.x, ny.y) - ony) * iy, ay23 = (pair2(ny.z, ny.w) - ony) * iy;
      const pt_f2 az01 = (pair2(nz.x, nz.y) - onz) * iz, az23 = (pair2(nz.z, nz.w) - onz) * iz;
      const pt_f2 bx01 = (pair2(fx.x, fx.y) - ofx) * ix, bx23 = (pair2(fx.z, fx.w) - ofx) * ix;
      const pt_f2 by01 = (pair2(fy.x, fy.y) - ofy) * iy, by23 = (pair2(fy.z, fy.w) - ofy) * iy;
      const pt_f2 bz01 = (pair2(fz.x, fz.y) - ofz) * iz, bz23 = (pair2(fz.z, fz.w) - ofz) * iz;
      // NaN (a plane through the origin of a ray parallel to it) never restricts.  The interval starts at
      // S.tmin_own: these boxes are this tree's own, a triangle hit counts from t = 1e-6
      // (object.cpp:425), and a leaf box a margin thick can be left before the reference's EPSILON = 1e-5
      // by a ray with a long direction vector.
      const float tn0 = fmaxf(fmaxf(fmaxf(S.tmin_own, ax01.x), ay01.x), az01.x);
      const float tn1 = fmaxf(fmaxf(fmaxf(S.tmin_own, ax01.y), ay01.y), az01.y);
      const float tn2 = fmaxf(fmaxf(fmaxf(S.tmin_own, ax23.x), ay23.x), az23.x);
      const float tn3 = fmaxf(fmaxf(fmaxf(S.tmin_own, ax23.y), ay23.y), az23.y);
      bool h0, h1, h2, h3;
      if (ANY) {
        // no far limit: the infinity is left out of the minimum and the question is "not nearer than the entry",
        // which accepts the all-NaN case as the infinity did
        const float tf0 = fminf(fminf(bx01.x, by01.x), bz01.x), tf1 = fminf(fminf(bx01.y, by01.y), bz01.y);
        const float tf2 = fminf(fminf(bx23.x, by23.x), bz23.x), tf3 = fminf(fminf(bx23.y, by23.y), bz23.y);
        h0 = !(tf0 < tn0); h1 = !(tf1 < tn1); h2 = !(tf2 < tn2); h3 = !(tf3 < tn3);
      } else {
        const float tf0 = fminf(fminf(fminf(far, bx01.x), by01.x), bz01.x), tf1 = fminf(fminf(fminf(far, bx01.y), by01.y), bz01.y);
        const float tf2 = fminf(fminf(fminf(far, bx23.x), by23.x), bz23.x), tf3 = fminf(fminf(fminf(far, bx23.y), by23.y), bz23.y);
        h0 = tf0 >= tn0; h1 = tf1 >= tn1; h2 = tf2 >= tn2; h3 = tf3 >= tn3;
      }
      if (h0 || h1 || h2 || h3) {
        // nearest child next; the others wait on the stack
        const float inf = __builtin_inff();
        const float k0 = h0 ? tn0 : inf, k1 = h1 ? tn1 : inf, k2 = h2 ? tn2 : inf, k3 = h3 ? tn3 : inf;
        const float km = fminf(fminf(k0, k1), fminf(k2, k3));
        const bool c0 = h0 && k0 == km;
        const bool c1 = !c0 && h1 && k1 == km;
        const bool c2 = !c0 && !c1 && h2 && k2 == km;
        const bool c3 = !c0 && !c1 && !c2;
        if (h3 && !c3) { push_ref(C, stk, stride, st.sp, r3); ++st.sp; }
        if (h2 && !c2) { push_ref(C, stk, stride, st.sp, r2); ++st.sp; }
        if (h1 && !c1) { push_ref(C, stk, stride, st.sp, r1); ++st.sp; }
        if (h0 && !c0) { push_ref(C, stk, stride, st.sp, r0); ++st.sp; }
        st.cur = c0 ? r0 : (c1 ? r1 : (c2 ? r2 : r3));
      } else if (st.sp == 0) {
        st.live = false;
      } else {
        --st.sp;
        st.cur = pop_ref(C, stk, stride, st.sp);
      }
    } else {
      float4 q0, q1, q2;
      uint4 q3;
      if (RES || st.cur < C.n_nodes) {
        const uint32_t at = C.node_base + st.cur;
        q0 = lds_f4(at); q1 = lds_f4(at + C.n_nodes); q2 = lds_f4(at + 2u * C.n_nodes);
        const float4 r = lds_f4(at + 3u * C.n_nodes);
        q3 = make_uint4(__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w));
      } else {
        const size_t at = 4u * (size_t)st.cur;
        q0 = glb_f4(S.nodes, at); q1 = glb_f4(S.nodes, at + 1); q2 = glb_f4(S.nodes, at + 2);
        const float4 r = glb_f4(S.nodes, at + 3);
        q3 = make_uint4(__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w));
      }
      if (COUNT) n_boxes += 2;
      // child 0: lo = q0.xyz, hi = (q0.w, q1.x, q1.y); child 1: lo = (q1.z, q1.w, q2.x), hi = q2.yzw
      // both children side by side in packed pairs: (plane - origin) * (1/d), object.cpp:184-185
      const pt_f2 ax = (pair2(st.sx ? q0.w : q0.x, st.sx ? q2.y : q1.z) - splat2(st.on.x)) * splat2(st.ix);
      const pt_f2 bx = (pair2(st.sx ? q0.x : q0.w, st.sx ? q1.z : q2.y) - splat2(st.of.x)) * splat2(st.ix);
      const pt_f2 ay = (pair2(st.sy ? q1.x : q0.y, st.sy ? q2.z : q1.w) - splat2(st.on.y)) * splat2(st.iy);
      const pt_f2 by = (pair2(st.sy ? q0.y : q1.x, st.sy ? q1.w : q2.z) - splat2(st.of.y)) * splat2(st.iy);
      const pt_f2 az = (pair2(st.sz ? q1.y : q0.z, st.sz ? q2.w : q2.x) - splat2(st.on.z)) * splat2(st.iz);
      const pt_f2 bz = (pair2(st.sz ? q0.z : q1.y, st.sz ? q2.x : q2.w) - splat2(st.of.z)) * splat2(st.iz);
      // tmin starts at EPSILON, tmax at infinity; NaN never updates (object.cpp:180,188-191).  That is the
      // test of an internal node of the reference tree, and it stays exactly that in the pruning walk too:
      // its box has no margin, so its entry distance says nothing reliable about the t of a hit below it (a
      // hit point a rounding error outside the box, a ray nearly parallel to that face) and it is not skipped
      // by distance.  Leaf and group boxes are the flattener's own, widened: they start at S.tmin_own
      // and end at the cull distance.
      const bool int0 = !PRUNE || q3.z == KIND_INTERNAL, int1 = !PRUNE || q3.w == KIND_INTERNAL;
      const float tn0 = fmaxf(fmaxf(fmaxf(int0 ? 1e-5f : S.tmin_own, ax.x), ay.x), az.x);
      const float tn1 = fmaxf(fmaxf(fmaxf(int1 ? 1e-5f : S.tmin_own, ax.y), ay.y), az.y);
      const float tf0 = fminf(fminf(fminf(int0 ? __builtin_inff() : far, bx.x), by.x), bz.x);
      const float tf1 = fminf(fminf(fminf(int1 ? __builtin_inff() : far, bx.y), by.y), bz.y);
      bool p0, p1;
      if (!PRUNE) {
        // only an internal node of the reference tree is box-tested there (object.cpp:277)
        p0 = (q3.z == KIND_INTERNAL) ? (tf0 >= tn0) : true;
        p1 = (q3.w == KIND_INTERNAL) ? (tf1 >= tn1) : true;
      } else {
        // leaf and group boxes were widened by the flattener, and their tf started at the cull
        // distance, so one comparison covers the box test and the pruning
        p0 = tf0 >= tn0;
        p1 = tf1 >= tn1;
      }
      if (p0 && p1) {
        const bool swap = PRUNE ? (tn1 < tn0) : false;  // near child first (an any-hit walk ends sooner too)
        push_ref(C, stk, stride, st.sp, swap ? q3.x : q3.y);
        ++st.sp;
        st.cur = swap ? q3.y : q3.x;
      } else if (p0) {
        st.cur = q3.x;
      } else if (p1) {
        st.cur = q3.y;
      } else if (st.sp == 0) {
        st.live = false;
      } else {
        --st.sp;
        st.cur = pop_ref(C, stk, stride, st.sp);
      }
    }
  }
  if (st.live && (st.cur & ptl::REF_LEAF) && st.pend == 0) {  // park the leaf, go on with the next node
    st.pend = st.cur;
    if (st.sp == 0) {
      st.live = false;
    } else {
      --st.sp;
      st.cur = pop_ref(C, stk, stride, st.sp);
    }
  }
  const unsigned long long m_pend = __ballot(st.pend != 0);
  const unsigned long long m_node = __ballot(st.live && !(st.cur & ptl::REF_LEAF));
  if (m_pend == 0) return;
  // closest-hit walks run with (nearly) full waves: an absolute count will do; any-hit walks sit
  // inside the light loop of the shading kernel with whatever lanes hit something
  bool enough;
  if (ANY) enough = (uint32_t)__popcll(m_pend) * 64u >= S.quorum_any * (uint32_t)__popcll(__ballot(!st.done()));
  else enough = (uint32_t)__popcll(m_pend) >= quorum;
  if (enough || m_node == 0) {
    if (st.pend != 0) {
      const uint32_t first = st.pend & ptl::REF_FIRST_MASK;
      const uint32_t cnt = ((st.pend >> 28) & (WIDE ? 1u : 7u)) + 1u;  // the fast tree's leaves hold one or two
      st.pend = 0;
      const bool in_lds = RES || first + cnt <= C.n_tris;
      const void *gtris = WIDE ? (const void *)S.tris4 : (const void *)S.tris;
      for (uint32_t k = 0; k < cnt; k += 2) {
        // an odd last triangle is paired with itself and its second copy ignored
        const bool two = k + 1 < cnt;
        const uint32_t s0 = first + k, s1 = two ? s0 + 1 : s0;
        if (COUNT) n_tris += two ? 2u : 1u;
        float4 a0, b0, c0, a1, b1, c1;
        if (in_lds) {
          const uint32_t at0 = C.tri_base + s0, at1 = C.tri_base + s1;
          a0 = lds_f4(at0); b0 = lds_f4(at0 + C.n_tris); c0 = lds_f4(at0 + 2u * C.n_tris);
          a1 = lds_f4(at1); b1 = lds_f4(at1 + C.n_tris); c1 = lds_f4(at1 + 2u * C.n_tris);
        } else {
          const size_t at0 = 3u * (size_t)s0, at1 = 3u * (size_t)s1;
          a0 = glb_f4(gtris, at0); b0 = glb_f4(gtris, at0 + 1); c0 = glb_f4(gtris, at0 + 2);
          a1 = glb_f4(gtris, at1); b1 = glb_f4(gtris, at1 + 1); c1 = glb_f4(gtris, at1 + 2);
        }
        bool h0, h1;
        pt_f2 t2, u2, v2;
        tri_test2(a0, b0, c0, a1, b1, c1, st.o, st.d, h0, h1, t2, u2, v2);
        h1 = h1 && two;
        if (ANY) {
          if (WIDE) {
            // a hit counts only where the reference reaches that triangle's leaf
            if (h0) { h0 = gate_pass<RES>(S, C, st, __float_as_uint(c0.y)); if (COUNT) n_boxes += 1; }
            if (h1 && !h0) { h1 = gate_pass<RES>(S, C, st, __float_as_uint(c1.y)); if (COUNT) n_boxes += 1; }
          }
          if (h0 || h1) {
            st.found = true;
            st.live = false;
            break;
          }
        } else {
          consider_hit<MODE, COUNT, WANT_N, RES>(S, C, st, n_boxes, h0, t2.x, s0, c0);
          consider_hit<MODE, COUNT, WANT_N, RES>(S, C, st, n_boxes, h1, t2.y, s1, c1);
          if (PRUNE) st.cull = st.best.t + (st.best.t * PT_SLACK_REL + PT_SLACK_ABS);
        }
      }
    }
  }
}

#ifdef PT_DEBUG_ROUNDS
__device__ unsigned long long g_dbg[24];
#endif
// The round of the eight-child tree (scenes outside LDS; pt_layout.h, Oct8Node).  A node names its children by
// two base indices and a mask, so what waits on the stack is one entry of two words per node (first child, the
// children still to visit) instead of a reference per child, and a step does no per-child bookkeeping: it tests
// the eight boxes, forms the masks, and goes on with the first child in the node's own order (children are
// sorted along the node's longest axis; a ray that runs against that axis takes them from the other end).
// Leaves are parked and tested by quorum as in walk_round, a node's leaf children as one group.
template <int MODE, bool ANY, bool COUNT, bool WANT_N>
__device__ __forceinline__ void walk_round_oct(const SceneDev &S, const LdsScene &C, RayState &st, uint32_t stk,
                                               uint32_t stride, uint32_t &n_boxes, uint32_t &n_tris, uint32_t quorum) {
#ifdef PT_DEBUG_ROUNDS
  {
    const unsigned long long mN = __ballot(st.live && (st.meta >> 16) == 0u), mB = __ballot(!st.done());
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&g_dbg[ANY ? 8 : 0], 1ull);
      if (mN) atomicAdd(&g_dbg[(ANY ? 8 : 0) + 1], 1ull);
      atomicAdd(&g_dbg[(ANY ? 8 : 0) + 2], (unsigned long long)__popcll(mN));
      atomicAdd(&g_dbg[(ANY ? 8 : 0) + 3], (unsigned long long)__popcll(mB));
    }
  }
#endif
  // a lane that holds two groups of parked leaves waits for the triangle tests before it reads another node
  if (st.live && (st.meta >> 16) == 0u) {
    const float far = (!ANY) ? st.cull : __builtin_inff();
    const uint32_t node = st.cur;
    float4 p0, p1, qx, qy, qz;
    if (node < C.n_nodes) {
      const uint32_t at = C.node_base + node, n = C.n_nodes;
      p0 = lds_f4(at); p1 = lds_f4(at + n); qx = lds_f4(at + 2u * n); qy = lds_f4(at + 3u * n); qz = lds_f4(at + 4u * n);
    } else {
      const size_t at = 8u * (size_t)node;
      p0 = glb_f4(S.nodes8, at); p1 = glb_f4(S.nodes8, at + 1); qx = glb_f4(S.nodes8, at + 2); qy = glb_f4(S.nodes8, at + 3);
      qz = glb_f4(S.nodes8, at + 4);
    }
    if (COUNT) n_boxes += 8;
    // A child's plane is base + index * step, so its distance along the ray is index * (step / d) + (base - o) / d:
    // one fused multiply-add per plane after a few products per node.
    const uint32_t ex = __float_as_uint(p0.w);
    const float ax = __uint_as_float((ex & 0xffu) << 23) * st.ix;
    const float ay = __uint_as_float((ex & 0xff00u) << 15) * st.iy;
    const float az = __uint_as_float((ex & 0xff0000u) << 7) * st.iz;
    float bnx = (p0.x - st.o.x) * st.ix, bny = (p0.y - st.o.y) * st.iy, bnz = (p0.z - st.o.z) * st.iz;
    float bfx = bnx, bfy = bny, bfz = bnz;
    if (__ballot((st.sgn & 0x38u) != 0u) != 0ull) {
      // some ray of the wave runs parallel to a coordinate plane: its slab on that axis is S.margin wider (walk_init)
      const float mx = (st.sgn & 8u) ? S.margin : 0.0f, my = (st.sgn & 16u) ? S.margin : 0.0f, mz = (st.sgn & 32u) ? S.margin : 0.0f;
      bnx = (p0.x - (st.o.x + mx)) * st.ix; bfx = (p0.x - (st.o.x - mx)) * st.ix;
      bny = (p0.y - (st.o.y + my)) * st.iy; bfy = (p0.y - (st.o.y - my)) * st.iy;
      bnz = (p0.z - (st.o.z + mz)) * st.iz; bfz = (p0.z - (st.o.z - mz)) * st.iz;
    }
    // index bytes of this ray's near and far planes: words 0-1 of an axis' piece hold lo, words 2-3 hi
    const bool gx = (st.sgn & 1u) != 0u, gy = (st.sgn & 2u) != 0u, gz = (st.sgn & 4u) != 0u;
    const uint32_t x0 = __float_as_uint(qx.x), x1 = __float_as_uint(qx.y), x2 = __float_as_uint(qx.z), x3 = __float_as_uint(qx.w);
    const uint32_t y0 = __float_as_uint(qy.x), y1 = __float_as_uint(qy.y), y2 = __float_as_uint(qy.z), y3 = __float_as_uint(qy.w);
    const uint32_t z0 = __float_as_uint(qz.x), z1 = __float_as_uint(qz.y), z2 = __float_as_uint(qz.z), z3 = __float_as_uint(qz.w);
    const uint32_t nxw[2] = {gx ? x2 : x0, gx ? x3 : x1}, fxw[2] = {gx ? x0 : x2, gx ? x1 : x3};
    const uint32_t nyw[2] = {gy ? y2 : y0, gy ? y3 : y1}, fyw[2] = {gy ? y0 : y2, gy ? y1 : y3};
    const uint32_t nzw[2] = {gz ? z2 : z0, gz ? z3 : z1}, fzw[2] = {gz ? z0 : z2, gz ? z1 : z3};
    uint32_t hm = 0u;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      // two children side by side in packed pairs
      const int w = k >> 2, s0 = 8 * (k & 3), s1 = s0 + 8;
      const pt_f2 t0x = __builtin_elementwise_fma(pair2((float)((nxw[w] >> s0) & 0xffu), (float)((nxw[w] >> s1) & 0xffu)), splat2(ax), splat2(bnx));
      const pt_f2 t0y = __builtin_elementwise_fma(pair2((float)((nyw[w] >> s0) & 0xffu), (float)((nyw[w] >> s1) & 0xffu)), splat2(ay), splat2(bny));
      const pt_f2 t0z = __builtin_elementwise_fma(pair2((float)((nzw[w] >> s0) & 0xffu), (float)((nzw[w] >> s1) & 0xffu)), splat2(az), splat2(bnz));
      const pt_f2 t1x = __builtin_elementwise_fma(pair2((float)((fxw[w] >> s0) & 0xffu), (float)((fxw[w] >> s1) & 0xffu)), splat2(ax), splat2(bfx));
      const pt_f2 t1y = __builtin_elementwise_fma(pair2((float)((fyw[w] >> s0) & 0xffu), (float)((fyw[w] >> s1) & 0xffu)), splat2(ay), splat2(bfy));
      const pt_f2 t1z = __builtin_elementwise_fma(pair2((float)((fzw[w] >> s0) & 0xffu), (float)((fzw[w] >> s1) & 0xffu)), splat2(az), splat2(bfz));
      // NaN (an infinite 1/d against a plane it cannot order) never restricts.  A visibility walk has no far limit:
      // it leaves the infinity out of the minimum and asks "not nearer than the entry" instead, which accepts the
      // all-NaN case as the infinity did.
      const float na = fmaxf(fmaxf(fmaxf(S.tmin_own, t0x.x), t0y.x), t0z.x);
      const float nb = fmaxf(fmaxf(fmaxf(S.tmin_own, t0x.y), t0y.y), t0z.y);
      if (ANY) {
        const float fa = fminf(fminf(t1x.x, t1y.x), t1z.x), fb = fminf(fminf(t1x.y, t1y.y), t1z.y);
        hm |= !(fa < na) ? (1u << k) : 0u;
        hm |= !(fb < nb) ? (2u << k) : 0u;
      } else {
        const float fa = fminf(fminf(fminf(far, t1x.x), t1y.x), t1z.x), fb = fminf(fminf(fminf(far, t1x.y), t1y.y), t1z.y);
        hm |= (fa >= na) ? (1u << k) : 0u;
        hm |= (fb >= nb) ? (2u << k) : 0u;
      }
    }
    const uint32_t cbase = __float_as_uint(p1.x), tbase = __float_as_uint(p1.y), masks = __float_as_uint(p1.z);
    const uint32_t imask = masks & 0xffu, lall = (masks >> 8) & 0xffu;   // internal children; leaf children
    const uint32_t hn = hm & imask, hl = hm & lall;
    // visiting order: ascending slots, or descending when the ray runs against the axis the children are sorted on
    const uint32_t desc = (st.sgn >> ((ex >> 24) & 3u)) & 1u;
    if (hl != 0u) {
      if (st.pend == 0u) { st.pend = hl; st.pbase = tbase; st.meta = (st.meta & ~0xffu) | lall; }
      else { st.qbase = tbase; st.meta = (st.meta & 0xffu) | (lall << 8) | (hl << 16); }
    }
    if (hn != 0u) {
      if ((st.gw & 0xffu) != 0u) {  // the siblings still to visit wait on the stack
        push_pair(C, stk, stride, st.sp, st.gbase, st.gw);
        st.sp += 2;
      }
      st.gbase = cbase;
      st.gw = (desc << 16) | (imask << 8) | hn;
    }
    if ((st.gw & 0xffu) == 0u) {
      if (st.sp == 0) {
        st.live = false;
      } else {
        st.sp -= 2;
        pop_pair(C, stk, stride, st.sp, st.gbase, st.gw);
      }
    }
    if (st.live) {
      const uint32_t left = st.gw & 0xffu;
      const uint32_t k = (st.gw & 0x10000u) ? 31u - (uint32_t)__builtin_clz(left) : (uint32_t)__builtin_ctz(left);
      st.cur = st.gbase + (uint32_t)__popc((st.gw >> 8) & ((1u << k) - 1u));
      st.gw &= ~(1u << k);
    }
  }
  const unsigned long long m_pend = __ballot(st.pend != 0u);
  if (m_pend == 0ull) return;
  const unsigned long long m_node = __ballot(st.live && (st.meta >> 16) == 0u);
  bool enough;
  if (ANY) enough = (uint32_t)__popcll(m_pend) * 64u >= S.quorum_any * (uint32_t)__popcll(__ballot(!st.done()));
  else enough = (uint32_t)__popcll(m_pend) >= quorum;
#ifdef PT_DEBUG_ROUNDS
  if ((enough || m_node == 0ull) && (threadIdx.x & 63) == 0) {
    atomicAdd(&g_dbg[(ANY ? 8 : 0) + 4], 1ull);
    atomicAdd(&g_dbg[(ANY ? 8 : 0) + 5], (unsigned long long)__popcll(m_pend));
  }
#endif
  if ((enough || m_node == 0ull) && st.pend != 0u) {
    // one leaf of the parked group: two triangle records (a leaf of one triangle is followed by a record no ray hits)
    const uint32_t k = (uint32_t)__builtin_ctz(st.pend);
    st.pend &= st.pend - 1u;
    const uint32_t s0 = st.pbase + 2u * (uint32_t)__popc(st.meta & 0xffu & ((1u << k) - 1u)), s1 = s0 + 1u;
    if (st.pend == 0u && (st.meta >> 16) != 0u) {  // the second group moves up
      st.pend = st.meta >> 16;
      st.pbase = st.qbase;
      st.meta = (st.meta >> 8) & 0xffu;
    }
    if (COUNT) n_tris += 2u;
    const size_t at0 = 3u * (size_t)s0;
    const float4 a0 = glb_f4(S.tris4, at0), b0 = glb_f4(S.tris4, at0 + 1), c0 = glb_f4(S.tris4, at0 + 2);
    const float4 a1 = glb_f4(S.tris4, at0 + 3), b1 = glb_f4(S.tris4, at0 + 4), c1 = glb_f4(S.tris4, at0 + 5);
    bool h0, h1;
    pt_f2 t2, u2, v2;
    tri_test2(a0, b0, c0, a1, b1, c1, st.o, st.d, h0, h1, t2, u2, v2);
    if (ANY) {
      // a hit counts only where the reference reaches that triangle's leaf
      if (h0) { h0 = gate_pass<false>(S, C, st, __float_as_uint(c0.y)); if (COUNT) n_boxes += 1; }
      if (h1 && !h0) { h1 = gate_pass<false>(S, C, st, __float_as_uint(c1.y)); if (COUNT) n_boxes += 1; }
      if (h0 || h1) {
        st.found = true;
        st.live = false;
        st.pend = 0u;
        st.meta = 0u;
      }
    } else {
      consider_hit<MODE, COUNT, WANT_N, false>(S, C, st, n_boxes, h0, t2.x, s0, c0);
      consider_hit<MODE, COUNT, WANT_N, false>(S, C, st, n_boxes, h1, t2.y, s1, c1);
      st.cull = st.best.t + (st.best.t * PT_SLACK_REL + PT_SLACK_ABS);
    }
  }
}

// One round.
template <int MODE, bool ANY, bool COUNT, bool WANT_N, bool RES>
__device__ __forceinline__ void walk_step(const SceneDev &S, const LdsScene &C, RayState &st, uint32_t stk,
                                          uint32_t stride, uint32_t &n_boxes, uint32_t &n_tris, uint32_t quorum) {
  if constexpr (mode_oct(MODE)) walk_round_oct<MODE, ANY, COUNT, WANT_N>(S, C, st, stk, stride, n_boxes, n_tris, quorum);
  else walk_round<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, stride, n_boxes, n_tris, quorum);
}

// The whole walk for one ray per lane; all lanes return together.
template <int MODE, bool ANY, bool COUNT, bool WANT_N = false, bool RES = false>
__device__ __forceinline__ bool walk(const SceneDev &S, const LdsScene &C, V3 o, V3 d, uint32_t stk, uint32_t stride,
                                     HitRec &best, uint32_t &n_boxes, uint32_t &n_tris) {
  RayState st;
  walk_init<MODE, WANT_N, RES>(S, C, st, o, d);
  do {
    walk_step<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, stride, n_boxes, n_tris, S.quorum);
  } while (__ballot(!st.done()) != 0);
  best = st.best;
  if (!ANY) hit_finish<MODE, RES>(S, C, o, d, best, WANT_N && S.n_tex != 0, WANT_N);
  return ANY ? st.found : st.best.slot != 0xffffffffu;
}

}  // namespace ptd
#endif

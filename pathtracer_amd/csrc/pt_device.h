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

using ptl::KIND_EMPTY;
using ptl::KIND_GROUP;
using ptl::KIND_INTERNAL;
using ptl::KIND_LEAF;

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

// ------------------------------------------------------------------ tree walk
struct SceneDev {
  const ptl::WideNode *nodes;
  const ptl::TriRec *tris;
  const ptl::MatRec *mats;
  const ptl::LightRec *lights;
  const ptl::LightTri *light_tris;
  uint32_t n_nodes, n_tris, n_lights, n_mats;
};

struct HitRec {
  float t;
  uint32_t slot;  // 0xffffffff = miss
  float u, v;
};

// Slack of the optional tests of the fast mode (a relative 2^-12 plus a small
// absolute term).  The reference compares nothing against a running best, so any
// pruning must never drop a triangle whose computed t could still win.
#define PT_SLACK_REL 2.44140625e-4f
#define PT_SLACK_ABS 2.0e-5f

// Möller–Trumbore exactly as Triangle::intersect evaluates it (object.cpp:406-429).
// Returns true when the reference would return a hit.
__device__ __forceinline__ bool tri_test(const ptl::TriRec *__restrict__ tris, uint32_t slot, V3 o, V3 d,
                                         float &t, float &u, float &v) {
  const float4 *p = reinterpret_cast<const float4 *>(tris + slot);
  const float4 a = p[0], b = p[1], c = p[2];
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

// Walks the tree for one ray.  MODE: PT_TRACE_EXACT reaches exactly the leaves the
// reference reaches; PT_TRACE_FAST additionally skips leaves and groups whose box
// the ray misses (with slack) and subtrees that start beyond the best hit (with
// slack), and descends near child first.  ANY: stop at the first accepted triangle.
// `stk` is this lane's column of the LDS stack (entries `stride` apart).
template <int MODE, bool ANY, bool COUNT>
__device__ __forceinline__ bool walk(const SceneDev &S, V3 o, V3 d, uint32_t *stk, int stride, HitRec &best,
                                     uint32_t &n_boxes, uint32_t &n_tris) {
  const float qnan = __builtin_nanf("");
  // 1/d per axis as AABB::intersect forms it (object.cpp:183); an axis with d == 0 is
  // skipped there, which a NaN reciprocal reproduces: NaN never updates tmin/tmax.
  const float ix = d.x != 0.0f ? 1.0f / d.x : qnan;
  const float iy = d.y != 0.0f ? 1.0f / d.y : qnan;
  const float iz = d.z != 0.0f ? 1.0f / d.z : qnan;
  const bool sx = ix < 0.0f, sy = iy < 0.0f, sz = iz < 0.0f;
  best.t = __builtin_inff();
  best.slot = 0xffffffffu;
  best.u = 0.0f;
  best.v = 0.0f;
  float cull = __builtin_inff();
  uint32_t cur = 0;
  int sp = 0;
  for (;;) {
    if (!(cur & ptl::REF_LEAF)) {
      const float4 *p = reinterpret_cast<const float4 *>(S.nodes + cur);
      const float4 q0 = p[0], q1 = p[1], q2 = p[2];
      const uint4 q3 = *reinterpret_cast<const uint4 *>(p + 3);
      if (COUNT) n_boxes += 2;
      // child 0: lo = q0.xyz, hi = (q0.w, q1.x, q1.y); child 1: lo = (q1.z, q1.w, q2.x), hi = q2.yzw
      float tn0 = 1e-5f, tf0 = __builtin_inff(), tn1 = 1e-5f, tf1 = __builtin_inff();
      {
        const float a0 = ((sx ? q0.w : q0.x) - o.x) * ix, b0 = ((sx ? q0.x : q0.w) - o.x) * ix;
        const float a1 = ((sx ? q2.y : q1.z) - o.x) * ix, b1 = ((sx ? q1.z : q2.y) - o.x) * ix;
        tn0 = fmaxf(tn0, a0); tf0 = fminf(tf0, b0);
        tn1 = fmaxf(tn1, a1); tf1 = fminf(tf1, b1);
      }
      {
        const float a0 = ((sy ? q1.x : q0.y) - o.y) * iy, b0 = ((sy ? q0.y : q1.x) - o.y) * iy;
        const float a1 = ((sy ? q2.z : q1.w) - o.y) * iy, b1 = ((sy ? q1.w : q2.z) - o.y) * iy;
        tn0 = fmaxf(tn0, a0); tf0 = fminf(tf0, b0);
        tn1 = fmaxf(tn1, a1); tf1 = fminf(tf1, b1);
      }
      {
        const float a0 = ((sz ? q1.y : q0.z) - o.z) * iz, b0 = ((sz ? q0.z : q1.y) - o.z) * iz;
        const float a1 = ((sz ? q2.w : q2.x) - o.z) * iz, b1 = ((sz ? q2.x : q2.w) - o.z) * iz;
        tn0 = fmaxf(tn0, a0); tf0 = fminf(tf0, b0);
        tn1 = fmaxf(tn1, a1); tf1 = fminf(tf1, b1);
      }
      const uint32_t k0 = q3.z, k1 = q3.w;
      bool p0, p1;
      if (MODE == PT_TRACE_EXACT) {
        p0 = (k0 == KIND_INTERNAL) ? (tf0 >= tn0) : (k0 != KIND_EMPTY);
        p1 = (k1 == KIND_INTERNAL) ? (tf1 >= tn1) : (k1 != KIND_EMPTY);
      } else {
        const bool soft0 = tf0 + ((fabsf(tf0) + tn0) * PT_SLACK_REL + PT_SLACK_ABS) >= tn0;
        const bool soft1 = tf1 + ((fabsf(tf1) + tn1) * PT_SLACK_REL + PT_SLACK_ABS) >= tn1;
        p0 = (k0 == KIND_INTERNAL) ? (tf0 >= tn0) : (k0 != KIND_EMPTY && soft0);
        p1 = (k1 == KIND_INTERNAL) ? (tf1 >= tn1) : (k1 != KIND_EMPTY && soft1);
        if (!ANY) {
          p0 = p0 && !(tn0 > cull);
          p1 = p1 && !(tn1 > cull);
        }
      }
      if (p0 && p1) {
        const bool swap = (MODE == PT_TRACE_FAST && !ANY) ? (tn1 < tn0) : false;
        stk[sp * stride] = swap ? q3.x : q3.y;
        ++sp;
        cur = swap ? q3.y : q3.x;
        continue;
      }
      if (p0) { cur = q3.x; continue; }
      if (p1) { cur = q3.y; continue; }
    } else {
      const uint32_t first = cur & ptl::REF_FIRST_MASK;
      const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
      for (uint32_t k = 0; k < cnt; ++k) {
        float t, u, v;
        if (COUNT) ++n_tris;
        if (tri_test(S.tris, first + k, o, d, t, u, v)) {
          if (ANY) return true;
          // larger slot wins a tie: the reference's leftmost-in-leaf / right-sibling rule
          if (t < best.t || (t == best.t && (int32_t)(first + k) > (int32_t)best.slot)) {
            best.t = t;
            best.slot = first + k;
            best.u = u;
            best.v = v;
            if (MODE == PT_TRACE_FAST) cull = t + (t * PT_SLACK_REL + PT_SLACK_ABS);
          }
        }
      }
    }
    if (sp == 0) break;
    --sp;
    cur = stk[sp * stride];
  }
  return best.slot != 0xffffffffu;
}

}  // namespace ptd
#endif

// pt_kernels.h — the gfx950 kernels of the render hot path (included once, by pt_gpu.hip, which
// holds the launch code and the C ABI).  Wavefront organisation: every stage is its own kernel
// over a structure-of-arrays queue of live paths; one lane carries one ray.
#ifndef PT_KERNELS_H
#define PT_KERNELS_H

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "pt_device.h"

using namespace ptd;



static const int kBlock = 256;  // block size of the streaming kernels (ray generation, resolve)
static const int kScanBlock = 1024;
static const int kScanPer = 4;
static const int kScanSpan = kScanBlock * kScanPer;  // chunks one workgroup of k_scan_local covers

// Path buffers are arrays of 16-byte records, one record per path and array, handed to the kernels as one
// base pointer and the distance between consecutive arrays.  A wave reads or writes one array as 64
// consecutive float4 (1 KiB, fully coalesced, one dwordx4 instruction); thirteen separate float arrays
// took thirteen dword instructions per path and as many array base addresses, which did not fit the scalar
// registers: the shading kernel re-read 160 spilled scalars per chunk through v_readlane.

struct QueueDev {  // one live path per index: three float4 arrays and one float2 array, `stride` paths apart
  float *base;
  uint32_t stride;
  // A = (o.x, o.y, o.z, id)   B = (d.x, d.y, d.z, -)   C = (T.x, T.y, T.z, L.x)   D = (L.y, L.z)
  // The closest-hit kernel reads A and B only; the shading kernel reads C and D after its visibility walks.
  __device__ __forceinline__ float4 *A() const { return reinterpret_cast<float4 *>(base); }
  __device__ __forceinline__ float4 *B() const { return reinterpret_cast<float4 *>(base) + stride; }
  __device__ __forceinline__ float4 *C() const { return reinterpret_cast<float4 *>(base) + 2 * (size_t)stride; }
  __device__ __forceinline__ float2 *D() const { return reinterpret_cast<float2 *>(base + 12 * (size_t)stride); }
  __device__ __forceinline__ uint32_t id(size_t i) const { return __float_as_uint(base[4 * i + 3]); }
  __device__ __forceinline__ void load_ray(size_t i, V3 &o, V3 &d, uint32_t &id) const {
    const float4 a = A()[i], b = B()[i];
    o = mk(a.x, a.y, a.z);
    d = mk(b.x, b.y, b.z);
    id = __float_as_uint(a.w);
  }
  __device__ __forceinline__ void load_sums(size_t i, V3 &T, V3 &L) const {
    const float4 c = C()[i];
    const float2 e = D()[i];
    T = mk(c.x, c.y, c.z);
    L = mk(c.w, e.x, e.y);
  }
  __device__ __forceinline__ void store(size_t i, V3 o, V3 d, V3 T, V3 L, uint32_t id) const {
    A()[i] = make_float4(o.x, o.y, o.z, __uint_as_float(id));
    B()[i] = make_float4(d.x, d.y, d.z, 0.0f);
    C()[i] = make_float4(T.x, T.y, T.z, L.x);
    D()[i] = make_float2(L.y, L.z);
  }
};
static const size_t kQueueBytesPerPath = 56;

struct HitDev {  // closest hit of every queued ray
  float *base;
  uint32_t stride;
  // H = (t, n.x, n.y, n.z): n = Triangle::normal() of the hit triangle, un-normalised
  // M = material id     U = (u, v, triangle record, -): written and read only when the scene has a textured material
  __device__ __forceinline__ float4 *H() const { return reinterpret_cast<float4 *>(base); }
  __device__ __forceinline__ uint32_t *M() const { return reinterpret_cast<uint32_t *>(base + 4 * (size_t)stride); }
  __device__ __forceinline__ float4 *U() const { return reinterpret_cast<float4 *>(base + 8 * (size_t)stride); }
  __device__ __forceinline__ float t(size_t i) const { return base[4 * i]; }
};
static const size_t kHitBytesPerPath = 16 + 16 + 16;  // M's array is padded to 16 bytes per path so that U stays 16-byte aligned

struct RadDev {  // finished radiance per (sample in batch, pixel): (r, g, b, -)
  float *base;
  uint32_t stride;
  __device__ __forceinline__ float4 *R() const { return reinterpret_cast<float4 *>(base); }
};
static const size_t kRadBytesPerPath = 16;

// What the per-path kernels need of a frame (the camera part stays in FrameDev for k_raygen).
struct PathFrame {
  uint32_t P;            // W*H
  uint64_t seed;
  uint32_t sample0;      // global index of the batch's first sample
  uint32_t stride;       // distance between consecutive samples' global indices
  double rr_prob;        // camera.cpp:118: a double
  float rr_prob_f;       // narrowed, for the weight of the cosine-sampling extension
  int rr_min_depth, max_depth, sampling;
  float bg[3];
  float pdf_f;           // (float)(1/(2*pi) * rr_prob), camera.cpp:152
  float pi_f;            // (float)M_PI, camera.cpp:122
  uint32_t dims_per_vertex;
};

struct FrameDev {
  int W, H;
  uint32_t P;            // W*H
  float origin[3];
  double rot[9];         // rotate() coefficients, math.cpp:130-140, row-major
  float vw, neg_vh;      // (float)vw, (float)(-vh)
  float pin_z;           // third component of point_towards before the rotation
  float aperture, focus;
  uint64_t seed;
  uint32_t sample0;      // global index of the batch's first sample
  uint32_t stride;       // distance between consecutive samples' global indices
  double rr_prob;        // camera.cpp:118: a double
  float rr_prob_f;       // narrowed, for the weight of the cosine-sampling extension
  int rr_min_depth, max_depth, sampling;
  float bg[3];
  float pdf_f;           // (float)(1/(2*pi) * rr_prob), camera.cpp:152
  float pi_f;            // (float)M_PI, camera.cpp:122
  uint32_t dims_per_vertex;
  uint32_t pixel_major;  // measuring knob: the batch's paths queued pixel by pixel (a pixel's samples side by side)
};

// What a traversal kernel keeps in LDS (see LdsScene) and how many words of the
// per-launch counter block it uses as its work-queue head.
struct LdsPlan {
  uint32_t n_nodes;    // top nodes staged in LDS
  uint32_t n_tris;     // 0, or all triangles
  uint32_t n_gates;    // 0, or all gate boxes (fast tree only)
  uint32_t stack_lds;  // stack entries per lane kept in LDS
  uint32_t *spill;     // global spill area for deeper stacks (null when none is needed)
};

__device__ __forceinline__ V3 rotate9(const double *c, V3 p) {
  const double px = p.x, py = p.y, pz = p.z;
  V3 r;
  r.x = (float)((c[0] * px) + (c[1] * py) + (c[2] * pz));
  r.y = (float)((c[3] * px) + (c[4] * py) + (c[5] * pz));
  r.z = (float)((c[6] * px) + (c[7] * py) + (c[8] * pz));
  return r;
}

// sin and cos of an angle in [0, 2*pi] in double precision, for the reference's cos(2*M_PI*y) / sin(2*M_PI*y)
// (camera.cpp:83-84).  The device library's sin() and cos() each reduce the argument for any magnitude and cost
// some hundred instructions apiece; here one reduction by multiples of pi/2 (two-constant Cody-Waite, exact
// for the five multiples that occur) serves both, followed by the classic degree-13/14 kernels on
// [-pi/4, pi/4] (error under one unit in the last place of a double).  The results are multiplied by a float
// and narrowed to float, so a last-place difference from the host's libm shows in about one float out of 2^29.
__device__ __forceinline__ void sincos_0_2pi(double a, double &sn, double &cs) {
  const double q = __builtin_rint(a * 6.36619772367581382433e-01);        // a * 2/pi
  double r = __builtin_fma(-q, 1.57079632679489655800e+00, a);             // pi/2, high part
  r = __builtin_fma(-q, 6.12323399573676603587e-17, r);                    // pi/2, low part
  const double z = r * r;
  const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * pc);
  const int n = (int)q & 3;
  sn = (n & 1) ? c : s;
  cs = (n & 1) ? s : c;
  if (n & 2) sn = -sn;
  if ((n + 1) & 2) cs = -cs;
}

__device__ __forceinline__ void camera_ray(const FrameDev &F, int px, int py, PathRng &rng, V3 &o, V3 &d) {
  // camera.cpp:162-167: float arithmetic on x and y, z prepared in double by the host
  const float x = ((float)px - (float)F.W / 2) * F.vw / (float)F.W;
  const float y = ((float)py - (float)F.H / 2) * F.neg_vh / (float)F.H;
  const V3 dc = mk(x, y, F.pin_z);
  if (F.aperture > 0.0f) {  // thin lens: not in the reference
    const float u1 = rng.real(0.0f, 1.0f, 0), u2 = rng.real(0.0f, 1.0f, 1);
    const float r = F.aperture * sqrtf(u1);
    const double ph = 2 * M_PI * (double)u2;
    double sp, cp;
    sincos_0_2pi(ph, sp, cp);
    const V3 lens = mk((float)(cp * (double)r), (float)(sp * (double)r), 0.0f);
    const float s = F.focus / dc.z;
    const V3 pf = scale(dc, s);
    const float inv_s = dc.z / F.focus;
    const V3 dl = scale(sub(pf, lens), inv_s);
    o = add(mk(F.origin[0], F.origin[1], F.origin[2]), rotate9(F.rot, lens));
    d = rotate9(F.rot, dl);
  } else {
    o = mk(F.origin[0], F.origin[1], F.origin[2]);
    d = rotate9(F.rot, dc);
  }
}

__global__ __launch_bounds__(kBlock) void k_raygen(FrameDev F, uint32_t n, QueueDev q) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // a path's id is sample-in-batch * P + pixel whatever its place in the queue: radiance is written by id
  uint32_t s = i / F.P, p = i - s * F.P;
  if (F.pixel_major) {
    const uint32_t sb = n / F.P;
    p = i / sb;
    s = i - p * sb;
  }
  const int py = (int)(p / (uint32_t)F.W), px = (int)(p - (uint32_t)py * (uint32_t)F.W);
  PathRng rng;
  rng.init(F.seed, p, F.sample0 + s * F.stride);
  V3 o, d;
  camera_ray(F, px, py, rng, o, d);
  q.store(i, o, d, mk(1.0f, 1.0f, 1.0f), mk(0.0f, 0.0f, 0.0f), s * F.P + p);
}

// Camera::point_towards for every pixel, for callers that only want the directions.
__global__ __launch_bounds__(kBlock) void k_camera_dirs(FrameDev F, float *dir) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= F.P) return;
  const int py = (int)(p / (uint32_t)F.W), px = (int)(p - (uint32_t)py * (uint32_t)F.W);
  PathRng rng;
  rng.init(F.seed, p, 0);
  FrameDev G = F;
  G.aperture = 0.0f;
  V3 o, d;
  camera_ray(G, px, py, rng, o, d);
  dir[3 * (size_t)p + 0] = d.x;
  dir[3 * (size_t)p + 1] = d.y;
  dir[3 * (size_t)p + 2] = d.z;
}

__device__ __forceinline__ void wave_add64(unsigned long long *dst, uint32_t v) {
  uint32_t s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, (unsigned long long)s);
}

// The traversal kernels are persistent: a launch holds at most as many workgroups as the
// chip keeps resident, every workgroup stages its part of the scene into LDS once, and
// every wave then pulls chunks of 64 rays from a shared counter until none are left.

// The LDS copy is split by quarter: quarter k of node i sits at nodes[k*n_nodes + i] (and
// third k of triangle i at tris[k*n_tris + i]).  A wave's 64 lanes then read 16-byte pieces
// whose bank depends on i mod 16; with whole 64-byte records side by side only 4 of the 16
// bank groups would ever be hit by one instruction.
// The fast tree's nodes are staged as 7 pieces (6 plane pieces + the references), its gate boxes as 2.
template <int BLOCK, int MODE>
__device__ __forceinline__ LdsScene stage_scene(const SceneDev &S, LdsPlan plan, uint32_t &stack_base) {
  constexpr bool WIDE = mode_wide(MODE), OCT = mode_oct(MODE);
  constexpr uint32_t NP = OCT ? 5u : (WIDE ? 7u : 4u);   // node pieces kept in LDS
  constexpr uint32_t NG = WIDE ? 8u : 4u;   // float4 per node in global memory
  const float4 *gn = OCT ? reinterpret_cast<const float4 *>(S.nodes8)
                         : (WIDE ? reinterpret_cast<const float4 *>(S.nodes4) : reinterpret_cast<const float4 *>(S.nodes));
  const float4 *gt = WIDE ? reinterpret_cast<const float4 *>(S.tris4) : reinterpret_cast<const float4 *>(S.tris);
  LdsScene C;
  C.node_base = 0;
  C.tri_base = NP * plan.n_nodes;
  C.gate_base = C.tri_base + 3u * plan.n_tris;
  C.n_nodes = plan.n_nodes;
  C.n_tris = plan.n_tris;
  C.n_gates = WIDE ? plan.n_gates : 0u;
  C.stack_lds = plan.stack_lds;
  C.spill_stride = gridDim.x * BLOCK;
  C.spill = plan.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (uint32_t i = threadIdx.x; i < NG * plan.n_nodes; i += BLOCK) {
    const uint32_t piece = i % NG, node = i / NG;
    if (piece < NP) pt_smem[piece * plan.n_nodes + node] = gn[i];
  }
  for (uint32_t i = threadIdx.x; i < 3u * plan.n_tris; i += BLOCK)
    pt_smem[C.tri_base + (i % 3u) * plan.n_tris + (i / 3u)] = gt[i];
  if (WIDE) {
    const float4 *gg = reinterpret_cast<const float4 *>(S.gates);
    for (uint32_t i = threadIdx.x; i < 2u * C.n_gates; i += BLOCK) pt_smem[C.gate_base + (i & 1u) * C.n_gates + (i >> 1)] = gg[i];
  }
  __syncthreads();
  stack_base = 4u * (C.gate_base + 2u * C.n_gates) + threadIdx.x;  // in 32-bit words
  return C;
}

// Work is claimed in runs of `len` chunks of 64 items.  One returning atomic on a single
// word sustains under 100 claims per microsecond chip-wide, so a claim per 64 rays would cap
// a large launch at about 6 G rays/s on its own; a long run, on the other hand, serialises a
// small launch on few waves.  The host picks `len` per launch (run_length()).
// Returns the first chunk of the next run for this wave (all lanes must be active).
__device__ __forceinline__ uint32_t next_run(uint32_t *head, uint32_t len) {
  uint32_t c = 0;
  if ((threadIdx.x & 63) == 0) c = atomicAdd(head, len);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
}

// Length of a wave's next run after it has claimed chunks [run, run + len) of n_chunks: a share of what is left
// (as far as this wave knows: the head has moved on since), never more than the launch's first run length.
// Runs of one fixed length leave the waves finishing up to a whole run apart, a quarter of the launch at the
// lengths the host picks; with shrinking runs they finish within a chunk or two of each other.
__device__ __forceinline__ uint32_t next_len(uint64_t n_chunks, uint32_t run, uint32_t len, uint32_t first_len, uint32_t waves) {
  const uint64_t end = (uint64_t)run + len;
  const uint64_t left = n_chunks > end ? n_chunks - end : 0;
  const uint64_t share = left / (2ull * waves);
  return share < 1 ? 1u : (share > first_len ? first_len : (uint32_t)share);
}

// Walks one ray per work item for items [0, n), persistent over the whole launch.
//   fetch(i, o, d)  loads item i's ray; returns false when the item needs no walk (it then has
//                   written whatever the item's output is); it may take i by reference and replace it by
//                   the index retire is to get (all lanes that fetch together hold consecutive items);
//   retire(i, st)   stores the result of a walked item.
//
// RES (scene in LDS): walks are short and cost no memory latency; a wave walks whole chunks of 64
// items, all lanes together.  Handing out items lane by lane was measured slower there: its
// bookkeeping runs with few lanes active.
// Otherwise rays differ widely in the number of steps and every step waits on memory, so a wave
// does not walk 64 rays in lockstep until the slowest is done (measured: 42 % of the lanes active).
// Each lane carries its own ray; when S.refill lanes are idle the wave stores their results and
// hands them the next items of the run of chunks it has claimed, claiming another run when that
// one is used up (1.8 times the rays per second on a 260 k-triangle scene).
template <int MODE, bool ANY, bool COUNT, bool WANT_N, bool RES, bool REFILL, int BLOCK, typename Fetch, typename Retire>
__device__ __forceinline__ void walk_items(const SceneDev &S, const LdsScene &C, uint32_t stk, uint64_t n, uint32_t kRun,
                                           uint32_t *head, uint32_t &nb, uint32_t &nt, Fetch fetch, Retire retire) {
  if constexpr (!REFILL) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_chunks = (n + 63u) / 64u;
    const uint32_t waves = gridDim.x * (BLOCK / 64);
    uint32_t len = kRun;
    for (;;) {
      const uint32_t run = next_run(head, len);
      if ((uint64_t)run * 64u >= n) break;
      const uint32_t this_len = len;
      len = next_len(n_chunks, run, len, kRun, waves);
      for (uint32_t c = 0; c < this_len; ++c) {
        const uint64_t base = ((uint64_t)run + c) * 64u;
        if (base >= n) break;
        uint64_t i = base + lane;  // fetch may replace the item's number by the place of its result
        V3 o, d;
        if (i < n && fetch(i, o, d)) {
          RayState st;
          walk_init<MODE, WANT_N, RES>(S, C, st, o, d);
          do {
            walk_step<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, BLOCK, nb, nt, S.quorum);
          } while (__ballot(!st.done()) != 0);
          retire(i, st);
        }
      }
    }
  } else {
    uint64_t pool = 0, pool_end = 0;  // items of the claimed run not handed out yet (same in all lanes)
    bool more = true;                 // the queue may hold further runs
    const uint64_t NONE = ~0ull;
    uint64_t mine = NONE;             // item this lane works on or has finished
    bool busy = false;
    const uint64_t n_chunks = (n + 63u) / 64u;
    const uint32_t waves = gridDim.x * (BLOCK / 64);
    uint32_t len = kRun;
    RayState st;
    st.live = false;
    st.pend = 0;
    st.meta = 0;
    st.gw = 0;
#ifdef PT_DEBUG_ROUNDS
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_te = 0;
#endif
    for (;;) {
      const unsigned long long m_busy = __ballot(busy);
      const uint32_t n_idle = 64u - (uint32_t)__popcll(m_busy);
      const bool can_fetch = more || pool < pool_end;
      uint32_t n_idle_now = n_idle;
      if ((n_idle >= (ANY ? S.refill_any : S.refill) && can_fetch) || m_busy == 0) {
        if (!busy && mine != NONE) {
          retire(mine, st);
          mine = NONE;
        }
        if (pool >= pool_end && more) {
          const uint32_t run = next_run(head, len);
          if ((uint64_t)run * 64u >= n) {
            more = false;
#ifdef PT_DEBUG_ROUNDS
            dbg_te = wall_clock64();
#endif
          } else {
            pool = (uint64_t)run * 64u;
            const uint64_t e = ((uint64_t)run + len) * 64u;
            pool_end = e < n ? e : n;
            len = next_len(n_chunks, run, len, kRun, waves);
          }
        }
        if (!busy) {
          const unsigned long long idle = ~m_busy;
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
          uint64_t i = pool + rank;  // fetch may replace the item's number by the place of its result
          V3 o, d;
          if (i < pool_end && fetch(i, o, d)) {
            walk_init<MODE, WANT_N, RES>(S, C, st, o, d);
            mine = i;
            busy = true;
          }
        }
        pool = pool_end - pool > n_idle ? pool + n_idle : pool_end;
        const unsigned long long m_now = __ballot(busy);
#ifdef PT_DEBUG_ROUNDS
        if ((threadIdx.x & 63) == 0) {
          atomicAdd(&ptd::g_dbg[(ANY ? 8 : 0) + 6], 1ull);
          atomicAdd(&ptd::g_dbg[(ANY ? 8 : 0) + 7], (unsigned long long)__popcll(m_now & ~m_busy));
        }
#endif
        if (m_now == 0) {
          if (!more && pool >= pool_end) {
#ifdef PT_DEBUG_ROUNDS
            if ((threadIdx.x & 63) == 0) {  // wave time after the queue ran dry, and in all (100 MHz ticks)
              const unsigned long long t1 = wall_clock64();
              atomicAdd(&ptd::g_dbg[ANY ? 18 : 16], t1 - (dbg_te ? dbg_te : t1));
              atomicAdd(&ptd::g_dbg[ANY ? 19 : 17], t1 - dbg_t0);
            }
#endif
            break;
          }
          continue;  // nothing to walk among these items, or the run is used up: fetch again
        }
        n_idle_now = 64u - (uint32_t)__popcll(m_now);
      }
      // the quorum for testing parked leaves counts against the lanes that have a ray at all
      walk_step<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, BLOCK, nb, nt, ((64u - n_idle_now) * S.quorum) >> 6);
      if (busy && st.done()) busy = false;
    }
  }
}

// Does the path of queue entry i go on after a vertex whose closest hit is at t?  It does iff it
// hit something and passes Russian roulette (camera.cpp:65, 118-119).
__device__ __forceinline__ bool survives_id(const SceneDev &S, const PathFrame &F, int depth, uint32_t id, float t) {
  if (t == __builtin_inff()) return false;
  if (F.max_depth > 0 && depth >= F.max_depth) return false;
  if (depth <= F.rr_min_depth) return true;  // the roulette draw exists only when depth > rr_min_depth
  PathRng rng;
  rng.init(F.seed, id % F.P, F.sample0 + (id / F.P) * F.stride);
  const uint32_t dim = 2u + (uint32_t)depth * F.dims_per_vertex + 3u * S.n_lights;
  return roulette_pass(F.rr_prob, rng.real(0.0f, 1.0f, dim));
}
__device__ __forceinline__ bool survives(const SceneDev &S, const PathFrame &F, int depth, const QueueDev &q, uint32_t i,
                                         float t) {
  if (t == __builtin_inff()) return false;
  if (F.max_depth > 0 && depth >= F.max_depth) return false;
  if (depth <= F.rr_min_depth) return true;
  return survives_id(S, F, depth, q.id(i), t);
}

// Closest hit for every queued ray.  The kernel also decides which paths go on after this
// vertex and leaves that as one byte per ray in `alive_out`, from which k_scan_local forms the
// 64-bit mask of each chunk, so that the survivors' places in the next queue are known before
// the shading kernel writes them.
#ifndef PT_TRACE_OCC
#define PT_TRACE_OCC 6
#endif
// Outside LDS in 256-thread workgroups the walk waits on memory and a fifth wave per SIMD is worth more than
// the registers beyond 96 (measured: the same kernel at 4 waves per SIMD walks a quarter fewer rays per second).
template <int MODE, int BLOCK, bool COUNT, bool RES, bool REFILL>
__global__ __launch_bounds__(BLOCK, (!RES && BLOCK == 256 && mode_wide(MODE) && !COUNT) ? PT_TRACE_OCC : 1) void k_trace_queue(SceneDev S, LdsPlan plan, uint32_t kRun, PathFrame F, int depth,
                                                       uint32_t n, QueueDev q, HitDev hit, uint8_t *alive_out,
                                                       uint32_t *head, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, COUNT, true, RES, REFILL, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        uint32_t id;
        q.load_ray(i, o, d, id);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        HitRec h = st.best;
        hit_finish<MODE, RES>(S, C, st.o, st.d, h, S.n_tex != 0, true);
        hit.H()[i] = make_float4(h.t, h.n.x, h.n.y, h.n.z);
        hit.M()[i] = h.mat;
        if (S.n_tex) hit.U()[i] = make_float4(h.u, h.v, __uint_as_float(h.slot), 0.0f);
        alive_out[i] = survives(S, F, depth, q, (uint32_t)i, h.t) ? 1 : 0;
      });
  if (COUNT) {
    wave_add64(stats + 0, nb);
    wave_add64(stats + 1, nt);
  }
}

// Which of a caller's rays a launch of the ray-query kernels walks.  The pruning walks are exact for origins
// within `limit` of the world origin (8 times the scene's largest coordinate: their margins are relative to
// that coordinate, and the rounding error of a box test grows with the origin's distance), so a query in a
// pruning mode is two launches: the pruning walk for the rays inside that range (SIDE_NEAR) and the exact walk
// for the others (SIDE_FAR, which includes origins that are not finite).  A ray the launch leaves out costs it
// the read of its origin.
enum : int { SIDE_ALL = 0, SIDE_NEAR = 1, SIDE_FAR = 2 };
struct RaySide {
  int side;
  float limit;
  __device__ __forceinline__ bool takes(V3 o) const {
    if (side == SIDE_ALL) return true;
    const bool near = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= limit;  // false for NaN
    return side == SIDE_NEAR ? near : !near;
  }
};

template <int MODE, int BLOCK, bool COUNT, bool RES>
__global__ __launch_bounds__(BLOCK) void k_trace_api(SceneDev S, LdsPlan plan, uint32_t kRun, size_t n, const float *ox,
                                                     const float *oy, const float *oz, const float *dx,
                                                     const float *dy, const float *dz, float *t, int32_t *tri,
                                                     float *u, float *v, uint32_t *head, unsigned long long *stats,
                                                     RaySide side) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, COUNT, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
        if (!side.takes(o)) return false;
        d = mk(dx[i], dy[i], dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        HitRec h = st.best;
        hit_finish<MODE, RES>(S, C, st.o, st.d, h, u != nullptr || v != nullptr, false);
        t[i] = h.t;
        tri[i] = h.slot != 0xffffffffu ? (int32_t)(mode_wide(MODE) ? S.tri_id4[h.slot] : S.tris[h.slot].tri_id) : -1;
        if (u) u[i] = h.u;
        if (v) v[i] = h.v;
      });
  if (COUNT) {
    wave_add64(stats + 0, nb);
    wave_add64(stats + 1, nt);
  }
}

template <int MODE, int BLOCK, bool RES>
__global__ __launch_bounds__(BLOCK) void k_any_api(SceneDev S, LdsPlan plan, uint32_t kRun, size_t n, const float *ox,
                                                   const float *oy, const float *oz, const float *dx, const float *dy,
                                                   const float *dz, uint8_t *hit, uint32_t *head, RaySide side) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, true, false, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
        if (!side.takes(o)) return false;
        d = mk(dx[i], dy[i], dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) { hit[i] = st.found ? 1 : 0; });
}

// HitData::hits (object.cpp:203, 281-282: "for special visualizations"): every internal node on
// the way back up from the winning triangle adds one, so for a hit it is the number of internal
// nodes of the reference between the root and the triangle's leaf (levels[slot], counted by the
// flattener).  For a miss the recursion hands up the right child's empty result at every internal
// node whose box the ray passes (`lhit.t < rhit.t ? lhit : rhit` with both infinite), and the
// first mesh's result in a top-level leaf: the count is the length of that chain.
__device__ __forceinline__ bool box_pass_exact(float lx, float ly, float lz, float hx, float hy, float hz, V3 o, V3 d) {
  const float nan = __builtin_nanf("");
  const float ix = d.x != 0.0f ? 1.0f / d.x : nan, iy = d.y != 0.0f ? 1.0f / d.y : nan, iz = d.z != 0.0f ? 1.0f / d.z : nan;
  const float ax = ((ix < 0.0f ? hx : lx) - o.x) * ix, bx = ((ix < 0.0f ? lx : hx) - o.x) * ix;
  const float ay = ((iy < 0.0f ? hy : ly) - o.y) * iy, by = ((iy < 0.0f ? ly : hy) - o.y) * iy;
  const float az = ((iz < 0.0f ? hz : lz) - o.z) * iz, bz = ((iz < 0.0f ? lz : hz) - o.z) * iz;
  const float tn = fmaxf(fmaxf(fmaxf(1e-5f, ax), ay), az);
  const float tf = fminf(fminf(fminf(__builtin_inff(), bx), by), bz);
  return tf >= tn;  // the same planes, operations and NaN behaviour as the node step of the exact walk
}

__device__ __forceinline__ int32_t miss_chain_hits(const SceneDev &S, V3 o, V3 d) {
  const float4 *nodes = reinterpret_cast<const float4 *>(S.nodes);
  // node 0 is the super-root: its child 0 is the reference's root
  float4 q0 = nodes[0], q1 = nodes[1], q2 = nodes[2], q3 = nodes[3];
  uint32_t kind = __float_as_uint(q3.z), ref = __float_as_uint(q3.x);
  float lx = q0.x, ly = q0.y, lz = q0.z, hx = q0.w, hy = q1.x, hz = q1.y;
  int32_t hits = 0;
  for (;;) {
    if (kind == KIND_INTERNAL) {
      if (!box_pass_exact(lx, ly, lz, hx, hy, hz, o, d)) break;
      ++hits;
    } else if (kind != KIND_GROUP) {
      break;  // a leaf of the reference (whole or split): its empty result carries 0
    }
    const size_t at = 4u * (size_t)ref;
    q1 = nodes[at + 1]; q2 = nodes[at + 2]; q3 = nodes[at + 3];
    kind = __float_as_uint(q3.w); ref = __float_as_uint(q3.y);  // child 1: the right child / the first mesh
    lx = q1.z; ly = q1.w; lz = q2.x; hx = q2.y; hy = q2.z; hz = q2.w;
  }
  return hits;
}

template <int MODE, int BLOCK, bool RES>
__global__ __launch_bounds__(BLOCK) void k_hits_api(SceneDev S, LdsPlan plan, uint32_t kRun, size_t n, const float *ox,
                                                    const float *oy, const float *oz, const float *dx, const float *dy,
                                                    const float *dz, const uint8_t *levels, int32_t *hits,
                                                    uint32_t *head, RaySide side) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, false, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
        if (!side.takes(o)) return false;
        d = mk(dx[i], dy[i], dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        hits[i] = st.best.slot != 0xffffffffu ? (int32_t)levels[st.best.slot] : miss_chain_hits(S, st.o, st.d);
      });
}

// Visibility rays of the split shading pipeline (scenes outside LDS), prepared by k_shade<..., 1>.  The ray from
// vertex i towards its sample of light l is kept at at(i, l, L): the vertices in chunks of 64, a chunk's rays light
// by light, so that the shading phases (one vertex per lane, one light after the other) read and write whole
// lines.  k_shadow_walk hands the rays out vertex by vertex (walk item j = i*L + l): the rays of one vertex leave
// from one point and reach their leaves together, which keeps more lanes in step than neighbouring vertices'
// rays towards one light do (1 M triangles, 64 lights: 6.96 -> 8.4 G rays/s at 4 samples per pixel; one light: the
// same order).  Its fetch gathers a vertex's directions from L lines that the chunk's other vertices share;
// transposing them in the shading phase (an LDS tile per wave, a vertex's directions side by side) was measured
// slower: 32-byte pieces written per vertex cost that phase more than the walk gains.
struct ShadowDev {
  float *vbase;            // per vertex: hit point x, y, z; arrays vstride floats apart
  uint32_t vstride;
  float *ibase;            // per item: direction towards the light sample (3 arrays), then what the light adds if
  size_t istride;          // the ray hits anything (camera.cpp:101-114; 3 arrays); arrays istride floats apart
  uint8_t *vis_;           // per item: the ray hit something
  __device__ __forceinline__ float *x() const { return vbase; }
  __device__ __forceinline__ float *y() const { return vbase + vstride; }
  __device__ __forceinline__ float *z() const { return vbase + 2 * (size_t)vstride; }
  __device__ __forceinline__ float *tx() const { return ibase; }
  __device__ __forceinline__ float *ty() const { return ibase + istride; }
  __device__ __forceinline__ float *tz() const { return ibase + 2 * istride; }
  __device__ __forceinline__ float *wx() const { return ibase + 3 * istride; }
  __device__ __forceinline__ float *wy() const { return ibase + 4 * istride; }
  __device__ __forceinline__ float *wz() const { return ibase + 5 * istride; }
  __device__ __forceinline__ uint8_t *vis() const { return vis_; }
  __device__ __forceinline__ static size_t at(uint32_t i, uint32_t l, uint32_t L) { return ((size_t)(i >> 6) * L + l) * 64u + (i & 63u); }
  // the answers, one byte each, vertex by vertex: the walk stores them one lane at a time, and a vertex's bytes
  // side by side merge in L2 (light by light they went out as a 32-byte sector per byte)
  __device__ __forceinline__ static size_t vis_at(uint32_t i, uint32_t l, uint32_t L) { return (size_t)i * L + l; }
  // (vertex, light) pairs the arrays hold for n vertices: whole chunks
  __host__ __device__ __forceinline__ static uint64_t items(uint32_t n, uint32_t L) { return (uint64_t)((n + 63u) & ~63u) * L; }
  // work items of the walk.  In whole-vertex mode a wave's 64 lanes are filled with whole vertices only: 64/L
  // vertices side by side (L <= 64, the lanes left over stay empty) or one vertex in ceil(L/64) fills.
  __host__ __device__ __forceinline__ static uint64_t walk_items(uint32_t n, uint32_t L, bool whole) {
    const uint64_t np = (n + 63u) & ~63u;
    if (!whole || L == 0) return np * L;
    if (L <= 64u) {
      const uint32_t per_fill = 64u / L;
      return (np + per_fill - 1u) / per_fill * 64u;
    }
    return np * ((L + 63u) / 64u) * 64u;
  }
};

#ifndef PT_SHADOW_OCC
#define PT_SHADOW_OCC 6
#endif
// Outside LDS in 256-thread workgroups: the register count is capped for five waves per SIMD (96 registers).
template <int MODE, int BLOCK, bool COUNT, bool RES>
__global__ __launch_bounds__(BLOCK, (!RES && BLOCK == 256) ? PT_SHADOW_OCC : 1) void k_shadow_walk(SceneDev S, LdsPlan plan, uint32_t kRun, uint32_t n, HitDev hit,
                                                       ShadowDev sd, uint32_t *head, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, true, COUNT, false, RES, !RES, BLOCK>(
      S, C, stk, ShadowDev::walk_items(n, S.n_lights, S.cohort != 0u), kRun, head, nb, nt,
      [&](uint64_t &j, V3 &o, V3 &d) {
        const uint32_t L = S.n_lights;
        uint32_t i, l;
        if (L == 1u) {  // one light: item = vertex in every order
          i = (uint32_t)j;
          l = 0u;
        } else if (S.cohort) {
          // Whole vertices: the lanes that fetch together are one fill (the wave hands rays out only when all
          // of its lanes are idle, and runs start at multiples of 64), fill f = j / 64 the same for all of them.
          const uint32_t slot = (uint32_t)j & 63u;
          const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(j >> 38)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(j >> 6));
          if (L <= 64u) {
            const uint32_t per_fill = 64u / L, v = slot / L;
            if (v >= per_fill) return false;
            i = (uint32_t)(f * per_fill) + v;
            l = slot - v * L;
          } else {
            const uint32_t fills = (L + 63u) / 64u;
            const uint64_t v = f / fills;
            i = (uint32_t)v;
            l = (uint32_t)(f - v * fills) * 64u + slot;
            if (l >= L) return false;
          }
        } else if (S.vertex_major >= 2u) {
          // measuring knob: wave fills of (64/g vertices) x (g lights), g = S.vertex_major dividing 64 and L
          const uint32_t g = S.vertex_major, vb = 64u / g;
          const uint64_t per_blk = (uint64_t)vb * L;
          const uint64_t blk = j / per_blk;
          const uint32_t r = (uint32_t)(j - blk * per_blk);
          const uint32_t lg = r / 64u, r2 = r % 64u;
          i = (uint32_t)(blk * vb) + r2 / g;
          l = lg * g + r2 % g;
        } else if (S.vertex_major) {
          // j = i*L + l.  The lanes that fetch together hold consecutive items: the 64-bit division is done once
          // for the first of them, on the scalar unit, and a lane divides only its small distance from that item.
          const uint64_t j0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(j >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)j);
          const uint64_t i0 = j0 / L;
          const uint32_t a = (uint32_t)(j0 - i0 * L) + (uint32_t)(j - j0);  // < L + 64
          uint32_t q = (uint32_t)((float)a / (float)L);
          if (q * L > a) --q;
          else if ((q + 1u) * L <= a) ++q;
          i = (uint32_t)i0 + q;
          l = a - q * L;
        } else {  // light by light (measuring knob): j = l*(vertices in whole chunks) + i
          const uint32_t np = (n + 63u) & ~63u;
          l = (uint32_t)(j / np);
          i = (uint32_t)(j % np);
        }
        if (i >= n || hit.t(i) == __builtin_inff()) return false;  // no vertex, no light loop: nobody reads vis[k]
        const size_t k = ShadowDev::at(i, l, L);
        j = ShadowDev::vis_at(i, l, L);  // where retire puts the answer
        o = mk(sd.x()[i], sd.y()[i], sd.z()[i]);
        d = mk(sd.tx()[k], sd.ty()[k], sd.tz()[k]);
        return true;
      },
      [&](uint64_t k, const RayState &st) { sd.vis()[k] = st.found ? 1 : 0; });
  if (COUNT) {
    wave_add64(stats + 2, nb);
    wave_add64(stats + 3, nt);
  }
}

// random_direction, camera.cpp:77-90
__device__ __forceinline__ V3 uniform_dir(V3 n, float x, float y) {
  const float phi = 1.0f - 2.0f * x;
  const float theta = (float)fmax(0.0, (double)(1.0f - phi * phi));
  const float st = sqrtf(theta);
  const double ang = 2 * M_PI * (double)y;
  double sa, ca;
  sincos_0_2pi(ang, sa, ca);
  V3 v = mk((float)(ca * (double)st), (float)(sa * (double)st), phi);
  v = normalize(v);
  return dot(v, n) < 0.0f ? neg(v) : v;
}
// cosine-weighted direction about n: not in the reference
__device__ __forceinline__ V3 cosine_dir(V3 n, float x, float y) {
  const float r = sqrtf(x);
  const float lz = sqrtf(fmaxf(0.0f, 1.0f - x));
  const double ang = 2 * M_PI * (double)y;
  double sa, ca;
  sincos_0_2pi(ang, sa, ca);
  const float lx = (float)(ca * (double)r), ly = (float)(sa * (double)r);
  const float s = copysignf(1.0f, n.z);
  const float a = -1.0f / (s + n.z);
  const float b = n.x * n.y * a;
  const V3 t1 = mk(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
  const V3 t2 = mk(b, s + n.y * n.y * a, -n.y);
  return add(add(scale(t1, lx), scale(t2, ly)), scale(n, lz));
}

// static_cast<int>(float) as the reference's x86-64 build performs it (cvttss2si): out of range
// and NaN give INT_MIN.
__device__ __forceinline__ int cvt_int_x86(float f) {
  return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? (int)0x80000000 : (int)f;
}

// brdf of a vertex on a textured triangle, camera.cpp:124-146: texture coordinates interpolated
// with alpha = 1-u-v on vt0, beta = u on vt1, gamma = v on vt2 (HitData::barycentric is
// (u, v, 1-u-v), object.cpp:428), nearest texel, no flip, the texel itself (no 1/pi).
__device__ __forceinline__ V3 texel_brdf(const SceneDev &S, bool wide, uint32_t tex, uint32_t slot, float bu, float bv) {
  const float *uv = (wide ? S.tri_uv4 : S.tri_uv) + (size_t)slot * 6;  // `slot` is a record of the tree that was walked
  const float beta = bu, gamma = bv, alpha = 1 - bu - bv;
  const float tu = alpha * uv[0] + beta * uv[2] + gamma * uv[4];
  const float tv = alpha * uv[1] + beta * uv[3] + gamma * uv[5];
  const ptl::TexRec tr = S.textures[tex - 1u];
  const int w = (int)tr.width, h = (int)tr.height;
  int x = cvt_int_x86(tu * (float)w), y = cvt_int_x86(tv * (float)h);
  x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);  // std::clamp
  y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
  const uint8_t *px = S.texels + tr.offset + ((size_t)y * (size_t)w + (size_t)x) * 3;
  return divs(mk((float)px[0], (float)px[1], (float)px[2]), 255.0f);  // Image::at, object.cpp:128-134
}

// One light of the direct-light loop (camera.cpp:96-115) for the vertex at x with shading normal
// nrm: the direction `to` of the visibility ray and `w`, what the light adds to the vertex if that
// ray hits anything (the reference adds it when the ray hits *something*, camera.cpp:101-102).
__device__ __forceinline__ void sample_light(const SceneDev &S, PathRng &rng, uint32_t base, uint32_t l, V3 x,
                                             V3 nrm, V3 &to, V3 &w) {
  const float4 *lp = reinterpret_cast<const float4 *>(S.lights + l);
  const float4 l0 = lp[0], l1 = lp[1];
  const uint32_t first = __float_as_uint(l0.x), count = __float_as_uint(l0.y);
  const float area_f = l0.z, inv_area_f = l0.w;
  const V3 lke = mk(l1.x, l1.y, l1.z);
  // Polygon::sample, object.cpp:513-523: first triangle whose running area reaches r
  const double r = (double)rng.real(0.0f, area_f, base + 3u * l);
  uint32_t lo = 0, hi = count - 1;  // past the end the reference is undefined; take the last
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (r <= S.light_tris[first + mid].accum) hi = mid; else lo = mid + 1;
  }
  const float4 *qp = reinterpret_cast<const float4 *>(S.light_tris + first + lo);
  const float4 g0 = qp[0], g1 = qp[1], g2 = qp[2];
  const V3 lu = mk(g0.x, g0.y, g0.z), lv = mk(g0.w, g1.x, g1.y), ln = mk(g1.z, g1.w, g2.x);
  // Triangle::sample, object.cpp:432-442 (the point is relative to v0; v0 is not added)
  const double s = (double)rng.real(0.0f, 1.0f, base + 3u * l + 1u);
  const double tt = (double)rng.real(0.0f, 1.0f, base + 3u * l + 2u);
  V3 point;
  if (s + tt <= 1.0) point = add(scale(lu, (float)s), scale(lv, (float)tt));
  else point = add(scale(lu, (float)(1.0 - s)), scale(lv, (float)(1.0 - tt)));
  to = normalize(sub(point, x));
  float c = dot(to, nrm);
  if (c < 0.0f) c = 0.0f;
  float y = dot(to, ln);
  if (y < 0.0f) y = 0.0f;
  const double nn = (double)norm(to);
  y = (float)((double)y / (nn * nn));  // std::pow(norm, 2) in double, camera.cpp:111
  w = scale(scale(scale(lke, c), y), inv_area_f);
}

// The bounce of a path that goes on (camera.cpp:120-156): next direction w and the new throughput.
// m0 is the first float4 of the vertex's material; slot, bu, bv (triangle record and barycentrics of the
// hit) are read only for a textured material.
template <int MODE, bool TEX>
__device__ __forceinline__ void bounce(const SceneDev &S, const PathFrame &F, PathRng &rng, uint32_t base, V3 nrm, float4 m0,
                                       uint32_t slot, float bu, float bv, V3 &T, V3 &w) {
  const V3 kd = mk(m0.x, m0.y, m0.z);
  const float rx = rng.real(0.0f, 1.0f, base + 3u * S.n_lights + 1u);
  const float ry = rng.real(0.0f, 1.0f, base + 3u * S.n_lights + 2u);
  const uint32_t tex = __float_as_uint(m0.w);
  const bool textured = TEX && tex != 0;                                // camera.cpp:121
  V3 texel = mk(0, 0, 0);
  if (textured) texel = texel_brdf(S, mode_wide(MODE), tex, slot, bu, bv);
  if (F.sampling == PT_SAMPLE_COSINE) {
    w = cosine_dir(nrm, rx, ry);
    T = divs(mul(T, textured ? scale(texel, F.pi_f) : kd), F.rr_prob_f);
  } else {
    w = uniform_dir(nrm, rx, ry);
    const V3 brdf = textured ? texel : divs(kd, F.pi_f);                // camera.cpp:122-146
    T = divs(scale(mul(T, brdf), fabsf(dot(nrm, w))), F.pdf_f);         // camera.cpp:155-156
  }
}

// One path vertex per work item.  Which paths survive was decided by k_trace_queue and
// prefix-summed by k_scan_*, so a survivor is written straight to its place in `out`: chunk
// offset + its rank among the chunk's survivors (wave ballot mask + popcount).
// The compaction is stable: the queue stays in pixel order and no two launches differ.
// Claiming an output slot per wave with an atomic on one counter instead was measured at
// 260 ms per frame on its own: a single word takes under 100 atomics per microsecond.
// cnt[1] = paths that hit something, cnt[3] = this kernel's work-queue head (PHASE 0).
//
// PHASE 0: the whole vertex in one kernel, visibility rays walked in place (scenes in LDS).
// PHASE 1 / 2 (scenes outside LDS): 1 prepares the visibility rays of every vertex and light
// (ShadowDev), k_shadow_walk walks them with per-lane refill, 2 does the rest of the vertex
// with the answers.  The weights w are added in light order in both forms, so the sums agree
// bit for bit.  TEX: the scene has a diffuse map (untextured scenes run without that code).
template <int MODE, int BLOCK, bool COUNT, bool RES, int PHASE, bool TEX>
__global__ __launch_bounds__(BLOCK) void k_shade(SceneDev S, LdsPlan plan, uint32_t kRun, PathFrame F, int depth, uint32_t n,
                                                 QueueDev in, HitDev hit, QueueDev out, ShadowDev sd,
                                                 const unsigned long long *chunk_mask, const uint32_t *chunk_base,
                                                 const uint32_t *block_off, uint32_t *cnt, RadDev rad,
                                                 unsigned long long *stats) {
  uint32_t stk = 0;
  LdsScene C{};
  if constexpr (PHASE == 0) C = stage_scene<BLOCK, MODE>(S, plan, stk);
  const int lane = threadIdx.x & 63;
  uint32_t nb = 0, nt = 0, n_hit = 0;
  uint32_t *head = cnt + 3;
  const uint32_t n_chunks = (n + 63u) / 64u, waves = gridDim.x * (BLOCK / 64);
  // PHASE 0 walks visibility rays, so its chunks differ in cost and are claimed from the work queue.  PHASE 1
  // and 2 stream: every chunk costs about the same, and wave w simply takes chunks w, w + waves, ...  (claimed
  // from the one queue head they were bound by its atomic, 0.5 ms a launch whatever the number of paths).
  uint32_t len = PHASE == 0 ? kRun : 1u;
  uint32_t run;
  if constexpr (PHASE == 0) run = next_run(head, len);
  else run = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)));
  for (uint32_t c = 0;; ++c) {
    if (c == len) {
      if constexpr (PHASE == 0) {
        len = next_len(n_chunks, run, len, kRun, waves);
        run = next_run(head, len);
      } else {
        if (run >= n_chunks - (n_chunks < waves ? n_chunks : waves)) break;  // run + waves would pass the end (or wrap)
        run += waves;
      }
      c = 0;
    }
    const uint32_t first_item = (run + c) * 64u;
    if ((uint64_t)(run + c) * 64u >= n) break;
    const uint32_t i = first_item + (uint32_t)lane;
    const bool active = i < n;
    const unsigned long long m = PHASE == 1 ? 0ull : chunk_mask[run + c];
    const bool cont = (m >> lane) & 1ull;  // this path goes on (decided with the closest hit)
    bool alive = false, did_hit = false;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0), T = mk(0, 0, 0), L = mk(0, 0, 0);
    uint32_t id = 0;
    if (active) {
      const float4 hr = hit.H()[i];
      const float t = hr.x;
      if (t == __builtin_inff()) {  // camera.cpp:65-66
        id = in.id(i);
        if (PHASE != 1) {
          in.load_sums(i, T, L);
          L = add(L, mul(T, mk(F.bg[0], F.bg[1], F.bg[2])));
        }
      } else {
        did_hit = true;
        V3 x, nrm;
        {
          V3 ro, rd;
          in.load_ray(i, ro, rd, id);
          x = add(ro, scale(rd, t));                                          // camera.cpp:68
          nrm = normalize(mk(hr.y, hr.z, hr.w));                              // camera.cpp:69
          if (dot(nrm, normalize(rd)) > 0.0f) nrm = neg(nrm);                 // camera.cpp:72-73
        }

        PathRng rng;
        rng.init(F.seed, id % F.P, F.sample0 + (id / F.P) * F.stride);
        const uint32_t base = 2u + (uint32_t)depth * F.dims_per_vertex;

        if (PHASE == 1) {
          sd.x()[i] = x.x; sd.y()[i] = x.y; sd.z()[i] = x.z;
          for (uint32_t l = 0; l < S.n_lights; ++l) {
            V3 to, w;
            sample_light(S, rng, base, l, x, nrm, to, w);
            const size_t k = ShadowDev::at(i, l, S.n_lights);
            sd.tx()[k] = to.x; sd.ty()[k] = to.y; sd.tz()[k] = to.z;
            sd.wx()[k] = w.x; sd.wy()[k] = w.y; sd.wz()[k] = w.z;
          }
        } else {
          // direct light: every light at every vertex, camera.cpp:96-115.  Only the hit point, the normal and
          // the generator are alive across the visibility walks; the rest of the path's state is loaded after them.
          V3 local = mk(0, 0, 0);
          for (uint32_t l = 0; l < S.n_lights; ++l) {
            if (PHASE == 0) {
              V3 to, w;
              sample_light(S, rng, base, l, x, nrm, to, w);
              HitRec sh;
              // camera.cpp:101-102: the light is skipped only when this ray hits nothing
              if (!walk<MODE, true, COUNT, false, RES>(S, C, x, to, stk, BLOCK, sh, nb, nt)) continue;
              local = add(local, w);
            } else {
              const size_t k = ShadowDev::at(i, l, S.n_lights);
              if (!sd.vis()[ShadowDev::vis_at(i, l, S.n_lights)]) continue;
              local = add(local, mk(sd.wx()[k], sd.wy()[k], sd.wz()[k]));
            }
          }
          in.load_sums(i, T, L);
          const uint32_t mat_id = hit.M()[i];
          const float4 *mp = reinterpret_cast<const float4 *>(S.mats + mat_id);
          const float4 m0 = mp[0], m1 = mp[1];
          const V3 ke = mk(m1.x, m1.y, m1.z);

          // Russian roulette (camera.cpp:118-119) was drawn by k_trace_queue: `cont`
          if (cont) local = add(local, ke);  // camera.cpp:154
          L = add(L, mul(T, local));
          if (cont) {
            V3 w;
            uint32_t slot = 0;
            float bu = 0, bv = 0;
            if (TEX && __float_as_uint(m0.w) != 0) {
              const float4 ur = hit.U()[i];
              bu = ur.x; bv = ur.y; slot = __float_as_uint(ur.z);
            }
            bounce<MODE, TEX>(S, F, rng, base, nrm, m0, slot, bu, bv, T, w);
            o = x;  // no offset along the normal, camera.cpp:155
            d = w;
            alive = true;
          }
        }
      }
    }
    if (PHASE == 1) continue;
    n_hit += (uint32_t)__popcll(__ballot(did_hit));
    if (alive) {
      const uint32_t j = block_off[(run + c) / kScanSpan] + chunk_base[run + c] +
                         (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      out.store(j, o, d, T, L, id);
    } else if (active) {
      rad.R()[id] = make_float4(L.x, L.y, L.z, 0.0f);
    }
  }
  if (PHASE != 1 && lane == 0 && n_hit) atomicAdd(&cnt[1], n_hit);
  if (COUNT && PHASE == 0) {
    wave_add64(stats + 2, nb);
    wave_add64(stats + 3, nt);
  }
}

// Waves per SIMD the tail kernel is compiled for when the scene is outside LDS (256-thread workgroups).  It
// holds a whole path beside the walk's state, 176 registers uncapped (2 waves); its walks wait on memory, and
// more waves in flight are worth more than the spills: 7 M triangles at 4 samples per pixel 38.2 ms uncapped,
// 35.8 ms at 3 waves (168 registers, 36 bytes of scratch), 35.2 ms at 4 (128 registers, 224 bytes).
#ifndef PT_TAIL_OCC
#define PT_TAIL_OCC 4
#endif
// The tail of a batch: once few paths are left, every pass of the wavefront pipeline costs its fixed
// price (launches, a pass over mostly empty queues, the slowest ray of the pass) for almost no work.
// This kernel takes the remaining paths and follows each one to its end, one path per lane: closest
// hit, the light loop with its visibility walks, roulette, bounce, again.  The operations on a path and
// their order are those of k_trace_queue + k_shade, so the radiance is the same bit for bit.
// cnt[1] += vertices that hit something, cnt[6] += closest-hit rays, cnt[7] = max(depth levels done).
template <int MODE, int BLOCK, bool COUNT, bool RES, bool TEX>
__global__ __launch_bounds__(BLOCK, (!RES && BLOCK == 256) ? PT_TAIL_OCC : 1) void k_tail(SceneDev S, LdsPlan plan, uint32_t kRun, PathFrame F, int depth0, uint32_t n,
                                                QueueDev in, RadDev rad, uint32_t *cnt, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK, MODE>(S, plan, stk);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t nb = 0, nt = 0, sb = 0, stt = 0, n_hit = 0, n_rays = 0, levels = 0;
  uint32_t *head = cnt + 2;
  // paths of the claimed run not handed out yet (the same in all lanes); a lane whose path has ended
  // stores its radiance and takes the next one, so the wave stays full while the queue lasts
  uint64_t pool = 0, pool_end = 0;
  bool more = true;
  bool alive = false, mine = false;
  const uint64_t n_chunks = ((uint64_t)n + 63u) / 64u;
  const uint32_t waves = gridDim.x * (BLOCK / 64);
  uint32_t len = kRun;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 1), T = mk(0, 0, 0), L = mk(0, 0, 0);
  uint32_t id = 0;
  int depth = depth0;
  for (;;) {
    const unsigned long long m_alive = __ballot(alive);
    const uint32_t n_idle = 64u - (uint32_t)__popcll(m_alive);
    if ((n_idle >= 8u && (more || pool < pool_end)) || m_alive == 0) {
      if (!alive && mine) {
        rad.R()[id] = make_float4(L.x, L.y, L.z, 0.0f);
        const uint32_t done = (uint32_t)(depth - depth0);
        if (done > levels) levels = done;
        mine = false;
      }
      if (pool >= pool_end && more) {
        const uint32_t run = next_run(head, len);
        if ((uint64_t)run * 64u >= n) {
          more = false;
        } else {
          pool = (uint64_t)run * 64u;
          const uint64_t e = ((uint64_t)run + len) * 64u;
          pool_end = e < n ? e : n;
          len = next_len(n_chunks, run, len, kRun, waves);
        }
      }
      if (!alive) {
        const unsigned long long idle = ~m_alive;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const uint64_t i = pool + rank;
        if (i < pool_end) {
          in.load_ray(i, o, d, id);
          in.load_sums(i, T, L);
          depth = depth0;
          alive = true;
          mine = true;
        }
      }
      pool = pool_end - pool > n_idle ? pool + n_idle : pool_end;
      if (__ballot(alive) == 0) {
        if (!more && pool >= pool_end) break;
        continue;
      }
    }
    if (alive) {
      ++n_rays;
      HitRec h;
      const bool got = walk<MODE, false, COUNT, true, RES>(S, C, o, d, stk, BLOCK, h, nb, nt);
      if (!got) {  // camera.cpp:65-66
        L = add(L, mul(T, mk(F.bg[0], F.bg[1], F.bg[2])));
        alive = false;
      } else {
        ++n_hit;
        const V3 x = add(o, scale(d, h.t));                                  // camera.cpp:68
        V3 nrm = normalize(h.n);                                             // camera.cpp:69
        if (dot(nrm, normalize(d)) > 0.0f) nrm = neg(nrm);                   // camera.cpp:72-73
        PathRng rng;
        rng.init(F.seed, id % F.P, F.sample0 + (id / F.P) * F.stride);
        const uint32_t base = 2u + (uint32_t)depth * F.dims_per_vertex;
        const float4 *mp = reinterpret_cast<const float4 *>(S.mats + h.mat);
        const float4 m0 = mp[0], m1 = mp[1];
        V3 local = mk(0, 0, 0);
        for (uint32_t l = 0; l < S.n_lights; ++l) {  // camera.cpp:96-115
          V3 to, w;
          sample_light(S, rng, base, l, x, nrm, to, w);
          HitRec sh;
          if (!walk<MODE, true, COUNT, false, RES>(S, C, x, to, stk, BLOCK, sh, sb, stt)) continue;
          local = add(local, w);
        }
        const bool cont = survives_id(S, F, depth, id, h.t);                 // camera.cpp:118-119
        if (cont) local = add(local, mk(m1.x, m1.y, m1.z));                  // camera.cpp:154
        L = add(L, mul(T, local));
        if (cont) {
          V3 w;
          bounce<MODE, TEX>(S, F, rng, base, nrm, m0, h.slot, h.u, h.v, T, w);
          o = x;  // no offset along the normal, camera.cpp:155
          d = w;
        } else {
          alive = false;
        }
      }
      ++depth;
    }
  }
  {
    uint32_t a = n_hit, b = n_rays, c = levels;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
      const uint32_t c2 = __shfl_down(c, off);
      c = c2 > c ? c2 : c;
    }
    if (lane == 0) {
      if (a) atomicAdd(&cnt[1], a);
      if (b) atomicAdd(&cnt[6], b);
      if (c) atomicMax(&cnt[7], c);
    }
  }
  if (COUNT) {
    wave_add64(stats + 0, nb);
    wave_add64(stats + 1, nt);
    wave_add64(stats + 2, sb);
    wave_add64(stats + 3, stt);
  }
}

// Exclusive prefix sums of the survivor counts popcount(chunk_mask[c]), two levels:
// k_scan_local scans kScanSpan chunks per workgroup and leaves each workgroup's total in
// block_tot; k_scan_top scans those totals in place and writes the grand total to *total.

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot, uint32_t &block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < (int)(blockDim.x >> 6) ? wave_tot[lane] : 0u;
    uint32_t winc = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(winc, off);
      if (lane >= off) winc += t;
    }
    if (lane < (int)(blockDim.x >> 6)) wave_tot[lane] = winc - w;
    if (lane == 63) wave_tot[64] = winc;
  }
  __syncthreads();
  block_total = wave_tot[64];
  return wave_tot[wave] + inc - v;
}

// 8 bytes that are each 0 or 1 -> 8 bits, byte k to bit k (the products land on distinct bits).
__device__ __forceinline__ unsigned long long pack8(unsigned long long w) { return (w * 0x0102040810204080ull) >> 56; }

// Also turns the per-ray survivor bytes of k_trace_queue into the 64-bit mask of each chunk.
__global__ __launch_bounds__(kScanBlock) void k_scan_local(uint32_t n, uint32_t n_chunks, const uint8_t *alive,
                                                           unsigned long long *chunk_mask, uint32_t *chunk_base,
                                                           uint32_t *block_tot) {
  __shared__ uint32_t wave_tot[65];
  const uint32_t first = (blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
  uint32_t c[kScanPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    c[k] = 0u;
    if (first + k < n_chunks) {
      const uint4 *p = reinterpret_cast<const uint4 *>(alive + (size_t)(first + k) * 64u);
      unsigned long long m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 w = p[j];
        m |= (pack8((unsigned long long)w.x | ((unsigned long long)w.y << 32)) |
              (pack8((unsigned long long)w.z | ((unsigned long long)w.w << 32)) << 8)) << (16 * j);
      }
      const uint32_t valid = n - (first + k) * 64u;  // the last chunk may be short; its tail bytes were never written
      if (valid < 64u) m &= (1ull << valid) - 1ull;
      chunk_mask[first + k] = m;
      c[k] = (uint32_t)__popcll(m);
    }
    sum += c[k];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(sum, wave_tot, total);
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (first + k < n_chunks) chunk_base[first + k] = run;
    run += c[k];
  }
  if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_top(uint32_t n_blocks, uint32_t *block_tot, uint32_t *total) {
  __shared__ uint32_t wave_tot[65];
  const uint32_t v = threadIdx.x < n_blocks ? block_tot[threadIdx.x] : 0u;
  uint32_t all;
  const uint32_t ex = block_exclusive_scan(v, wave_tot, all);
  if (threadIdx.x < n_blocks) block_tot[threadIdx.x] = ex;
  if (threadIdx.x == 0) *total = all;
}

// accum[p] += sum over the batch's samples, in sample order (so the float sum does not
// depend on how paths were scheduled).
__global__ __launch_bounds__(kBlock) void k_resolve(uint32_t P, uint32_t S, RadDev rad, float *accum) {
  const float4 *R = rad.R();
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= P) return;
  float r = accum[3 * (size_t)p + 0], g = accum[3 * (size_t)p + 1], b = accum[3 * (size_t)p + 2];
  for (uint32_t s = 0; s < S; ++s) {
    const size_t k = (size_t)s * P + p;
    const float4 x = R[k];
    r += x.x; g += x.y; b += x.z;
  }
  accum[3 * (size_t)p + 0] = r; accum[3 * (size_t)p + 1] = g; accum[3 * (size_t)p + 2] = b;
}

// The reference's own accumulator (camera.cpp:188-190): pixels = (i * pixels + radiance * 255) / (i + 1) after
// pass i, every operation in float as vector_'s operators perform it (math.cpp:168-183), the batch's samples
// applied in pass order.  `first` is the pass index of the batch's first sample.
__global__ __launch_bounds__(kBlock) void k_resolve_mean(uint32_t P, uint32_t S, uint32_t first, RadDev rad, float *pixels) {
  const float4 *R = rad.R();
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= P) return;
  float r = pixels[3 * (size_t)p + 0], g = pixels[3 * (size_t)p + 1], b = pixels[3 * (size_t)p + 2];
  for (uint32_t s = 0; s < S; ++s) {
    const float4 x = R[(size_t)s * P + p];
    const float i = (float)(first + s), n1 = (float)(first + s + 1u);
    r = (i * r + x.x * 255.0f) / n1;
    g = (i * g + x.y * 255.0f) / n1;
    b = (i * b + x.z * 255.0f) / n1;
  }
  pixels[3 * (size_t)p + 0] = r; pixels[3 * (size_t)p + 1] = g; pixels[3 * (size_t)p + 2] = b;
}

// dst += src (merging per-device accumulators that happen to live on one device).
__global__ __launch_bounds__(kBlock) void k_add_into(size_t n, float *dst, const float *src) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// Camera::pixels from the sums: pixels = sum / spp * 255 (the running mean of camera.cpp:188-190, as a sum).
__global__ __launch_bounds__(kBlock) void k_resolve_pixels(size_t n, float spp, const float *sum, float *pixels) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) pixels[i] = sum[i] / spp * 255.0f;
}

// The display conversion of the GUI callback (main.cpp:43-47): norm(i) = std::clamp((int)i, 0, 255) with the
// float widened to double first (exact), i.e. truncation toward zero as the x86-64 build performs it, where a
// value out of int range or NaN converts to INT_MIN and so clamps to 0.  One RGB byte triple per pixel.
// FROM_SUMS: the input holds radiance sums of `spp` samples; otherwise it holds Camera::pixels itself.
template <bool FROM_SUMS>
__global__ __launch_bounds__(kBlock) void k_resolve_display(size_t n, float spp, const float *in, uint8_t *rgb) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int v = cvt_int_x86(FROM_SUMS ? in[i] / spp * 255.0f : in[i]);
  rgb[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

#endif

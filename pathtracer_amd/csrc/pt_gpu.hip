// pt_gpu.hip — gfx950 kernels of the render hot path and the C ABI in front of them
// (include/ptgpu.h).  Wavefront organisation: every stage is its own kernel over a
// structure-of-arrays queue of live paths; one lane carries one ray.
//
//   k_raygen        Camera::point_towards (camera.cpp:161-169) [+ thin lens]  -> path queue
//   k_trace_queue   Scene::castray (scene.cpp:18-20) for every queued ray     -> hit buffer,
//                   plus the roulette draw: one byte per ray, "this path goes on"
//   k_scan_*        those bytes -> 64-bit chunk masks -> every survivor's place in the next queue
//   k_shade<..,0>   the body of Camera::pathtrace (camera.cpp:63-159) for one path vertex:
//                   normal, the light loop with its visibility rays, the next direction,
//                   throughput update, survivors written straight to their places
//   k_shade<..,1>, k_shadow_walk, k_shade<..,2>
//                   the same vertex in three kernels for scenes that do not fit LDS: visibility
//                   rays prepared, walked with per-lane refill, vertex finished
//   k_resolve       adds the finished samples of a batch to the caller's accumulator
//   k_trace_api / k_any_api   Seam A for callers that bring their own rays
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "pt_device.h"
#include "pt_flatten.h"

using namespace ptd;

// ===================================================================== device side

static const int kBlock = 256;  // block size of the streaming kernels (ray generation, resolve)
static const int kScanBlock = 1024;
static const int kScanPer = 4;
static const int kScanSpan = kScanBlock * kScanPer;  // chunks one workgroup of k_scan_local covers

struct QueueDev {  // one live path per index, arrays `cap` long
  float *ox, *oy, *oz, *dx, *dy, *dz;  // ray
  float *tx, *ty, *tz;                 // throughput
  float *lx, *ly, *lz;                 // radiance gathered so far
  uint32_t *id;                        // sample_in_batch * P + pixel
};

struct HitDev {  // closest hit of every queued ray
  float *t;
  float *nx, *ny, *nz;  // Triangle::normal() of the hit triangle, un-normalised
  uint32_t *mat;
  float *u, *v;         // barycentrics and triangle record of the hit: written and read only when
  uint32_t *slot;       // the scene has a textured material
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
  float rr_prob;
  int rr_min_depth, max_depth, sampling;
  float bg[3];
  float pdf_f;           // (float)(1/(2*pi) * rr_prob), camera.cpp:152
  float pi_f;            // (float)M_PI, camera.cpp:122
  uint32_t dims_per_vertex;
};

// What a traversal kernel keeps in LDS (see LdsScene) and how many words of the
// per-launch counter block it uses as its work-queue head.
struct LdsPlan {
  uint32_t n_nodes;    // top nodes staged in LDS
  uint32_t n_tris;     // 0, or all triangles
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

__device__ __forceinline__ void camera_ray(const FrameDev &F, int px, int py, PathRng &rng, V3 &o, V3 &d) {
  // camera.cpp:162-167: float arithmetic on x and y, z prepared in double by the host
  const float x = ((float)px - (float)F.W / 2) * F.vw / (float)F.W;
  const float y = ((float)py - (float)F.H / 2) * F.neg_vh / (float)F.H;
  const V3 dc = mk(x, y, F.pin_z);
  if (F.aperture > 0.0f) {  // thin lens: not in the reference
    const float u1 = rng.real(0.0f, 1.0f, 0), u2 = rng.real(0.0f, 1.0f, 1);
    const float r = F.aperture * sqrtf(u1);
    const double ph = 2 * M_PI * (double)u2;
    const V3 lens = mk((float)(cos(ph) * (double)r), (float)(sin(ph) * (double)r), 0.0f);
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
  const uint32_t s = i / F.P, p = i - s * F.P;
  const int py = (int)(p / (uint32_t)F.W), px = (int)(p - (uint32_t)py * (uint32_t)F.W);
  PathRng rng;
  rng.init(F.seed, p, F.sample0 + s * F.stride);
  V3 o, d;
  camera_ray(F, px, py, rng, o, d);
  q.ox[i] = o.x; q.oy[i] = o.y; q.oz[i] = o.z;
  q.dx[i] = d.x; q.dy[i] = d.y; q.dz[i] = d.z;
  q.tx[i] = 1.0f; q.ty[i] = 1.0f; q.tz[i] = 1.0f;
  q.lx[i] = 0.0f; q.ly[i] = 0.0f; q.lz[i] = 0.0f;
  q.id[i] = i;
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
template <int BLOCK>
__device__ __forceinline__ LdsScene stage_scene(const SceneDev &S, LdsPlan plan, uint32_t &stack_base) {
  const float4 *gn = reinterpret_cast<const float4 *>(S.nodes);
  const float4 *gt = reinterpret_cast<const float4 *>(S.tris);
  LdsScene C;
  C.node_base = 0;
  C.tri_base = 4u * plan.n_nodes;
  C.n_nodes = plan.n_nodes;
  C.n_tris = plan.n_tris;
  C.stack_lds = plan.stack_lds;
  C.spill_stride = gridDim.x * BLOCK;
  C.spill = plan.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (uint32_t i = threadIdx.x; i < 4u * plan.n_nodes; i += BLOCK) pt_smem[(i & 3u) * plan.n_nodes + (i >> 2)] = gn[i];
  for (uint32_t i = threadIdx.x; i < 3u * plan.n_tris; i += BLOCK)
    pt_smem[C.tri_base + (i % 3u) * plan.n_tris + (i / 3u)] = gt[i];
  __syncthreads();
  stack_base = 4u * (C.tri_base + 3u * plan.n_tris) + threadIdx.x;  // in 32-bit words
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

// Walks one ray per work item for items [0, n), persistent over the whole launch.
//   fetch(i, o, d)  loads item i's ray; returns false when the item needs no walk (it then has
//                   written whatever the item's output is);
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
    for (;;) {
      const uint32_t run = next_run(head, kRun);
      if ((uint64_t)run * 64u >= n) break;
      for (uint32_t c = 0; c < kRun; ++c) {
        const uint64_t base = ((uint64_t)run + c) * 64u;
        if (base >= n) break;
        const uint64_t i = base + lane;
        V3 o, d;
        if (i < n && fetch(i, o, d)) {
          RayState st;
          walk_init<MODE, WANT_N>(S, st, o, d);
          do {
            walk_round<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, BLOCK, nb, nt, S.quorum);
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
    RayState st;
    st.live = false;
    st.pend = 0;
    for (;;) {
      const unsigned long long m_busy = __ballot(busy);
      const uint32_t n_idle = 64u - (uint32_t)__popcll(m_busy);
      const bool can_fetch = more || pool < pool_end;
      uint32_t n_idle_now = n_idle;
      if ((n_idle >= S.refill && can_fetch) || m_busy == 0) {
        if (!busy && mine != NONE) {
          retire(mine, st);
          mine = NONE;
        }
        if (pool >= pool_end && more) {
          const uint32_t run = next_run(head, kRun);
          if ((uint64_t)run * 64u >= n) {
            more = false;
          } else {
            pool = (uint64_t)run * 64u;
            const uint64_t e = ((uint64_t)run + kRun) * 64u;
            pool_end = e < n ? e : n;
          }
        }
        if (!busy) {
          const unsigned long long idle = ~m_busy;
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
          const uint64_t i = pool + rank;
          V3 o, d;
          if (i < pool_end && fetch(i, o, d)) {
            walk_init<MODE, WANT_N>(S, st, o, d);
            mine = i;
            busy = true;
          }
        }
        pool = pool_end - pool > n_idle ? pool + n_idle : pool_end;
        const unsigned long long m_now = __ballot(busy);
        if (m_now == 0) {
          if (!more && pool >= pool_end) break;
          continue;  // nothing to walk among these items, or the run is used up: fetch again
        }
        n_idle_now = 64u - (uint32_t)__popcll(m_now);
      }
      // the quorum for testing parked leaves counts against the lanes that have a ray at all
      walk_round<MODE, ANY, COUNT, WANT_N, RES>(S, C, st, stk, BLOCK, nb, nt, ((64u - n_idle_now) * S.quorum) >> 6);
      if (busy && st.done()) busy = false;
    }
  }
}

// Does the path of queue entry i go on after a vertex whose closest hit is at t?  It does iff it
// hit something and passes Russian roulette (camera.cpp:65, 118-119).
__device__ __forceinline__ bool survives(const SceneDev &S, const FrameDev &F, int depth, const QueueDev &q, uint32_t i,
                                         float t) {
  if (t == __builtin_inff()) return false;
  if (F.max_depth > 0 && depth >= F.max_depth) return false;
  if (depth <= F.rr_min_depth) return true;  // the roulette draw exists only when depth > rr_min_depth
  const uint32_t id = q.id[i];
  PathRng rng;
  rng.init(F.seed, id % F.P, F.sample0 + (id / F.P) * F.stride);
  const uint32_t dim = 2u + (uint32_t)depth * F.dims_per_vertex + 3u * S.n_lights;
  return (double)rng.real(0.0f, 1.0f, dim) < (double)F.rr_prob;
}

// Closest hit for every queued ray.  The kernel also decides which paths go on after this
// vertex and leaves that as one byte per ray in `alive_out`, from which k_scan_local forms the
// 64-bit mask of each chunk, so that the survivors' places in the next queue are known before
// the shading kernel writes them.
template <int MODE, int BLOCK, bool COUNT, bool RES, bool REFILL>
__global__ __launch_bounds__(BLOCK) void k_trace_queue(SceneDev S, LdsPlan plan, uint32_t kRun, FrameDev F, int depth,
                                                       uint32_t n, QueueDev q, HitDev hit, uint8_t *alive_out,
                                                       uint32_t *head, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, COUNT, true, RES, REFILL, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(q.ox[i], q.oy[i], q.oz[i]);
        d = mk(q.dx[i], q.dy[i], q.dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        const HitRec &h = st.best;
        hit.t[i] = h.t;
        hit.nx[i] = h.n.x; hit.ny[i] = h.n.y; hit.nz[i] = h.n.z;
        hit.mat[i] = h.mat;
        if (S.n_tex) {
          hit.u[i] = h.u; hit.v[i] = h.v;
          hit.slot[i] = h.slot;
        }
        alive_out[i] = survives(S, F, depth, q, (uint32_t)i, h.t) ? 1 : 0;
      });
  if (COUNT) {
    wave_add64(stats + 0, nb);
    wave_add64(stats + 1, nt);
  }
}

template <int MODE, int BLOCK, bool COUNT, bool RES>
__global__ __launch_bounds__(BLOCK) void k_trace_api(SceneDev S, LdsPlan plan, uint32_t kRun, size_t n, const float *ox,
                                                     const float *oy, const float *oz, const float *dx,
                                                     const float *dy, const float *dz, float *t, int32_t *tri,
                                                     float *u, float *v, uint32_t *head, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, COUNT, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
        d = mk(dx[i], dy[i], dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        const HitRec &h = st.best;
        t[i] = h.t;
        tri[i] = h.slot != 0xffffffffu ? (int32_t)S.tris[h.slot].tri_id : -1;
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
                                                   const float *dz, uint8_t *hit, uint32_t *head) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, true, false, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
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
                                                    uint32_t *head) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, false, false, false, RES, !RES, BLOCK>(
      S, C, stk, n, kRun, head, nb, nt,
      [&](uint64_t i, V3 &o, V3 &d) {
        o = mk(ox[i], oy[i], oz[i]);
        d = mk(dx[i], dy[i], dz[i]);
        return true;
      },
      [&](uint64_t i, const RayState &st) {
        hits[i] = st.best.slot != 0xffffffffu ? (int32_t)levels[st.best.slot] : miss_chain_hits(S, st.o, st.d);
      });
}

// Visibility rays of the split shading pipeline (scenes outside LDS): item k = l*n + i is the
// ray from vertex i towards its sample of light l, prepared by k_shade<..., 1>.
struct ShadowDev {
  float *x, *y, *z;        // per vertex: hit point
  float *tx, *ty, *tz;     // per item: direction towards the light sample
  float *wx, *wy, *wz;     // per item: what the light adds if the ray hits anything (camera.cpp:101-114)
  uint8_t *vis;            // per item: the ray hit something
};

template <int MODE, int BLOCK, bool COUNT, bool RES>
__global__ __launch_bounds__(BLOCK) void k_shadow_walk(SceneDev S, LdsPlan plan, uint32_t kRun, uint32_t n, HitDev hit,
                                                       ShadowDev sd, uint32_t *head, unsigned long long *stats) {
  uint32_t stk;
  const LdsScene C = stage_scene<BLOCK>(S, plan, stk);
  uint32_t nb = 0, nt = 0;
  walk_items<MODE, true, COUNT, false, RES, !RES, BLOCK>(
      S, C, stk, (uint64_t)n * S.n_lights, kRun, head, nb, nt,
      [&](uint64_t k, V3 &o, V3 &d) {
        const uint32_t i = (uint32_t)(k % n);
        if (hit.t[i] == __builtin_inff()) return false;  // no vertex, no light loop: nobody reads vis[k]
        o = mk(sd.x[i], sd.y[i], sd.z[i]);
        d = mk(sd.tx[k], sd.ty[k], sd.tz[k]);
        return true;
      },
      [&](uint64_t k, const RayState &st) { sd.vis[k] = st.found ? 1 : 0; });
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
  V3 v = mk((float)(cos(ang) * (double)st), (float)(sin(ang) * (double)st), phi);
  v = normalize(v);
  return dot(v, n) < 0.0f ? neg(v) : v;
}
// cosine-weighted direction about n: not in the reference
__device__ __forceinline__ V3 cosine_dir(V3 n, float x, float y) {
  const float r = sqrtf(x);
  const float lz = sqrtf(fmaxf(0.0f, 1.0f - x));
  const double ang = 2 * M_PI * (double)y;
  const float lx = (float)(cos(ang) * (double)r), ly = (float)(sin(ang) * (double)r);
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
__device__ __forceinline__ V3 texel_brdf(const SceneDev &S, uint32_t tex, uint32_t slot, float bu, float bv) {
  const float *uv = S.tri_uv + (size_t)slot * 6;
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

// One path vertex per work item.  Which paths survive was decided by k_trace_queue and
// prefix-summed by k_scan_*, so a survivor is written straight to its place in `out`: chunk
// offset + its rank among the chunk's survivors (wave ballot mask + popcount).
// The compaction is stable: the queue stays in pixel order and no two launches differ.
// Claiming an output slot per wave with an atomic on one counter instead was measured at
// 260 ms per frame on its own: a single word takes under 100 atomics per microsecond.
// cnt[1] = paths that hit something, cnt[3] / cnt[4] = this kernel's work-queue head (phase 1: cnt[4]).
//
// PHASE 0: the whole vertex in one kernel, visibility rays walked in place (scenes in LDS).
// PHASE 1 / 2 (scenes outside LDS): 1 prepares the visibility rays of every vertex and light
// (ShadowDev), k_shadow_walk walks them with per-lane refill, 2 does the rest of the vertex
// with the answers.  The weights w are added in light order in both forms, so the sums agree
// bit for bit.  TEX: the scene has a diffuse map (untextured scenes run without that code).
template <int MODE, int BLOCK, bool COUNT, bool RES, int PHASE, bool TEX>
__global__ __launch_bounds__(BLOCK) void k_shade(SceneDev S, LdsPlan plan, uint32_t kRun, FrameDev F, int depth, uint32_t n,
                                                 QueueDev in, HitDev hit, QueueDev out, ShadowDev sd,
                                                 const unsigned long long *chunk_mask, const uint32_t *chunk_base,
                                                 const uint32_t *block_off, uint32_t *cnt, float *Lr, float *Lg,
                                                 float *Lb, unsigned long long *stats) {
  uint32_t stk = 0;
  LdsScene C{};
  if constexpr (PHASE == 0) C = stage_scene<BLOCK>(S, plan, stk);
  const int lane = threadIdx.x & 63;
  uint32_t nb = 0, nt = 0, n_hit = 0;
  uint32_t *head = cnt + (PHASE == 1 ? 4 : 3);
  for (uint32_t run = next_run(head, kRun), c = 0;; ++c) {
    if (c == kRun) {
      run = next_run(head, kRun);
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
      id = in.id[i];
      const float t = hit.t[i];
      if (PHASE != 1) {
        T = mk(in.tx[i], in.ty[i], in.tz[i]);
        L = mk(in.lx[i], in.ly[i], in.lz[i]);
      }
      if (t == __builtin_inff()) {  // camera.cpp:65-66
        L = add(L, mul(T, mk(F.bg[0], F.bg[1], F.bg[2])));
      } else {
        did_hit = true;
        o = mk(in.ox[i], in.oy[i], in.oz[i]);
        d = mk(in.dx[i], in.dy[i], in.dz[i]);
        const V3 x = add(o, scale(d, t));                               // camera.cpp:68
        V3 nrm = normalize(mk(hit.nx[i], hit.ny[i], hit.nz[i]));        // camera.cpp:69
        if (dot(nrm, normalize(d)) > 0.0f) nrm = neg(nrm);              // camera.cpp:72-73

        PathRng rng;
        rng.init(F.seed, id % F.P, F.sample0 + (id / F.P) * F.stride);
        const uint32_t base = 2u + (uint32_t)depth * F.dims_per_vertex;

        if (PHASE == 1) {
          sd.x[i] = x.x; sd.y[i] = x.y; sd.z[i] = x.z;
          for (uint32_t l = 0; l < S.n_lights; ++l) {
            V3 to, w;
            sample_light(S, rng, base, l, x, nrm, to, w);
            const size_t k = (size_t)l * n + i;
            sd.tx[k] = to.x; sd.ty[k] = to.y; sd.tz[k] = to.z;
            sd.wx[k] = w.x; sd.wy[k] = w.y; sd.wz[k] = w.z;
          }
        } else {
          const uint32_t mat_id = hit.mat[i];
          const float4 *mp = reinterpret_cast<const float4 *>(S.mats + mat_id);
          const float4 m0 = mp[0], m1 = mp[1];
          const V3 kd = mk(m0.x, m0.y, m0.z), ke = mk(m1.x, m1.y, m1.z);
          // direct light: every light at every vertex, camera.cpp:96-115
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
              const size_t k = (size_t)l * n + i;
              if (!sd.vis[k]) continue;
              local = add(local, mk(sd.wx[k], sd.wy[k], sd.wz[k]));
            }
          }

          // Russian roulette (camera.cpp:118-119) was drawn by k_trace_queue: `cont`
          if (cont) local = add(local, ke);  // camera.cpp:154
          L = add(L, mul(T, local));
          if (cont) {
            const float rx = rng.real(0.0f, 1.0f, base + 3u * S.n_lights + 1u);
            const float ry = rng.real(0.0f, 1.0f, base + 3u * S.n_lights + 2u);
            V3 w;
            const uint32_t tex = __float_as_uint(m0.w);
            const bool textured = TEX && tex != 0;                                // camera.cpp:121
            V3 texel = mk(0, 0, 0);
            if (textured) texel = texel_brdf(S, tex, hit.slot[i], hit.u[i], hit.v[i]);
            if (F.sampling == PT_SAMPLE_COSINE) {
              w = cosine_dir(nrm, rx, ry);
              T = divs(mul(T, textured ? scale(texel, F.pi_f) : kd), F.rr_prob);
            } else {
              w = uniform_dir(nrm, rx, ry);
              const V3 brdf = textured ? texel : divs(kd, F.pi_f);                // camera.cpp:122-146
              T = divs(scale(mul(T, brdf), fabsf(dot(nrm, w))), F.pdf_f);         // camera.cpp:155-156
            }
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
      out.ox[j] = o.x; out.oy[j] = o.y; out.oz[j] = o.z;
      out.dx[j] = d.x; out.dy[j] = d.y; out.dz[j] = d.z;
      out.tx[j] = T.x; out.ty[j] = T.y; out.tz[j] = T.z;
      out.lx[j] = L.x; out.ly[j] = L.y; out.lz[j] = L.z;
      out.id[j] = id;
    } else if (active) {
      Lr[id] = L.x; Lg[id] = L.y; Lb[id] = L.z;
    }
  }
  if (PHASE != 1 && lane == 0 && n_hit) atomicAdd(&cnt[1], n_hit);
  if (COUNT && PHASE == 0) {
    wave_add64(stats + 2, nb);
    wave_add64(stats + 3, nt);
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
__global__ __launch_bounds__(kBlock) void k_resolve(uint32_t P, uint32_t S, const float *Lr, const float *Lg,
                                                    const float *Lb, float *accum) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= P) return;
  float r = accum[3 * (size_t)p + 0], g = accum[3 * (size_t)p + 1], b = accum[3 * (size_t)p + 2];
  for (uint32_t s = 0; s < S; ++s) {
    const size_t k = (size_t)s * P + p;
    r += Lr[k]; g += Lg[k]; b += Lb[k];
  }
  accum[3 * (size_t)p + 0] = r; accum[3 * (size_t)p + 1] = g; accum[3 * (size_t)p + 2] = b;
}

// ===================================================================== host side

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(e_ == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_DEVICE,                   \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                         \
  } while (0)

struct Workspace {
  size_t cap = 0;
  float *block = nullptr;  // 2 queues x 13 arrays + 8 hit arrays + 3 radiance arrays
  QueueDev q[2];
  HitDev hit{};
  float *Lr = nullptr, *Lg = nullptr, *Lb = nullptr;
  unsigned long long *chunk_mask = nullptr;  // one per 64 queue entries
  uint32_t *chunk_base = nullptr;
  uint32_t *block_tot = nullptr;             // one per kScanSpan chunks, at most kScanBlock
  uint8_t *alive = nullptr;                  // one per queue entry: the path goes on after this vertex
  float *shadow_block = nullptr;             // split shading only: ShadowDev arrays
  ShadowDev sd{};
  size_t shadow_items = 0;                   // capacity in (vertex, light) pairs
};

// Launch geometry of the traversal kernels for one scene.
struct Plan {
  int stack = 0;       // LDS stack entries per lane
  int block = 0;       // threads per workgroup
  LdsPlan lds{};       // staged part of the scene
  size_t lds_bytes = 0;
  unsigned max_blocks = 0;  // workgroups the chip keeps resident
  bool resident = false;    // all nodes, all triangles and the whole stack are in LDS
  bool split = false;       // shading runs as three kernels (visibility rays walked on their own)
  bool refill_resident = false;  // closest-hit kernel hands out rays lane by lane even when the scene is in LDS
};

struct PtScene {
  int device = 0;
  SceneDev dev{};
  ptl::WideNode *d_nodes = nullptr;
  ptl::TriRec *d_tris = nullptr;
  ptl::MatRec *d_mats = nullptr;
  ptl::LightRec *d_lights = nullptr;
  ptl::LightTri *d_ltris = nullptr;
  uint8_t *d_levels = nullptr;             // per triangle record: HitData::hits of a hit on it
  float *d_tri_uv = nullptr;               // textured scenes only
  ptl::TexRec *d_textures = nullptr;
  uint8_t *d_texels = nullptr;
  uint32_t max_depth = 0, n_meshes = 0;
  uint64_t bytes = 0;
  Plan plan;
  Workspace ws;
  uint32_t *d_spill = nullptr;             // traversal-stack spill area (deep trees only)
  uint32_t *d_cnt = nullptr;               // per launch: survivors, hits, work-queue heads of trace, shade, shade phase 1, visibility walk
  unsigned long long *d_stats = nullptr;   // node boxes, triangle tests: closest [0,1], visibility [2,3]
  uint32_t *h_cnt = nullptr;               // pinned
  unsigned long long *h_stats = nullptr;   // pinned
  bool count_stats = false;
  bool profile = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  PtProfile prof{};
};

// LDS stack entries per lane for a tree of the given depth (a few fixed sizes, so that the
// launch geometry does not change with every scene).
static int stack_class(uint32_t depth) {
  static const int sizes[] = {8, 12, 16, 24, 32, 48, 64, 96, 128};
  for (int sz : sizes)
    if (depth <= (uint32_t)sz) return sz;
  return 0;
}

// 160 KiB of LDS per CU: the per-lane stacks take STACK*4 bytes per thread, the rest
// holds the top of the tree and, when they fit as well, all triangles.
static int make_plan(PtScene *s) {
  Plan &p = s->plan;
  p.stack = stack_class(s->max_depth);
  if (!p.stack) return fail(PT_ERR_UNSUPPORTED, "tree deeper than 128 levels");
  // deeper trees keep only the first entries in LDS and spill the rest to global memory, so
  // that LDS still holds 16 waves per CU and a useful part of the tree
  int cap = 24;
  if (const char *e = std::getenv("PT_STACK_CAP")) {  // tuning knob
    const int c = std::atoi(e);
    if (c >= 4 && c <= 128) cap = c;
  }
  if (p.stack > cap) p.stack = cap;
  p.block = p.stack <= 32 ? 1024 : (p.stack <= 64 ? 512 : 256);
  if (const char *e = std::getenv("PT_BLOCK")) {  // tuning knob
    const int b = std::atoi(e);
    if ((b == 256 || b == 512 || b == 1024) && (size_t)p.stack * b * 4 <= 128 * 1024) p.block = b;
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  const size_t lds_cu = 160 * 1024;
  if (const char *e = std::getenv("PT_STACK")) {  // tuning knob: never below what the tree needs
    const int st = std::atoi(e);
    if (st >= p.stack && st <= 128) p.stack = st;
  }
  const size_t stack_bytes = (size_t)p.stack * p.block * 4;
  size_t budget = lds_cu - stack_bytes - 512;
  if (const char *e = std::getenv("PT_LDS_BUDGET_KB")) {
    const size_t kb = (size_t)std::atol(e) * 1024;
    if (kb < budget) budget = kb;
  }
  const size_t node_bytes = (size_t)s->dev.n_nodes * sizeof(ptl::WideNode);
  const size_t tri_bytes = (size_t)s->dev.n_tris * sizeof(ptl::TriRec);
  if (node_bytes + tri_bytes <= budget) {
    p.lds.n_nodes = s->dev.n_nodes;
    p.lds.n_tris = s->dev.n_tris;
  } else {
    const size_t fit = budget / sizeof(ptl::WideNode);
    p.lds.n_nodes = (uint32_t)(fit < s->dev.n_nodes ? fit : s->dev.n_nodes);
    p.lds.n_tris = 0;
  }
  p.lds_bytes = (size_t)p.lds.n_nodes * sizeof(ptl::WideNode) + (size_t)p.lds.n_tris * sizeof(ptl::TriRec) + stack_bytes;
  unsigned per_cu = (unsigned)(lds_cu / p.lds_bytes);
  const unsigned by_threads = 2048u / (unsigned)p.block;
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu < 1) per_cu = 1;
  p.max_blocks = per_cu * (unsigned)prop.multiProcessorCount;
  p.lds.stack_lds = (uint32_t)p.stack;
  p.lds.spill = nullptr;
  p.resident = p.lds.n_nodes == s->dev.n_nodes && p.lds.n_tris == s->dev.n_tris && s->max_depth <= (uint32_t)p.stack &&
               !std::getenv("PT_NO_RESIDENT");
  p.split = !p.resident;
  if (const char *e = std::getenv("PT_RES_REFILL")) p.refill_resident = std::atoi(e) != 0;  // tuning knob
  if (const char *e = std::getenv("PT_SPLIT")) p.split = std::atoi(e) != 0;  // tuning knob
  if (std::getenv("PT_VERBOSE"))
    std::fprintf(stderr, "ptgpu plan: depth %u, block %d, LDS stack %d entries/lane, %u nodes + %u triangles in LDS, %zu B LDS per workgroup, %u resident workgroups\n",
                 s->max_depth, p.block, p.stack, p.lds.n_nodes, p.lds.n_tris, p.lds_bytes, p.max_blocks);
  if (s->max_depth > (uint32_t)p.stack) {
    const size_t words = (size_t)p.max_blocks * p.block * (s->max_depth - (uint32_t)p.stack);
    HIP_TRY(hipMalloc(&s->d_spill, words * sizeof(uint32_t)));
    s->bytes += words * sizeof(uint32_t);
    p.lds.spill = s->d_spill;
  }
  return PT_OK;
}

static void free_workspace(Workspace &w) {
  if (w.block) (void)hipFree(w.block);
  if (w.shadow_block) (void)hipFree(w.shadow_block);
  w = Workspace();
}

static int ensure_workspace(PtScene *s, size_t cap) {
  Workspace &w = s->ws;
  if (w.cap >= cap) return PT_OK;
  free_workspace(w);
  const size_t arrays = 2 * 13 + 8 + 3;
  const size_t stride = (cap + 63) & ~(size_t)63;
  const size_t n_chunks = stride / 64;
  if ((n_chunks + kScanSpan - 1) / kScanSpan > (size_t)kScanBlock) return fail(PT_ERR_UNSUPPORTED, "too many paths in flight");
  const size_t extra = n_chunks * 12 + (size_t)kScanBlock * 4 + 256 + stride;
  HIP_TRY(hipMalloc(&w.block, arrays * stride * sizeof(float) + extra));
  float *p = w.block;
  auto take = [&]() { float *r = p; p += stride; return r; };
  for (int k = 0; k < 2; ++k) {
    QueueDev &q = w.q[k];
    q.ox = take(); q.oy = take(); q.oz = take(); q.dx = take(); q.dy = take(); q.dz = take();
    q.tx = take(); q.ty = take(); q.tz = take(); q.lx = take(); q.ly = take(); q.lz = take();
    q.id = reinterpret_cast<uint32_t *>(take());
  }
  w.hit.t = take(); w.hit.nx = take(); w.hit.ny = take(); w.hit.nz = take();
  w.hit.mat = reinterpret_cast<uint32_t *>(take());
  w.hit.u = take(); w.hit.v = take();
  w.hit.slot = reinterpret_cast<uint32_t *>(take());
  w.Lr = take(); w.Lg = take(); w.Lb = take();
  w.chunk_mask = reinterpret_cast<unsigned long long *>(p);
  w.chunk_base = reinterpret_cast<uint32_t *>(w.chunk_mask + n_chunks);
  w.block_tot = w.chunk_base + n_chunks;
  w.alive = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(w.block_tot + kScanBlock) + 63) & ~(uintptr_t)63);
  if (s->plan.split) {  // split shading: 3 floats per vertex, 6 floats + 1 byte per (vertex, light)
    const size_t items = stride * s->dev.n_lights;
    if (hipMalloc(&w.shadow_block, (3 * stride + 6 * items) * sizeof(float) + items) != hipSuccess) {
      w.shadow_block = nullptr;
      free_workspace(w);
      return fail(PT_ERR_NOMEM, "out of device memory for the visibility-ray workspace");
    }
    float *sp = w.shadow_block;
    w.sd.x = sp; sp += stride; w.sd.y = sp; sp += stride; w.sd.z = sp; sp += stride;
    w.sd.tx = sp; sp += items; w.sd.ty = sp; sp += items; w.sd.tz = sp; sp += items;
    w.sd.wx = sp; sp += items; w.sd.wy = sp; sp += items; w.sd.wz = sp; sp += items;
    w.sd.vis = reinterpret_cast<uint8_t *>(sp);
    w.shadow_items = items;
  }
  w.cap = cap;
  return PT_OK;
}

// Dynamic LDS above 64 KiB has to be allowed per kernel once.
template <typename K>
static int allow_lds(K kernel, size_t bytes) {
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PT_OK;
}

// Chunks of 64 items a wave claims at once: about a quarter of its share of the launch,
// between 1 and 16.
static uint32_t run_length(const Plan &p, size_t n) {
  const size_t chunks = (n + 63) / 64;
  const size_t waves = (size_t)p.max_blocks * (size_t)(p.block / 64);
  size_t r = chunks / (waves * 4);
  if (r < 1) r = 1;
  if (r > 16) r = 16;
  return (uint32_t)r;
}

static dim3 grid_for(const Plan &p, size_t n) {
  size_t g = (n + p.block - 1) / p.block;
  if (g > p.max_blocks) g = p.max_blocks;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

// Calls f(BLOCK) with the plan's workgroup size as an integral constant.
template <typename F>
static int dispatch_plan(const Plan &p, F &&f) {
  switch (p.block) {
    case 256: return f(std::integral_constant<int, 256>{});
    case 512: return f(std::integral_constant<int, 512>{});
    case 1024: return f(std::integral_constant<int, 1024>{});
    default: return fail(PT_ERR_UNSUPPORTED, "unsupported workgroup size");
  }
}

template <int MODE, bool COUNT>
static int launch_trace_queue(PtScene *s, const FrameDev &F, int depth, uint32_t n, const QueueDev &q, hipStream_t st) {
  return dispatch_plan(s->plan, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = !s->plan.resident ? k_trace_queue<MODE, BLOCK, COUNT, false, true>
             : s->plan.refill_resident ? k_trace_queue<MODE, BLOCK, COUNT, true, true> : k_trace_queue<MODE, BLOCK, COUNT, true, false>;
    int rc = allow_lds(k, s->plan.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(s->plan, n), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds, run_length(s->plan, n), F, depth, n, q,
                       s->ws.hit, s->ws.alive, s->d_cnt + 2, s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE, bool COUNT>
static int launch_shade(PtScene *s, const FrameDev &F, int depth, uint32_t n, const QueueDev &in,
                        const QueueDev &out, hipStream_t st) {
  return dispatch_plan(s->plan, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    const Workspace &w = s->ws;
    const uint32_t kRun = run_length(s->plan, n);
    if (!s->plan.split) {
      const bool tex = s->dev.n_tex != 0;
      auto k = s->plan.resident ? (tex ? k_shade<MODE, BLOCK, COUNT, true, 0, true> : k_shade<MODE, BLOCK, COUNT, true, 0, false>)
                                : (tex ? k_shade<MODE, BLOCK, COUNT, false, 0, true> : k_shade<MODE, BLOCK, COUNT, false, 0, false>);
      int rc = allow_lds(k, s->plan.lds_bytes);
      if (rc != PT_OK) return rc;
      hipLaunchKernelGGL(k, grid_for(s->plan, n), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds, kRun, F, depth, n, in,
                         w.hit, out, w.sd, w.chunk_mask, w.chunk_base, w.block_tot, s->d_cnt, w.Lr, w.Lg, w.Lb, s->d_stats);
      HIP_TRY(hipGetLastError());
      return PT_OK;
    }
    // prepare the visibility rays, walk them (with per-lane refill when the scene is outside LDS), finish the vertices
    if ((size_t)n * s->dev.n_lights > w.shadow_items) return fail(PT_ERR_DEVICE, "visibility-ray workspace too small");
    auto ka = k_shade<MODE, BLOCK, COUNT, false, 1, false>;
    hipLaunchKernelGGL(ka, grid_for(s->plan, n), dim3(BLOCK), 0, st, s->dev, s->plan.lds, kRun, F, depth, n, in, w.hit, out, w.sd,
                       w.chunk_mask, w.chunk_base, w.block_tot, s->d_cnt, w.Lr, w.Lg, w.Lb, s->d_stats);
    HIP_TRY(hipGetLastError());
    auto kw = s->plan.resident ? k_shadow_walk<MODE, BLOCK, COUNT, true> : k_shadow_walk<MODE, BLOCK, COUNT, false>;
    int rc = allow_lds(kw, s->plan.lds_bytes);
    if (rc != PT_OK) return rc;
    const size_t items = (size_t)n * s->dev.n_lights;
    hipLaunchKernelGGL(kw, grid_for(s->plan, items), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds,
                       run_length(s->plan, items), n, w.hit, w.sd, s->d_cnt + 5, s->d_stats);
    HIP_TRY(hipGetLastError());
    auto kb = s->dev.n_tex != 0 ? k_shade<MODE, BLOCK, COUNT, false, 2, true> : k_shade<MODE, BLOCK, COUNT, false, 2, false>;
    hipLaunchKernelGGL(kb, grid_for(s->plan, n), dim3(BLOCK), 0, st, s->dev, s->plan.lds, kRun, F, depth, n, in, w.hit, out, w.sd,
                       w.chunk_mask, w.chunk_base, w.block_tot, s->d_cnt, w.Lr, w.Lg, w.Lb, s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE, bool COUNT>
static int launch_trace_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz,
                            const float *dx, const float *dy, const float *dz, float *t, int32_t *tri, float *u,
                            float *v, hipStream_t st) {
  HIP_TRY(hipMemsetAsync(s->d_cnt, 0, 8 * sizeof(uint32_t), st));
  return dispatch_plan(s->plan, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = s->plan.resident ? k_trace_api<MODE, BLOCK, COUNT, true> : k_trace_api<MODE, BLOCK, COUNT, false>;
    int rc = allow_lds(k, s->plan.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(s->plan, n), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds, run_length(s->plan, n), n, ox, oy, oz,
                       dx, dy, dz, t, tri, u, v, s->d_cnt + 2, s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE>
static int launch_any_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                          const float *dy, const float *dz, uint8_t *hit, hipStream_t st) {
  HIP_TRY(hipMemsetAsync(s->d_cnt, 0, 8 * sizeof(uint32_t), st));
  return dispatch_plan(s->plan, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = s->plan.resident ? k_any_api<MODE, BLOCK, true> : k_any_api<MODE, BLOCK, false>;
    int rc = allow_lds(k, s->plan.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(s->plan, n), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds, run_length(s->plan, n), n, ox, oy, oz,
                       dx, dy, dz, hit, s->d_cnt + 2);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE>
static int launch_hits_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                           const float *dy, const float *dz, int32_t *hits, hipStream_t st) {
  HIP_TRY(hipMemsetAsync(s->d_cnt, 0, 8 * sizeof(uint32_t), st));
  return dispatch_plan(s->plan, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = s->plan.resident ? k_hits_api<MODE, BLOCK, true> : k_hits_api<MODE, BLOCK, false>;
    int rc = allow_lds(k, s->plan.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(s->plan, n), dim3(BLOCK), s->plan.lds_bytes, st, s->dev, s->plan.lds, run_length(s->plan, n), n, ox, oy, oz,
                       dx, dy, dz, s->d_levels, hits, s->d_cnt + 2);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

static int make_frame(const PtCamera *cam, int W, int H, uint64_t seed, const PtOptions *opt, uint32_t n_lights,
                      FrameDev &F) {
  if (!cam) return fail(PT_ERR_INVALID, "null camera");
  if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > (1ull << 28)) return fail(PT_ERR_INVALID, "bad image size");
  if (cam->vw == 0) return fail(PT_ERR_INVALID, "camera vw must not be 0");
  if (cam->aperture > 0 && !(cam->focus > 0)) return fail(PT_ERR_INVALID, "focus must be > 0 when aperture > 0");
  PtOptions o;
  if (opt) o = *opt; else pt_options_default(&o);
  if (!(o.rr_prob > 0.0f)) return fail(PT_ERR_INVALID, "rr_prob must be > 0");
  memset(&F, 0, sizeof(F));
  F.W = W; F.H = H; F.P = (uint32_t)W * (uint32_t)H;
  for (int k = 0; k < 3; ++k) { F.origin[k] = cam->origin[k]; F.bg[k] = o.bgcolor[k]; }
  // vector_::rotate, math.cpp:130-140: each coefficient is a double expression of double trig
  const double a = (double)cam->euler[0], b = (double)cam->euler[1], y = (double)cam->euler[2];
  const double ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cy = std::cos(y), sy = std::sin(y);
  F.rot[0] = cb * cy; F.rot[1] = sa * sb * cy - ca * sy; F.rot[2] = ca * sb * cy + sa * sy;
  F.rot[3] = cb * sy; F.rot[4] = sa * sb * sy + ca * cy; F.rot[5] = ca * sb * sy - sa * cy;
  F.rot[6] = -sb;     F.rot[7] = sa * cb;                F.rot[8] = ca * cb;
  F.vw = (float)cam->vw;
  F.neg_vh = (float)(-cam->vh);
  F.pin_z = (float)(1 / (2 * std::tan(cam->fov_deg / 2 * (M_PI / 180)) / cam->vw));  // camera.cpp:167
  F.aperture = cam->aperture; F.focus = cam->focus;
  F.seed = seed;
  F.stride = o.spp_stride ? o.spp_stride : 1;
  F.rr_prob = o.rr_prob; F.rr_min_depth = o.rr_min_depth; F.max_depth = o.max_depth; F.sampling = o.sampling;
  F.pdf_f = (float)(1 / (2 * M_PI) * (double)o.rr_prob);
  F.pi_f = (float)M_PI;
  F.dims_per_vertex = (3u * n_lights + 3u + 1u) & ~1u;
  return PT_OK;
}

extern "C" {

const char *pt_last_error(void) { return g_err.c_str(); }

int pt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(PT_ERR_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

void pt_options_default(PtOptions *o) {
  if (!o) return;
  o->rr_prob = 0.7f;      // camera.cpp:118
  o->rr_min_depth = 2;    // camera.cpp:119
  o->max_depth = 0;
  o->sampling = PT_SAMPLE_UNIFORM;
  o->bgcolor[0] = o->bgcolor[1] = o->bgcolor[2] = 0.0f;  // camera.cpp:52
  o->trace_mode = PT_TRACE_FAST;
  o->spp_stride = 1;
  o->batch_spp = 0;
}

void pt_camera_default(PtCamera *c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->fov_deg = 65.0f;  // camera.cpp:58
  c->vw = 1; c->vh = 1;  // camera.cpp:55
  c->aperture = 0.0f; c->focus = 1.0f;
}

void pt_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

void pt_scene_free(PtScene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  free_workspace(s->ws);
  if (s->d_nodes) (void)hipFree(s->d_nodes);
  if (s->d_tris) (void)hipFree(s->d_tris);
  if (s->d_mats) (void)hipFree(s->d_mats);
  if (s->d_lights) (void)hipFree(s->d_lights);
  if (s->d_ltris) (void)hipFree(s->d_ltris);
  if (s->d_levels) (void)hipFree(s->d_levels);
  if (s->d_tri_uv) (void)hipFree(s->d_tri_uv);
  if (s->d_textures) (void)hipFree(s->d_textures);
  if (s->d_texels) (void)hipFree(s->d_texels);
  if (s->d_spill) (void)hipFree(s->d_spill);
  if (s->d_cnt) (void)hipFree(s->d_cnt);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->h_cnt) (void)hipHostFree(s->h_cnt);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  for (int k = 0; k < 4; ++k)
    if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
  delete s;
}

static int upload(PtScene *s, const ptl::FlatScene &f) {
  HIP_TRY(hipSetDevice(s->device));
  auto up = [&](auto **dst, const auto &vec) -> hipError_t {
    const size_t bytes = vec.size() * sizeof(vec[0]);
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    s->bytes += bytes;
    return bytes ? hipMemcpy(*dst, vec.data(), bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  HIP_TRY(up(&s->d_nodes, f.nodes));
  HIP_TRY(up(&s->d_tris, f.tris));
  HIP_TRY(up(&s->d_mats, f.mats));
  HIP_TRY(up(&s->d_lights, f.lights));
  HIP_TRY(up(&s->d_ltris, f.light_tris));
  HIP_TRY(up(&s->d_levels, f.tri_levels));
  if (!f.textures.empty()) {
    HIP_TRY(up(&s->d_tri_uv, f.tri_uv));
    HIP_TRY(up(&s->d_textures, f.textures));
    HIP_TRY(up(&s->d_texels, f.texels));
  }
  s->dev.tri_uv = s->d_tri_uv; s->dev.textures = s->d_textures; s->dev.texels = s->d_texels;
  s->dev.n_tex = (uint32_t)f.textures.size();
  HIP_TRY(hipMalloc(&s->d_cnt, 8 * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&s->d_stats, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(s->d_stats, 0, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc(&s->h_cnt, 8 * sizeof(uint32_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc(&s->h_stats, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  s->dev.nodes = s->d_nodes; s->dev.tris = s->d_tris; s->dev.mats = s->d_mats;
  s->dev.lights = s->d_lights; s->dev.light_tris = s->d_ltris;
  s->dev.n_nodes = (uint32_t)f.nodes.size(); s->dev.n_tris = (uint32_t)f.tris.size();
  s->dev.n_lights = (uint32_t)f.lights.size(); s->dev.n_mats = (uint32_t)f.mats.size();
  s->dev.margin = f.max_abs_coord * (1.0f / 2048.0f) + 2.0e-5f;
  s->dev.quorum = 32;
  s->dev.quorum_any = 32;
  if (const char *e = std::getenv("PT_QUORUM_ANY")) {  // tuning knob
    const int q = std::atoi(e);
    if (q >= 1 && q <= 64) s->dev.quorum_any = (uint32_t)q;
  }
  s->dev.refill = 8;
  if (const char *e = std::getenv("PT_REFILL")) {  // tuning knob
    const int r = std::atoi(e);
    if (r >= 1 && r <= 64) s->dev.refill = (uint32_t)r;
  }
  if (const char *e = std::getenv("PT_QUORUM")) {  // tuning knob
    const int q = std::atoi(e);
    if (q >= 1 && q <= 64) s->dev.quorum = (uint32_t)q;
  }
  s->max_depth = f.max_depth; s->n_meshes = f.n_meshes;
  const int rc = make_plan(s);
  // measured: visibility walks outside LDS do best testing parked leaves at a quarter of the
  // unfinished lanes (terrain with 64 lights -5 %), those in LDS at half
  if (rc == PT_OK && !s->plan.resident && !std::getenv("PT_QUORUM_ANY")) s->dev.quorum_any = 16;
  return rc;
}

int pt_scene_upload(const PtSceneDesc *desc, int device, PtScene **out) {
  if (!desc || !out) return fail(PT_ERR_INVALID, "null argument");
  *out = nullptr;
  int ndev = pt_device_count();
  if (ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device: " + g_err);
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device ordinal out of range");
  ptl::FlatScene flat;
  std::string err;
  int rc;
  try {
    rc = ptl::flatten_scene(*desc, flat, err);
  } catch (const std::bad_alloc &) {
    return fail(PT_ERR_NOMEM, "out of host memory while flattening the scene");
  }
  if (rc != PT_OK) return fail(rc, err);
  if (stack_class(flat.max_depth) == 0) return fail(PT_ERR_UNSUPPORTED, "tree deeper than 128 levels");
  PtScene *s = new (std::nothrow) PtScene();
  if (!s) return fail(PT_ERR_NOMEM, "out of host memory");
  s->device = device;
  rc = upload(s, flat);
  if (rc != PT_OK) {
    std::string keep = g_err;
    pt_scene_free(s);
    g_err = keep;
    return rc;
  }
  *out = s;
  return PT_OK;
}

int pt_scene_info(const PtScene *s, PtSceneInfo *info) {
  if (!s || !info) return fail(PT_ERR_INVALID, "null argument");
  info->n_meshes = s->n_meshes;
  info->n_tris = s->dev.n_tris - 1;  // without the degenerate filler
  info->n_lights = s->dev.n_lights;
  info->n_wide_nodes = s->dev.n_nodes; info->max_depth = s->max_depth; info->device = (uint32_t)s->device;
  info->device_bytes = s->bytes;
  return PT_OK;
}

// Switches the per-ray work counters (node boxes tested, triangles tested) on or off.
// Counting kernels are separate instantiations; the default kernels carry no counters.
int pt_scene_count_stats(PtScene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  s->count_stats = enable != 0;
  return PT_OK;
}

static int trace_dev(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                     const float *dy, const float *dz, float *t, int32_t *tri, float *u, float *v, int mode,
                     hipStream_t st) {
  if (n == 0) return PT_OK;
  if (n > ((size_t)1 << 36)) return fail(PT_ERR_INVALID, "too many rays for one call");  // the work-queue head counts chunks of 64 in 32 bits
  if (mode == PT_TRACE_EXACT)
    return s->count_stats ? launch_trace_api<PT_TRACE_EXACT, true>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st)
                          : launch_trace_api<PT_TRACE_EXACT, false>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st);
  if (mode == PT_TRACE_FAST)
    return s->count_stats ? launch_trace_api<PT_TRACE_FAST, true>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st)
                          : launch_trace_api<PT_TRACE_FAST, false>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st);
  return fail(PT_ERR_INVALID, "unknown trace mode");
}

int pt_trace_closest_dev(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                         const float *dy, const float *dz, float *t, int32_t *tri, float *u, float *v, int mode,
                         void *stream) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !t || !tri)) return fail(PT_ERR_INVALID, "null ray or result array");
  HIP_TRY(hipSetDevice(s->device));
  return trace_dev(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, mode, (hipStream_t)stream);
}

static int read_stats4(PtScene *s, uint64_t out[4]) {
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(s->h_stats, s->d_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(s->d_stats, 0, 4 * sizeof(unsigned long long)));
  for (int k = 0; k < 4; ++k) out[k] = s->h_stats[k];
  return PT_OK;
}

// Reads and clears the work counters (see pt_scene_count_stats).
int pt_scene_read_stats(PtScene *s, uint64_t *node_boxes, uint64_t *tri_tests) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  uint64_t v[4];
  int rc = read_stats4(s, v);
  if (rc != PT_OK) return rc;
  if (node_boxes) *node_boxes = v[0] + v[2];
  if (tri_tests) *tri_tests = v[1] + v[3];
  return PT_OK;
}

int pt_scene_profile(PtScene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (enable)
    for (int k = 0; k < 4; ++k)
      if (!s->ev[k]) HIP_TRY(hipEventCreate(&s->ev[k]));
  s->profile = enable != 0;
  return PT_OK;
}

int pt_scene_read_profile(PtScene *s, PtProfile *out) {
  if (!s || !out) return fail(PT_ERR_INVALID, "null argument");
  *out = s->prof;
  memset(&s->prof, 0, sizeof(s->prof));
  return PT_OK;
}

int pt_trace_closest(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                     const float *dy, const float *dz, float *t, int32_t *tri, float *u, float *v, int mode) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !t || !tri)) return fail(PT_ERR_INVALID, "null ray or result array");
  if (mode != PT_TRACE_EXACT && mode != PT_TRACE_FAST) return fail(PT_ERR_INVALID, "unknown trace mode");
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  float *buf = nullptr;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  HIP_TRY(hipMalloc(&buf, 10 * m * sizeof(float)));
  int rc = PT_OK;
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    rc = trace_dev(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, buf + 6 * m,
                   reinterpret_cast<int32_t *>(buf + 7 * m), buf + 8 * m, buf + 9 * m, mode, 0);
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(t + off, buf + 6 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri + off, buf + 7 * m, k * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && u) e = hipMemcpy(u + off, buf + 8 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && v) e = hipMemcpy(v + off, buf + 9 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  (void)hipFree(buf);
  return rc;
}

int pt_trace_any(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                 const float *dy, const float *dz, uint8_t *hit, int mode) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !hit)) return fail(PT_ERR_INVALID, "null ray or result array");
  if (mode != PT_TRACE_EXACT && mode != PT_TRACE_FAST) return fail(PT_ERR_INVALID, "unknown trace mode");
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  float *buf = nullptr;
  HIP_TRY(hipMalloc(&buf, 7 * m * sizeof(float)));
  int rc = PT_OK;
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    uint8_t *dh = reinterpret_cast<uint8_t *>(buf + 6 * m);
    rc = mode == PT_TRACE_EXACT ? launch_any_api<PT_TRACE_EXACT>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0)
                                : launch_any_api<PT_TRACE_FAST>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0);
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hit + off, dh, k, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  (void)hipFree(buf);
  return rc;
}

int pt_trace_hits(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                  const float *dy, const float *dz, int32_t *hits, int mode) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !hits)) return fail(PT_ERR_INVALID, "null ray or result array");
  if (mode != PT_TRACE_EXACT && mode != PT_TRACE_FAST) return fail(PT_ERR_INVALID, "unknown trace mode");
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  float *buf = nullptr;
  HIP_TRY(hipMalloc(&buf, 7 * m * sizeof(float)));
  int rc = PT_OK;
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    int32_t *dh = reinterpret_cast<int32_t *>(buf + 6 * m);
    rc = mode == PT_TRACE_EXACT ? launch_hits_api<PT_TRACE_EXACT>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0)
                                : launch_hits_api<PT_TRACE_FAST>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0);
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hits + off, dh, k * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  (void)hipFree(buf);
  return rc;
}

int pt_camera_rays(const PtCamera *cam, int W, int H, int device, float *dir_xyz) {
  if (!dir_xyz) return fail(PT_ERR_INVALID, "null output");
  FrameDev F;
  int rc = make_frame(cam, W, H, 0, nullptr, 0, F);
  if (rc != PT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  float *d = nullptr;
  const size_t bytes = (size_t)F.P * 3 * sizeof(float);
  HIP_TRY(hipMalloc(&d, bytes));
  hipLaunchKernelGGL(k_camera_dirs, dim3((F.P + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, F, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(dir_xyz, d, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("camera rays: ") + hipGetErrorString(e));
  return PT_OK;
}

int pt_render_dev(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count,
                  uint64_t seed, const PtOptions *opt, float *accum, PtCounters *counters, void *stream) {
  if (!s || !accum) return fail(PT_ERR_INVALID, "null scene or accumulator");
  FrameDev F;
  int rc = make_frame(cam, W, H, seed, opt, s->dev.n_lights, F);
  if (rc != PT_OK) return rc;
  PtOptions o;
  if (opt) o = *opt; else pt_options_default(&o);
  if (o.trace_mode != PT_TRACE_EXACT && o.trace_mode != PT_TRACE_FAST) return fail(PT_ERR_INVALID, "unknown trace mode");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  PtCounters c;
  memset(&c, 0, sizeof(c));
  if (spp_count == 0) {
    if (counters) *counters = c;
    return PT_OK;
  }
  // samples per pixel in flight together, bounded by memory (about 149 bytes per path)
  uint32_t batch = o.batch_spp;
  if (batch == 0) {
    // every batch ends in a long tail of nearly empty passes (half of the 44 passes of a 33-spp
    // batch of the 1080p room hold under 1 % of its rays), so fewer, larger batches are faster
    // (64 Mi -> 128 Mi paths in flight: +8 %); a quarter of the free device memory at most
    uint64_t target = 192ull << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && s->ws.cap == 0) {
      const uint64_t fit = (uint64_t)free_b / 4 / 152;
      if (fit < target) target = fit;
    }
    if (s->ws.cap > target) target = s->ws.cap;  // a workspace that exists is used in full
    batch = (uint32_t)(target / F.P);
  }
  if (batch > spp_count) batch = spp_count;
  if (s->plan.split) {
    // split shading keeps 25 bytes per (vertex, light) pair: at most 24 GB of them
    const uint64_t per_sample = (uint64_t)F.P * s->dev.n_lights * 25ull;
    const uint64_t fit = (24ull << 30) / (per_sample ? per_sample : 1);
    if (batch > fit) batch = (uint32_t)fit;
  }
  if (batch < 1) batch = 1;
  while ((uint64_t)batch * F.P > (1ull << 31) && batch > 1) batch >>= 1;
  rc = ensure_workspace(s, (size_t)batch * F.P);
  if (rc != PT_OK) return rc;
  Workspace &w = s->ws;
  const bool exact = o.trace_mode == PT_TRACE_EXACT, cs = s->count_stats;
  const bool prof = s->profile;
  auto elapsed = [&](int a, int b, double &acc) -> int {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[a], s->ev[b]));
    acc += (double)ms;
    return PT_OK;
  };
  for (uint32_t done = 0; done < spp_count; done += batch) {
    const uint32_t sb = (spp_count - done) < batch ? (spp_count - done) : batch;
    uint32_t n = sb * F.P;
    F.sample0 = spp_begin + done * F.stride;
    if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
    hipLaunchKernelGGL(k_raygen, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, F, n, w.q[0]);
    HIP_TRY(hipGetLastError());
    if (prof) {
      HIP_TRY(hipEventRecord(s->ev[3], st));
      HIP_TRY(hipStreamSynchronize(st));
      if ((rc = elapsed(0, 3, s->prof.other_ms)) != PT_OK) return rc;
    }
    c.paths += n;
    int cur = 0;  // q[cur] holds the live paths, q[cur ^ 1] receives the survivors
    for (int depth = 0; n > 0; ++depth) {
      if (depth > 100000) return fail(PT_ERR_DEVICE, "path did not terminate after 100000 vertices");
      HIP_TRY(hipMemsetAsync(s->d_cnt, 0, 8 * sizeof(uint32_t), st));
      if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
      if (exact) rc = cs ? launch_trace_queue<PT_TRACE_EXACT, true>(s, F, depth, n, w.q[cur], st) : launch_trace_queue<PT_TRACE_EXACT, false>(s, F, depth, n, w.q[cur], st);
      else rc = cs ? launch_trace_queue<PT_TRACE_FAST, true>(s, F, depth, n, w.q[cur], st) : launch_trace_queue<PT_TRACE_FAST, false>(s, F, depth, n, w.q[cur], st);
      if (rc != PT_OK) return rc;
      const uint32_t n_chunks = (n + 63) / 64, n_sblocks = (n_chunks + kScanSpan - 1) / kScanSpan;
      hipLaunchKernelGGL(k_scan_local, dim3(n_sblocks), dim3(kScanBlock), 0, st, n, n_chunks, w.alive, w.chunk_mask, w.chunk_base, w.block_tot);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, st, n_sblocks, w.block_tot, s->d_cnt);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(s->ev[1], st));
      if (exact) rc = cs ? launch_shade<PT_TRACE_EXACT, true>(s, F, depth, n, w.q[cur], w.q[cur ^ 1], st) : launch_shade<PT_TRACE_EXACT, false>(s, F, depth, n, w.q[cur], w.q[cur ^ 1], st);
      else rc = cs ? launch_shade<PT_TRACE_FAST, true>(s, F, depth, n, w.q[cur], w.q[cur ^ 1], st) : launch_shade<PT_TRACE_FAST, false>(s, F, depth, n, w.q[cur], w.q[cur ^ 1], st);
      if (rc != PT_OK) return rc;
      if (prof) HIP_TRY(hipEventRecord(s->ev[2], st));
      HIP_TRY(hipMemcpyAsync(s->h_cnt, s->d_cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (prof) {
        if ((rc = elapsed(0, 1, s->prof.trace_ms)) != PT_OK) return rc;
        if ((rc = elapsed(1, 2, s->prof.shade_ms)) != PT_OK) return rc;
        s->prof.trace_launches += 1;
        s->prof.shade_launches += 1;
      }
      cur ^= 1;
      c.rays_closest += n;
      c.rays_shadow += (uint64_t)s->h_cnt[1] * s->dev.n_lights;
      c.bounce_passes += 1;
      n = s->h_cnt[0];
    }
    if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
    hipLaunchKernelGGL(k_resolve, dim3((F.P + kBlock - 1) / kBlock), dim3(kBlock), 0, st, F.P, sb, w.Lr, w.Lg, w.Lb, accum);
    HIP_TRY(hipGetLastError());
    if (prof) {
      HIP_TRY(hipEventRecord(s->ev[3], st));
      HIP_TRY(hipStreamSynchronize(st));
      if ((rc = elapsed(0, 3, s->prof.other_ms)) != PT_OK) return rc;
    }
  }
  if (cs) {
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t v[4];
    rc = read_stats4(s, v);
    if (rc != PT_OK) return rc;
    c.node_boxes = v[0]; c.tri_tests = v[1];
    c.shadow_node_boxes = v[2]; c.shadow_tri_tests = v[3];
  }
  if (counters) *counters = c;
  return PT_OK;
}

int pt_render(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count, uint64_t seed,
              const PtOptions *opt, float *accum, PtCounters *counters) {
  if (!s || !accum) return fail(PT_ERR_INVALID, "null scene or accumulator");
  if (W <= 0 || H <= 0) return fail(PT_ERR_INVALID, "bad image size");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)W * H * 3 * sizeof(float);
  float *d = nullptr;
  HIP_TRY(hipMalloc(&d, bytes));
  hipError_t e = hipMemcpy(d, accum, bytes, hipMemcpyHostToDevice);
  int rc = PT_OK;
  if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("copy of accumulator: ") + hipGetErrorString(e));
  if (rc == PT_OK) rc = pt_render_dev(s, cam, W, H, spp_begin, spp_count, seed, opt, d, counters, nullptr);
  if (rc == PT_OK) {
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(accum, d, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
  }
  (void)hipFree(d);
  return rc;
}

int pt_resolve(const float *sum, size_t n, uint32_t spp, float *pixels) {
  if (!sum || !pixels || spp == 0) return fail(PT_ERR_INVALID, "null array or spp == 0");
  for (size_t i = 0; i < n; ++i) pixels[i] = sum[i] / (float)spp * 255.0f;
  return PT_OK;
}

}  // extern "C"

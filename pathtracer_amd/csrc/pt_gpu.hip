// pt_gpu.hip — launch code and the C ABI (include/ptgpu.h) in front of the gfx950 kernels of
// pt_kernels.h:
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
//   k_trace_api / k_any_api / k_hits_api   Seam A for callers that bring their own rays
//
// Environment variables read when a scene is uploaded are tuning knobs for experiments
// (PT_BLOCK, PT_STACK, PT_STACK_CAP, PT_LDS_BUDGET_KB, PT_LEAF, PT_QUORUM, PT_QUORUM_ANY, PT_REFILL,
// PT_NO_RESIDENT, PT_SPLIT, PT_RES_REFILL, PT_VERBOSE); none is needed for normal use.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "pt_flatten.h"
#include "pt_kernels.h"

// ===================================================================== host side

static thread_local std::string g_err;

// The buffer of pt_render_dev_static: counters, path queues, hit records and visibility rays of a
// call that may not allocate.  One copy per GPU the library runs on, held until the process ends.
// Host code never uses this name as an address: static_scratch() returns the device address that
// hipGetSymbolAddress gave for the device in question.
static __device__ __align__(256) unsigned char scratch[PT_STATIC_SCRATCH_BYTES];

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
  float *block = nullptr;  // 2 queues, the hit records and the radiance records
  QueueDev q[2];
  HitDev hit{};
  RadDev rad{};
  unsigned long long *chunk_mask = nullptr;  // one per 64 queue entries
  uint32_t *chunk_base = nullptr;
  uint32_t *block_tot = nullptr;             // one per kScanSpan chunks, at most kScanBlock
  uint8_t *alive = nullptr;                  // one per queue entry: the path goes on after this vertex
  float *shadow_block = nullptr;             // split shading only: ShadowDev arrays
  ShadowDev sd{};
  size_t shadow_items = 0;                   // capacity in (vertex, light) pairs
  bool has_shadow = false;                   // laid out with the visibility-ray arrays
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
  uint32_t quorum_any = 32;      // see SceneDev::quorum_any
  uint32_t *d_spill = nullptr;   // traversal-stack spill area (trees deeper than the LDS part of the stack)
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
  ptl::Quad4Node *d_nodes4 = nullptr;      // the fast mode's own tree
  ptl::Oct8Node *d_nodes8 = nullptr;       // its eight-child form (scenes outside LDS)
  ptl::TriRec4 *d_tris4 = nullptr;
  uint32_t *d_tri_id4 = nullptr;
  ptl::GateBox *d_gates = nullptr;
  float *d_tri_uv4 = nullptr;
  uint32_t max_depth = 0, max_depth4 = 0, max_depth8 = 0, n_meshes = 0;
  float max_abs_coord = 0;                 // largest finite vertex coordinate
  bool use_oct = false;                    // PT_TRACE_FAST runs over the eight-child tree when the scene is outside LDS
  uint32_t n_unnested = 0;                 // boxes / triangles of the description that reach outside the internal box above them
  uint64_t bytes = 0;
  Plan plan;                               // launches over the reference-shaped tree (exact mode, fast-binary mode)
  Plan plan4;                              // launches over the fast tree
  Plan plan8;                              // launches over its eight-child form
  Workspace ws;
  uint32_t *d_cnt = nullptr;               // per launch: survivors, hits, work-queue heads of trace [2], fused shade [3], visibility walk [5]
  unsigned long long *d_stats = nullptr;   // node boxes, triangle tests: closest [0,1], visibility [2,3]
  uint32_t *h_cnt = nullptr;               // pinned
  unsigned long long *h_stats = nullptr;   // pinned
  bool count_stats = false;
  bool profile = false;
  bool short_rays = false;                 // the rays of the next launches all have |d| <= 2 (SceneDev::tmin_own)
  uint32_t tail_paths = 0;                 // PT_TAIL: a batch with at most this many live paths is finished by k_tail (0: never)
  bool tail_given = false;                 // PT_TAIL was set; otherwise tail_limit() chooses
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // 4, 5: around the visibility walk of split shading
  PtProfile prof{};
  // What the launches of the call in progress use: the scene's own workspace and counter block, or
  // (pt_render_dev_static) pieces of the library's static device buffer.
  Workspace *aw = nullptr;
  uint32_t *acnt = nullptr;
  // Asynchronous calls share the counter block and the workspace, so work enqueued on a new stream
  // waits for what the previous call left on the old one.
  float *d_stage = nullptr;                // device staging of the host-pointer ray queries, kept between calls (grow only)
  size_t stage_floats = 0;
  hipEvent_t ev_last = nullptr;
  hipStream_t last_stream = nullptr;
  bool has_last = false;
};

// Device address of `scratch` per device ordinal, looked up once per device with that device current
// (every GPU holds its own copy of the array at its own address).
static const int kMaxDevices = 64;
static unsigned char *g_scratch_ptr[kMaxDevices];
// Recorded after every call that worked in a device's static buffer and waited for by the next one, so that such
// calls are ordered on the GPU whatever scenes and streams they name (the buffer is one per GPU, not per scene).
static hipEvent_t g_scratch_last[kMaxDevices];
static std::mutex g_scratch_mutex;

static int setup_static_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return PT_OK;  // such a device simply has no static entry point
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return PT_OK;
  void *ptr = nullptr;
  HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  HIP_TRY(hipEventCreateWithFlags(&g_scratch_last[device], hipEventDisableTiming));
  g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
  return PT_OK;
}
static unsigned char *static_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  return g_scratch_ptr[device];
}

// Restores the caller's current device when an entry point returns.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int enter_stream(PtScene *s, hipStream_t st) {
  if (s->has_last && s->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->ev_last, 0));
  return PT_OK;
}
static int leave_stream(PtScene *s, hipStream_t st) {
  HIP_TRY(hipEventRecord(s->ev_last, st));
  s->last_stream = st;
  s->has_last = true;
  return PT_OK;
}

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
// kind: 0 = the reference-shaped tree, 1 = the fast tree, 2 = the fast tree's eight-child form
static int make_plan(PtScene *s, int kind) {
  const bool wide = kind != 0, oct = kind == 2;
  Plan &p = oct ? s->plan8 : (wide ? s->plan4 : s->plan);
  const uint32_t depth = oct ? s->max_depth8 : (wide ? s->max_depth4 : s->max_depth);
  const uint32_t n_nodes = oct ? s->dev.n_nodes8 : (wide ? s->dev.n_nodes4 : s->dev.n_nodes), n_tris = wide ? s->dev.n_tris4 : s->dev.n_tris;
  // the fast tree's nodes are kept without their padding pieces
  const size_t node_lds = oct ? 5 * 16 : (wide ? 7 * 16 : sizeof(ptl::WideNode));
  const size_t gate_bytes = wide ? (size_t)s->dev.n_gates * sizeof(ptl::GateBox) : 0;
  p.stack = stack_class(depth);
  if (!p.stack) return fail(PT_ERR_UNSUPPORTED, "tree deeper than 128 levels");
  // deeper trees keep only the first entries in LDS and spill the rest to global memory, so
  // that LDS still holds 16 waves per CU and a useful part of the tree
  int cap = wide ? 16 : 24;  // a descent of the four-child tree rarely holds more than a dozen entries
  if (oct) cap = 16;         // two words per node with siblings left to visit: eight pairs, measured against 8 / 12 / 24 entries (more of the tree's top fits beside a shorter stack; at 8 the 7 M triangles spill too often)
  if (const char *e = std::getenv("PT_STACK_CAP")) {  // tuning knob
    const int c = std::atoi(e);
    if (c >= 4 && c <= 128) cap = c;
  }
  if (p.stack > cap) p.stack = cap;
  p.block = p.stack <= 32 ? 1024 : (p.stack <= 64 ? 512 : 256);
  bool block_given = false;
  if (const char *e = std::getenv("PT_BLOCK")) {  // tuning knob
    const int b = std::atoi(e);
    if ((b == 256 || b == 512 || b == 1024) && (size_t)p.stack * b * 4 <= 128 * 1024) { p.block = b; block_given = true; }
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  const size_t lds_cu = 160 * 1024;
  if (const char *e = std::getenv("PT_STACK")) {  // tuning knob: entries per lane kept in LDS
    const int st = std::atoi(e);
    if (st >= 4 && st <= 128 && (size_t)st * p.block * 4 <= 128 * 1024) p.stack = st;
  }
  if (oct) p.stack &= ~1;  // its entries are pairs of words (push_pair)
  const size_t node_bytes = (size_t)n_nodes * node_lds;
  const size_t tri_bytes = (size_t)n_tris * sizeof(ptl::TriRec);
  // lays the LDS of a workgroup out for p.block threads with at most `share` bytes of the CU's LDS
  auto layout = [&](size_t share) {
    const size_t stack_bytes = (size_t)p.stack * p.block * 4;
    size_t budget = share > stack_bytes + 512 ? share - stack_bytes - 512 : 0;
    if (const char *e = std::getenv("PT_LDS_BUDGET_KB")) {
      const size_t kb = (size_t)std::atol(e) * 1024;
      if (kb < budget) budget = kb;
    }
    p.lds.n_gates = 0;
    if (!oct && node_bytes + tri_bytes + gate_bytes <= budget) {
      p.lds.n_nodes = n_nodes;
      p.lds.n_tris = n_tris;
      p.lds.n_gates = wide ? s->dev.n_gates : 0;
    } else {
      const size_t fit = budget / node_lds;
      p.lds.n_nodes = (uint32_t)(fit < n_nodes ? fit : n_nodes);
      p.lds.n_tris = 0;
    }
    p.lds_bytes = (size_t)p.lds.n_nodes * node_lds + (size_t)p.lds.n_tris * sizeof(ptl::TriRec) +
                  (size_t)p.lds.n_gates * sizeof(ptl::GateBox) + stack_bytes;
    unsigned per_cu = (unsigned)(lds_cu / p.lds_bytes);
    const unsigned by_threads = 2048u / (unsigned)p.block;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    p.max_blocks = per_cu * (unsigned)prop.multiProcessorCount;
    return p.lds.n_nodes == n_nodes && p.lds.n_tris == n_tris && !std::getenv("PT_NO_RESIDENT");
  };
  p.resident = layout(lds_cu) && !oct;  // the eight-child tree is for scenes outside LDS: it has no all-in-LDS kernels
  if (!p.resident && !block_given && p.stack <= 32) {
    // A scene outside LDS waits on memory, not on LDS: measured, the staged top of the tree buys nothing
    // there (160 KiB or 8 KiB of it: the same rate), while a fifth wave per SIMD does for the visibility
    // walks (64 lights: +11 %).  So: workgroups of 256 threads, and the visibility kernel is compiled to fit
    // five waves per SIMD at that size.  A sixth of the CU's LDS each: with a fifth (32 KiB) the fifth
    // workgroup was measured not to become resident.
    p.block = 256;
    // workgroups per CU the LDS is cut for (four waves each)
    // Measured with 69-72 registers per lane (up to seven waves per SIMD): 260 k triangles 6.43 / 6.54 / 6.12 G
    // rays/s at 5 / 6 / 7 workgroups, 1 M triangles with 64 lights 5.67 / 6.10 / 6.02, 7 M triangles 3.51 / 3.26 /
    // 3.16: more rays in flight help until their working set outgrows the caches, which the 7 M triangles
    // (over 600 MB of nodes and triangles against 256 MB of Infinity Cache) do at once.
    const size_t tree_bytes = node_bytes + tri_bytes;
    size_t share = tree_bytes > (200u << 20) ? 5 : 6;
    if (const char *e = std::getenv("PT_LDS_SHARE")) {  // tuning knob
      const int v = std::atoi(e);
      if (v >= 4 && v <= 8) share = (size_t)v;
    }
    p.resident = layout(lds_cu / share - 1024) && !oct;  // a little under the share, so that exactly `share` workgroups become resident
  }
  p.lds.stack_lds = (uint32_t)p.stack;
  p.lds.spill = nullptr;
  p.split = !p.resident;
  if (const char *e = std::getenv("PT_RES_REFILL")) p.refill_resident = std::atoi(e) != 0;  // tuning knob
  if (const char *e = std::getenv("PT_SPLIT")) p.split = std::atoi(e) != 0;  // tuning knob
  if (std::getenv("PT_VERBOSE"))
    std::fprintf(stderr, "ptgpu plan (%s tree): depth %u, block %d, LDS stack %d entries/lane, %u nodes + %u triangles + %u gates in LDS, %zu B LDS per workgroup, %u resident workgroups\n",
                 oct ? "fast, eight children" : (wide ? "fast" : "reference-shaped"), depth, p.block, p.stack, p.lds.n_nodes, p.lds.n_tris, p.lds.n_gates, p.lds_bytes, p.max_blocks);
  if (depth > (uint32_t)p.stack) {
    const size_t words = (size_t)p.max_blocks * p.block * (depth - (uint32_t)p.stack);
    HIP_TRY(hipMalloc(&p.d_spill, words * sizeof(uint32_t)));
    s->bytes += words * sizeof(uint32_t);
    p.lds.spill = p.d_spill;
  }
  // measured with the fast tree (1080p / 4K frames, G rays/s): visibility walks test their parked leaves
  // best at half of the unfinished lanes, in LDS and outside (64 lights: 4.7 at an eighth, 4.9 at a
  // quarter, 5.4 at half, 5.2 at three quarters)
  p.quorum_any = s->dev.quorum_any;
  return PT_OK;
}

// Live paths at or below which a batch is handed to the tail kernel.  Measured at 1080p (G rays/s at 16 / 64
// samples per pixel): the room in LDS 13.9 / 16.8 at 65 536, 15.6 / 17.4 anywhere from 3 M to 16 M; the 7 M
// triangles outside LDS at 4 / 16 samples 2.56 / 3.49 at 128 Ki, 2.65 / 3.55 at 512 Ki, 2.72 / 3.58 at 1 Mi,
// 2.67 / 3.55 at 2 Mi.
// The tail kernel walks a vertex's visibility rays one light after the other: with many lights the passes of
// the wavefront pipeline stay cheaper to the very end (64 lights: 12 ms of tail for 2 000 paths).
static uint32_t tail_limit(const PtScene *s, int mode) {
  if (s->tail_given) return s->tail_paths;
  if (s->dev.n_lights > 4) return 0;
  return (mode_oct(mode) ? s->plan8 : (mode_wide(mode) ? s->plan4 : s->plan)).resident ? (4u << 20) : (1u << 20);
}

// Plan and device view of the scene for a traversal mode.
static const Plan &plan_of(const PtScene *s, int mode) { return mode_oct(mode) ? s->plan8 : (mode_wide(mode) ? s->plan4 : s->plan); }
static SceneDev dev_of(const PtScene *s, int mode) {
  SceneDev d = s->dev;
  d.quorum_any = plan_of(s, mode).quorum_any;
  d.tmin_own = s->short_rays ? 1e-5f : PT_OWN_BOX_TMIN;
  return d;
}

static void free_workspace(Workspace &w) {
  if (w.block) (void)hipFree(w.block);
  if (w.shadow_block) (void)hipFree(w.shadow_block);
  w = Workspace();
}

// Bytes of the two parts of a workspace for `stride` paths in flight (stride a multiple of 64):
// the queues, hit and radiance arrays with the compaction tables, and (split shading only) the
// visibility-ray arrays: 3 floats per vertex, 6 floats + 1 byte per (vertex, light).
static const size_t kWsArrays = (2 * kQueueBytesPerPath + kHitBytesPerPath + kRadBytesPerPath) / sizeof(float);
static size_t ws_main_bytes(size_t stride) {
  const size_t n_chunks = stride / 64;
  return kWsArrays * stride * sizeof(float) + n_chunks * 12 + (size_t)kScanBlock * 4 + 256 + stride;
}
static size_t ws_shadow_bytes(size_t stride, uint32_t n_lights, bool split) {
  if (!split) return 0;
  const size_t items = stride * n_lights;
  return (3 * stride + 6 * items) * sizeof(float) + items;
}
static size_t ws_stride(size_t cap) { return (cap + 63) & ~(size_t)63; }

// Lays a workspace for `cap` paths out over memory the caller provides.
static int carve_workspace(Workspace &w, size_t cap, uint32_t n_lights, bool split, float *main_block, float *shadow_block) {
  const size_t stride = ws_stride(cap);
  const size_t n_chunks = stride / 64;
  if ((n_chunks + kScanSpan - 1) / kScanSpan > (size_t)kScanBlock || stride > 0x7fffffffu)
    return fail(PT_ERR_UNSUPPORTED, "too many paths in flight");
  float *p = main_block;
  for (int k = 0; k < 2; ++k) {
    w.q[k].base = p;
    w.q[k].stride = (uint32_t)stride;
    p += kQueueBytesPerPath / sizeof(float) * stride;
  }
  w.hit.base = p;
  w.hit.stride = (uint32_t)stride;
  p += kHitBytesPerPath / sizeof(float) * stride;
  w.rad.base = p;
  w.rad.stride = (uint32_t)stride;
  p += kRadBytesPerPath / sizeof(float) * stride;
  w.chunk_mask = reinterpret_cast<unsigned long long *>(p);
  w.chunk_base = reinterpret_cast<uint32_t *>(w.chunk_mask + n_chunks);
  w.block_tot = w.chunk_base + n_chunks;
  w.alive = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(w.block_tot + kScanBlock) + 63) & ~(uintptr_t)63);
  w.shadow_items = 0;
  w.has_shadow = split;
  if (split) {
    const size_t items = stride * n_lights;
    w.sd.vbase = shadow_block;
    w.sd.vstride = (uint32_t)stride;
    w.sd.ibase = shadow_block + 3 * stride;
    w.sd.istride = items;
    w.sd.vis_ = reinterpret_cast<uint8_t *>(w.sd.ibase + 6 * items);
    w.shadow_items = items;
  }
  w.cap = cap;
  return PT_OK;
}

static int ensure_workspace(PtScene *s, size_t cap, bool split) {
  Workspace &w = s->ws;
  if (w.cap >= cap && (w.has_shadow || !split)) return PT_OK;
  if (w.cap > cap) cap = w.cap;
  split = split || w.has_shadow;
  free_workspace(w);
  const size_t stride = ws_stride(cap);
  if ((stride / 64 + kScanSpan - 1) / kScanSpan > (size_t)kScanBlock || stride > 0x7fffffffu)
    return fail(PT_ERR_UNSUPPORTED, "too many paths in flight");
  if (hipMalloc(&w.block, ws_main_bytes(stride)) != hipSuccess) {
    w.block = nullptr;
    (void)hipGetLastError();
    return fail(PT_ERR_NOMEM, "out of device memory for the path workspace");
  }
  const size_t sb = ws_shadow_bytes(stride, s->dev.n_lights, split);
  if (sb && hipMalloc(&w.shadow_block, sb) != hipSuccess) {
    w.shadow_block = nullptr;
    (void)hipGetLastError();
    free_workspace(w);
    return fail(PT_ERR_NOMEM, "out of device memory for the visibility-ray workspace");
  }
  return carve_workspace(w, cap, s->dev.n_lights, split, w.block, w.shadow_block);
}

// Dynamic LDS above 64 KiB has to be allowed per kernel once.
template <typename K>
static int allow_lds(K kernel, size_t bytes) {
  // the attribute is one value per kernel and device: remember the largest one set and raise it only
  // when a launch needs more (the call costs microseconds and a frame has hundreds of launches)
  struct Seen { const void *fn; int dev; size_t bytes; };
  static thread_local std::vector<Seen> done;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const void *fn = reinterpret_cast<const void *>(kernel);
  Seen *hit = nullptr;
  for (auto &e : done)
    if (e.fn == fn && e.dev == dev) hit = &e;
  if (hit && hit->bytes >= bytes) return PT_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (hit) hit->bytes = bytes; else done.push_back(Seen{fn, dev, bytes});
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
static int launch_trace_queue(PtScene *s, const PathFrame &F, int depth, uint32_t n, const QueueDev &q, hipStream_t st) {
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  return dispatch_plan(P, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = !P.resident ? k_trace_queue<MODE, BLOCK, COUNT, false, true>
             : P.refill_resident ? k_trace_queue<MODE, BLOCK, COUNT, true, true> : k_trace_queue<MODE, BLOCK, COUNT, true, false>;
    int rc = allow_lds(k, P.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(P, n), dim3(BLOCK), P.lds_bytes, st, D, P.lds, run_length(P, n), F, depth, n, q,
                       s->aw->hit, s->aw->alive, s->acnt + 2, s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE, bool COUNT>
static int launch_shade(PtScene *s, const PathFrame &F, int depth, uint32_t n, const QueueDev &in,
                        const QueueDev &out, hipStream_t st) {
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  return dispatch_plan(P, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    const Workspace &w = *s->aw;
    const uint32_t kRun = run_length(P, n);
    if (!P.split) {
      const bool tex = s->dev.n_tex != 0;
      auto k = P.resident ? (tex ? k_shade<MODE, BLOCK, COUNT, true, 0, true> : k_shade<MODE, BLOCK, COUNT, true, 0, false>)
                                : (tex ? k_shade<MODE, BLOCK, COUNT, false, 0, true> : k_shade<MODE, BLOCK, COUNT, false, 0, false>);
      int rc = allow_lds(k, P.lds_bytes);
      if (rc != PT_OK) return rc;
      hipLaunchKernelGGL(k, grid_for(P, n), dim3(BLOCK), P.lds_bytes, st, D, P.lds, kRun, F, depth, n, in,
                         w.hit, out, w.sd, w.chunk_mask, w.chunk_base, w.block_tot, s->acnt, w.rad, s->d_stats);
      HIP_TRY(hipGetLastError());
      return PT_OK;
    }
    // prepare the visibility rays, walk them (with per-lane refill when the scene is outside LDS), finish the vertices
    if (ShadowDev::items(n, s->dev.n_lights) > w.shadow_items) return fail(PT_ERR_DEVICE, "visibility-ray workspace too small");
    auto ka = k_shade<MODE, BLOCK, COUNT, false, 1, false>;
    hipLaunchKernelGGL(ka, grid_for(P, n), dim3(BLOCK), 0, st, D, P.lds, kRun, F, depth, n, in, w.hit, out, w.sd,
                       w.chunk_mask, w.chunk_base, w.block_tot, s->acnt, w.rad, s->d_stats);
    HIP_TRY(hipGetLastError());
    auto kw = P.resident ? k_shadow_walk<MODE, BLOCK, COUNT, true> : k_shadow_walk<MODE, BLOCK, COUNT, false>;
    int rc = allow_lds(kw, P.lds_bytes);
    if (rc != PT_OK) return rc;
    const size_t items = (size_t)ShadowDev::walk_items(n, s->dev.n_lights, D.cohort != 0u);
    if (s->profile) HIP_TRY(hipEventRecord(s->ev[4], st));
    hipLaunchKernelGGL(kw, grid_for(P, items), dim3(BLOCK), P.lds_bytes, st, D, P.lds,
                       run_length(P, items), n, w.hit, w.sd, s->acnt + 5, s->d_stats);
    HIP_TRY(hipGetLastError());
    if (s->profile) HIP_TRY(hipEventRecord(s->ev[5], st));
    auto kb = s->dev.n_tex != 0 ? k_shade<MODE, BLOCK, COUNT, false, 2, true> : k_shade<MODE, BLOCK, COUNT, false, 2, false>;
    hipLaunchKernelGGL(kb, grid_for(P, n), dim3(BLOCK), 0, st, D, P.lds, kRun, F, depth, n, in, w.hit, out, w.sd,
                       w.chunk_mask, w.chunk_base, w.block_tot, s->acnt, w.rad, s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE, bool COUNT>
static int launch_tail(PtScene *s, const PathFrame &F, int depth, uint32_t n, const QueueDev &q, hipStream_t st) {
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  // The kernel holds a whole path's state beside the walk's: about 180 registers per lane.  Workgroups of 1024
  // would have 128 and spill; 512 (or the plan's, if smaller) fit.  The plan's LDS layout and spill area were
  // sized for its own workgroup size and hold the smaller one.
  auto go = [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    const bool tex = s->dev.n_tex != 0;
    auto k = P.resident ? (tex ? k_tail<MODE, BLOCK, COUNT, true, true> : k_tail<MODE, BLOCK, COUNT, true, false>)
                        : (tex ? k_tail<MODE, BLOCK, COUNT, false, true> : k_tail<MODE, BLOCK, COUNT, false, false>);
    int rc = allow_lds(k, P.lds_bytes);
    if (rc != PT_OK) return rc;
    // few paths, long dependent chains: spread the waves over every CU and let each claim little at a time
    size_t g = ((size_t)n + 63) / 64;
    if (g > P.max_blocks) g = P.max_blocks;
    if (g < 1) g = 1;
    const size_t waves = g * (size_t)(BLOCK / 64);
    size_t run = ((size_t)n + 63) / 64 / waves;
    if (run < 1) run = 1;
    if (run > 4) run = 4;
    hipLaunchKernelGGL(k, dim3((unsigned)g), dim3(BLOCK), P.lds_bytes, st, D, P.lds, (uint32_t)run, F, depth, n, q, s->aw->rad, s->acnt,
                       s->d_stats);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  };
  if (P.block == 256) return go(std::integral_constant<int, 256>{});
  return go(std::integral_constant<int, 512>{});
}

template <int MODE, bool COUNT>
static int launch_trace_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz,
                            const float *dx, const float *dy, const float *dz, float *t, int32_t *tri, float *u,
                            float *v, hipStream_t st, RaySide side) {
  HIP_TRY(hipMemsetAsync(s->acnt, 0, 8 * sizeof(uint32_t), st));
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  return dispatch_plan(P, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = P.resident ? k_trace_api<MODE, BLOCK, COUNT, true> : k_trace_api<MODE, BLOCK, COUNT, false>;
    int rc = allow_lds(k, P.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(P, n), dim3(BLOCK), P.lds_bytes, st, D, P.lds, run_length(P, n), n, ox, oy, oz,
                       dx, dy, dz, t, tri, u, v, s->acnt + 2, s->d_stats, side);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE>
static int launch_any_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                          const float *dy, const float *dz, uint8_t *hit, hipStream_t st, RaySide side) {
  HIP_TRY(hipMemsetAsync(s->acnt, 0, 8 * sizeof(uint32_t), st));
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  return dispatch_plan(P, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = P.resident ? k_any_api<MODE, BLOCK, true> : k_any_api<MODE, BLOCK, false>;
    int rc = allow_lds(k, P.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(P, n), dim3(BLOCK), P.lds_bytes, st, D, P.lds, run_length(P, n), n, ox, oy, oz,
                       dx, dy, dz, hit, s->acnt + 2, side);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

template <int MODE>
static int launch_hits_api(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                           const float *dy, const float *dz, int32_t *hits, hipStream_t st, RaySide side) {
  HIP_TRY(hipMemsetAsync(s->acnt, 0, 8 * sizeof(uint32_t), st));
  const Plan &P = plan_of(s, MODE);
  const SceneDev D = dev_of(s, MODE);
  return dispatch_plan(P, [&](auto BL) -> int {
    constexpr int BLOCK = decltype(BL)::value;
    auto k = P.resident ? k_hits_api<MODE, BLOCK, true> : k_hits_api<MODE, BLOCK, false>;
    int rc = allow_lds(k, P.lds_bytes);
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k, grid_for(P, n), dim3(BLOCK), P.lds_bytes, st, D, P.lds, run_length(P, n), n, ox, oy, oz,
                       dx, dy, dz, s->d_levels, hits, s->acnt + 2, side);
    HIP_TRY(hipGetLastError());
    return PT_OK;
  });
}

// Calls f(M) with the traversal mode as an integral constant.
static bool valid_mode(int mode) { return mode == PT_TRACE_EXACT || mode == PT_TRACE_FAST || mode == PT_TRACE_FAST_BINARY; }
// The pruning walk of the reference-shaped tree skips leaves by their bounds and by distance; an
// ill-conditioned triangle can report a hit outside both (triangle_is_ill_conditioned), so a scene that has
// one is walked exactly instead.  (The fast tree deals with such triangles itself.)
// The fast tree accepts a hit when the ray passes the lowest internal box above the triangle; that equals the
// reference's test of every internal box above it only when those boxes are nested (each inside its parent's,
// the triangles inside the lowest).  The reference's builder makes them so; a description whose boxes are not
// (refitted, padded or hand-made) is walked exactly.
static int effective_mode(const PtScene *s, int mode) {
  if (mode == PT_TRACE_FAST_BINARY && s->dev.n_ill != 0) return PT_TRACE_EXACT;
  if (mode == PT_TRACE_FAST && s->n_unnested != 0) return PT_TRACE_EXACT;
  if (mode == PT_TRACE_FAST && s->use_oct && !s->plan4.resident) return PT_TRACE_FAST_OCT;
  return mode;
}
// Origins up to this distance from the world origin (per coordinate) are inside the range of the pruning walks.
static float near_limit(const PtScene *s) { return 8.0f * s->max_abs_coord; }
// Largest |coordinate| among n origins on the host; NaN counts as beyond every limit.
static bool host_origins_within(const float *ox, const float *oy, const float *oz, size_t n, float limit) {
  bool ok = true;
  for (size_t i = 0; i < n; ++i) ok = ok && (std::fabs(ox[i]) <= limit) && (std::fabs(oy[i]) <= limit) && (std::fabs(oz[i]) <= limit);
  return ok;
}
template <typename F>
static int with_mode(int mode, F &&f) {
  switch (mode) {
    case PT_TRACE_EXACT: return f(std::integral_constant<int, PT_TRACE_EXACT>{});
    case PT_TRACE_FAST: return f(std::integral_constant<int, PT_TRACE_FAST>{});
    case PT_TRACE_FAST_BINARY: return f(std::integral_constant<int, PT_TRACE_FAST_BINARY>{});
    case PT_TRACE_FAST_OCT: return f(std::integral_constant<int, PT_TRACE_FAST_OCT>{});
    default: return fail(PT_ERR_INVALID, "unknown trace mode");
  }
}

static PathFrame path_frame(const FrameDev &F) {
  PathFrame p;
  p.P = F.P; p.seed = F.seed; p.sample0 = F.sample0; p.stride = F.stride;
  p.rr_prob = F.rr_prob; p.rr_prob_f = F.rr_prob_f; p.rr_min_depth = F.rr_min_depth; p.max_depth = F.max_depth; p.sampling = F.sampling;
  for (int k = 0; k < 3; ++k) p.bg[k] = F.bg[k];
  p.pdf_f = F.pdf_f; p.pi_f = F.pi_f; p.dims_per_vertex = F.dims_per_vertex;
  return p;
}

static int make_frame(const PtCamera *cam, int W, int H, uint64_t seed, const PtOptions *opt, uint32_t n_lights,
                      FrameDev &F) {
  if (!cam) return fail(PT_ERR_INVALID, "null camera");
  if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > (1ull << 28)) return fail(PT_ERR_INVALID, "bad image size");
  if (cam->vw == 0) return fail(PT_ERR_INVALID, "camera vw must not be 0");
  if (cam->aperture > 0 && !(cam->focus > 0)) return fail(PT_ERR_INVALID, "focus must be > 0 when aperture > 0");
  PtOptions o;
  if (opt) o = *opt; else pt_options_default(&o);
  if (!(o.rr_prob > 0.0)) return fail(PT_ERR_INVALID, "rr_prob must be > 0");
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
  F.rr_prob = o.rr_prob; F.rr_prob_f = (float)o.rr_prob; F.rr_min_depth = o.rr_min_depth; F.max_depth = o.max_depth; F.sampling = o.sampling;
  F.pdf_f = (float)(1 / (2 * M_PI) * o.rr_prob);  // camera.cpp:152, narrowed by operator/(vec, scalar)
  F.pi_f = (float)M_PI;
  F.dims_per_vertex = (3u * n_lights + 3u + 1u) & ~1u;
  if (const char *e = std::getenv("PT_QUEUE_ORDER")) F.pixel_major = std::strcmp(e, "pixel") == 0;  // measuring knob
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
  o->rr_prob = .7;        // camera.cpp:118
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

int pt_roulette_pass(double rr_prob, float draw) { return roulette_pass(rr_prob, draw) ? 1 : 0; }

void pt_scene_free(PtScene *s) {
  if (!s) return;
  DeviceGuard guard;
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
  if (s->plan.d_spill) (void)hipFree(s->plan.d_spill);
  if (s->plan4.d_spill) (void)hipFree(s->plan4.d_spill);
  if (s->plan8.d_spill) (void)hipFree(s->plan8.d_spill);
  if (s->d_nodes8) (void)hipFree(s->d_nodes8);
  if (s->d_nodes4) (void)hipFree(s->d_nodes4);
  if (s->d_tris4) (void)hipFree(s->d_tris4);
  if (s->d_tri_id4) (void)hipFree(s->d_tri_id4);
  if (s->d_gates) (void)hipFree(s->d_gates);
  if (s->d_tri_uv4) (void)hipFree(s->d_tri_uv4);
  if (s->d_stage) (void)hipFree(s->d_stage);
  if (s->d_cnt) (void)hipFree(s->d_cnt);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->h_cnt) (void)hipHostFree(s->h_cnt);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  for (int k = 0; k < 6; ++k)
    if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
  if (s->ev_last) (void)hipEventDestroy(s->ev_last);
  delete s;
}

static int upload(PtScene *s, const ptl::FlatScene &f) {
  DeviceGuard guard;
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
  HIP_TRY(up(&s->d_nodes4, f.nodes4));
  HIP_TRY(up(&s->d_nodes8, f.nodes8));
  HIP_TRY(up(&s->d_tris4, f.tris4));
  HIP_TRY(up(&s->d_tri_id4, f.tri_id4));
  HIP_TRY(up(&s->d_gates, f.gates));
  if (!f.textures.empty()) {
    HIP_TRY(up(&s->d_tri_uv4, f.tri_uv4));
    HIP_TRY(up(&s->d_tri_uv, f.tri_uv));
    HIP_TRY(up(&s->d_textures, f.textures));
    HIP_TRY(up(&s->d_texels, f.texels));
  }
  s->dev.tri_uv = s->d_tri_uv; s->dev.textures = s->d_textures; s->dev.texels = s->d_texels;
  s->dev.n_tex = (uint32_t)f.textures.size();
  HIP_TRY(hipMalloc(&s->d_cnt, 8 * sizeof(uint32_t)));
  HIP_TRY(hipEventCreateWithFlags(&s->ev_last, hipEventDisableTiming));
  s->aw = &s->ws;
  s->acnt = s->d_cnt;
  {
    const int rc = setup_static_scratch(s->device);
    if (rc != PT_OK) return rc;
  }
  HIP_TRY(hipMalloc(&s->d_stats, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(s->d_stats, 0, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc(&s->h_cnt, 8 * sizeof(uint32_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc(&s->h_stats, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  s->dev.nodes = s->d_nodes; s->dev.tris = s->d_tris; s->dev.mats = s->d_mats;
  s->dev.lights = s->d_lights; s->dev.light_tris = s->d_ltris;
  s->dev.n_nodes = (uint32_t)f.nodes.size(); s->dev.n_tris = (uint32_t)f.tris.size();
  s->dev.n_lights = (uint32_t)f.lights.size(); s->dev.n_mats = (uint32_t)f.mats.size();
  s->dev.nodes4 = s->d_nodes4; s->dev.tris4 = s->d_tris4; s->dev.tri_id4 = s->d_tri_id4; s->dev.gates = s->d_gates;
  s->dev.tri_uv4 = s->d_tri_uv4;
  s->dev.nodes8 = s->d_nodes8; s->dev.n_nodes8 = (uint32_t)f.nodes8.size();
  s->dev.n_nodes4 = (uint32_t)f.nodes4.size(); s->dev.n_tris4 = (uint32_t)f.tris4.size(); s->dev.n_gates = (uint32_t)f.gates.size();
  s->dev.n_ill = f.n_ill;
  s->dev.margin = f.max_abs_coord * (1.0f / 2048.0f) + 2.0e-5f;
  s->max_abs_coord = f.max_abs_coord;
  s->n_unnested = f.n_unnested;
  s->dev.quorum = 32;
  s->dev.quorum_any = 32;
  if (const char *e = std::getenv("PT_QUORUM_ANY")) {  // tuning knob
    const int q = std::atoi(e);
    if (q >= 1 && q <= 64) s->dev.quorum_any = (uint32_t)q;
  }
  s->dev.refill = 16;  // idle lanes at which a wave hands out new rays; measured on 7 M / 260 k triangles: 2 -> 2.82 / 5.38, 4 -> 2.96 / 5.71, 8 -> 3.08 / 5.98, 16 -> 3.13 / 6.11, 32 -> 3.07 / 5.98 G rays/s
  s->dev.vertex_major = 1;
  if (const char *e = std::getenv("PT_SHADOW_ORDER")) s->dev.vertex_major = std::strcmp(e, "light") != 0;  // measuring knob: "light" = light by light
  if (const char *e = std::getenv("PT_SHADOW_GROUP")) {  // measuring knob: wave fills of 64/g vertices x g lights
    const int g = std::atoi(e);
    if (g >= 2 && g < 64 && 64 % g == 0 && s->dev.n_lights % (uint32_t)g == 0) s->dev.vertex_major = (uint32_t)g;
  }
  if (const char *e = std::getenv("PT_REFILL")) {  // tuning knob
    const int r = std::atoi(e);
    if (r >= 1 && r <= 64) s->dev.refill = (uint32_t)r;
  }
  s->dev.refill_any = s->dev.refill;
  if (const char *e = std::getenv("PT_REFILL_ANY")) {  // tuning knob
    const int r = std::atoi(e);
    if (r >= 1 && r <= 64) s->dev.refill_any = (uint32_t)r;
  }
  // Visibility walks of a scene with many lights take whole vertices: a wave is filled with the rays of one
  // vertex (or of a few) and hands out the next ones when all of them are done.  Rays from one point start in
  // the same nodes and stay in step; a wave that mixes the rays of a vertex just begun with stragglers of
  // earlier ones runs both of its steps, nodes and triangles, with part of its lanes (64 lights: 8.75 -> 9.30 G
  // rays/s; one light, where 64 lanes are 64 different vertices: 7.3 -> 6.8 and 4.3 -> 3.2, so not there).
  // Only where a vertex's rays fill whole waves: measured on the same terrain, 128 lights +6 %, but 8 / 16 / 32 /
  // 48 / 96 lights (several vertices per wave, or lanes left empty) -11 / -15 / -6 / -10 / -10 %.
  s->dev.cohort = s->dev.vertex_major == 1u && s->dev.n_lights >= 64u && s->dev.n_lights % 64u == 0u;
  if (const char *e = std::getenv("PT_COHORT")) s->dev.cohort = s->dev.vertex_major == 1u && std::atoi(e) != 0;  // measuring knob
  if (s->dev.cohort) s->dev.refill_any = 64;
  if (const char *e = std::getenv("PT_QUORUM")) {  // tuning knob
    const int q = std::atoi(e);
    if (q >= 1 && q <= 64) s->dev.quorum = (uint32_t)q;
  }
  s->max_depth = f.max_depth; s->max_depth4 = f.max_depth4; s->max_depth8 = f.max_depth8; s->n_meshes = f.n_meshes;
  s->use_oct = s->dev.n_nodes8 != 0;
  if (const char *e = std::getenv("PT_NO_OCT")) s->use_oct = s->use_oct && std::atoi(e) == 0;  // tuning knob: stay on four children
  if (const char *e = std::getenv("PT_TAIL")) {  // tuning knob: 0 switches the tail kernel off
    const long v = std::atol(e);
    if (v >= 0 && v <= (1l << 30)) { s->tail_paths = (uint32_t)v; s->tail_given = true; }
  }
  int rc = make_plan(s, 0);
  if (rc == PT_OK) rc = make_plan(s, 1);
  if (rc == PT_OK && s->dev.n_nodes8 != 0) rc = make_plan(s, 2);
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
  if (!flat.nodes8.empty() && stack_class(flat.max_depth8) == 0) flat.nodes8.clear();  // too deep for its stack: the four-child form serves
  if (stack_class(flat.max_depth) == 0 || stack_class(flat.max_depth4) == 0) return fail(PT_ERR_UNSUPPORTED, "tree deeper than 128 levels");
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

int pt_scene_check(const PtSceneDesc *desc, PtSceneInfo *info) {
  if (!desc) return fail(PT_ERR_INVALID, "null argument");
  ptl::FlatScene flat;
  std::string err;
  int rc;
  try {
    rc = ptl::flatten_scene(*desc, flat, err);
    // a description whose boxes are not nested is never walked through the fast tree (effective_mode), whose
    // leaf boxes take those boxes at their word
    if (rc == PT_OK && flat.n_unnested == 0) rc = ptl::check_fast_tree(*desc, flat, err);
    if (rc == PT_OK && flat.n_unnested == 0) rc = ptl::check_oct_tree(*desc, flat, err);
  } catch (const std::bad_alloc &) {
    return fail(PT_ERR_NOMEM, "out of host memory while flattening the scene");
  }
  if (rc != PT_OK) return fail(rc, err);
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_meshes = flat.n_meshes;
    info->n_tris = (uint32_t)flat.tris.size() - 1;
    info->n_lights = (uint32_t)flat.lights.size();
    info->n_wide_nodes = (uint32_t)flat.nodes.size(); info->max_depth = flat.max_depth;
    info->n_fast_nodes = (uint32_t)flat.nodes4.size(); info->fast_max_depth = flat.max_depth4;
    info->n_gates = (uint32_t)flat.gates.size();
    info->boxes_nested = flat.n_unnested == 0 ? 1u : 0u;
    info->n_oct_nodes = (uint32_t)flat.nodes8.size();
  }
  return PT_OK;
}

int pt_scene_info(const PtScene *s, PtSceneInfo *info) {
  if (!s || !info) return fail(PT_ERR_INVALID, "null argument");
  info->n_meshes = s->n_meshes;
  info->n_tris = s->dev.n_tris - 1;  // without the degenerate filler
  info->n_lights = s->dev.n_lights;
  info->n_wide_nodes = s->dev.n_nodes; info->max_depth = s->max_depth; info->device = (uint32_t)s->device;
  info->n_fast_nodes = s->dev.n_nodes4; info->fast_max_depth = s->max_depth4; info->n_gates = s->dev.n_gates;
  info->device_bytes = s->bytes;
  info->boxes_nested = s->n_unnested == 0 ? 1u : 0u;
  info->n_oct_nodes = s->use_oct ? s->dev.n_nodes8 : 0u;
  return PT_OK;
}

// Switches the per-ray work counters (node boxes tested, triangles tested) on or off.
// Counting kernels are separate instantiations; the default kernels carry no counters.
int pt_scene_count_stats(PtScene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  s->count_stats = enable != 0;
  return PT_OK;
}

// `all_near`: the caller has looked at the origins (host arrays) and all of them are within near_limit(); the
// exact launch for the far ones is then left out.
static int trace_dev(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                     const float *dy, const float *dz, float *t, int32_t *tri, float *u, float *v, int mode,
                     hipStream_t st, bool all_near) {
  if (n == 0) return PT_OK;
  if (n > ((size_t)1 << 36)) return fail(PT_ERR_INVALID, "too many rays for one call");  // the work-queue head counts chunks of 64 in 32 bits
  if (!valid_mode(mode) && mode != PT_TRACE_FAST_OCT) return fail(PT_ERR_INVALID, "unknown trace mode");  // callers inside the library may have resolved it already
  mode = effective_mode(s, mode);
  int rc = enter_stream(s, st);
  if (rc != PT_OK) return rc;
  auto go = [&](int m, RaySide side) -> int {
    return with_mode(m, [&](auto MM) -> int {
      constexpr int M = decltype(MM)::value;
      return s->count_stats ? launch_trace_api<M, true>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st, side)
                            : launch_trace_api<M, false>(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, st, side);
    });
  };
  if (mode == PT_TRACE_EXACT || all_near) {
    rc = go(mode, RaySide{SIDE_ALL, 0.0f});
  } else {
    rc = go(mode, RaySide{SIDE_NEAR, near_limit(s)});
    if (rc == PT_OK) rc = go(PT_TRACE_EXACT, RaySide{SIDE_FAR, near_limit(s)});
  }
  if (rc != PT_OK) return rc;
  return leave_stream(s, st);
}

int pt_trace_closest_dev(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                         const float *dy, const float *dz, float *t, int32_t *tri, float *u, float *v, int mode,
                         void *stream) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !t || !tri)) return fail(PT_ERR_INVALID, "null ray or result array");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  return trace_dev(s, n, ox, oy, oz, dx, dy, dz, t, tri, u, v, mode, (hipStream_t)stream, false);
}

// Device staging for the host-pointer ray queries: kept with the scene and only ever grown, so that a caller
// that asks for a few rays at a time (Scene::castray is one ray per call in the reference) does not pay an
// allocation per call.
static int stage_buffer(PtScene *s, size_t floats, float **out) {
  if (s->stage_floats < floats) {
    if (s->d_stage) (void)hipFree(s->d_stage);
    s->d_stage = nullptr;
    s->stage_floats = 0;
    const size_t want = floats < 65536 ? 65536 : floats;
    HIP_TRY(hipMalloc(&s->d_stage, want * sizeof(float)));
    s->stage_floats = want;
  }
  *out = s->d_stage;
  return PT_OK;
}

static int read_stats4(PtScene *s, uint64_t out[4]) {
  DeviceGuard guard;
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
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  if (enable)
    for (int k = 0; k < 6; ++k)
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
  if (!valid_mode(mode)) return fail(PT_ERR_INVALID, "unknown trace mode");
  mode = effective_mode(s, mode);
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  float *buf = nullptr;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  int rc = stage_buffer(s, 10 * m, &buf);
  if (rc != PT_OK) return rc;
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    rc = trace_dev(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, buf + 6 * m,
                   reinterpret_cast<int32_t *>(buf + 7 * m), buf + 8 * m, buf + 9 * m, mode, 0,
                   host_origins_within(src[0], src[1], src[2], k, near_limit(s)));
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(t + off, buf + 6 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri + off, buf + 7 * m, k * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && u) e = hipMemcpy(u + off, buf + 8 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && v) e = hipMemcpy(v + off, buf + 9 * m, k * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  return rc;
}

int pt_trace_any(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                 const float *dy, const float *dz, uint8_t *hit, int mode) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !hit)) return fail(PT_ERR_INVALID, "null ray or result array");
  if (!valid_mode(mode)) return fail(PT_ERR_INVALID, "unknown trace mode");
  mode = effective_mode(s, mode);
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  float *buf = nullptr;
  int rc = stage_buffer(s, 7 * m, &buf);
  if (rc != PT_OK) return rc;
  rc = enter_stream(s, nullptr);
  if (rc != PT_OK) return rc;
  s->has_last = false;  // these calls end with hipDeviceSynchronize: nothing is left in flight
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    uint8_t *dh = reinterpret_cast<uint8_t *>(buf + 6 * m);
    auto go = [&](int md, RaySide side) -> int {
      return with_mode(md, [&](auto MM) -> int {
        return launch_any_api<decltype(MM)::value>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0, side);
      });
    };
    if (mode == PT_TRACE_EXACT || host_origins_within(src[0], src[1], src[2], k, near_limit(s))) {
      rc = go(mode, RaySide{SIDE_ALL, 0.0f});
    } else {  // the pruning walk for the origins in its range, the exact walk for the others
      rc = go(mode, RaySide{SIDE_NEAR, near_limit(s)});
      if (rc == PT_OK) rc = go(PT_TRACE_EXACT, RaySide{SIDE_FAR, near_limit(s)});
    }
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hit + off, dh, k, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  return rc;
}

int pt_trace_hits(PtScene *s, size_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                  const float *dy, const float *dz, int32_t *hits, int mode) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  if (n && (!ox || !oy || !oz || !dx || !dy || !dz || !hits)) return fail(PT_ERR_INVALID, "null ray or result array");
  if (!valid_mode(mode)) return fail(PT_ERR_INVALID, "unknown trace mode");
  mode = effective_mode(s, mode);
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  const size_t chunk = (size_t)1 << 24;
  const size_t m = n < chunk ? n : chunk;
  if (m == 0) return PT_OK;
  float *buf = nullptr;
  int rc = stage_buffer(s, 7 * m, &buf);
  if (rc != PT_OK) return rc;
  rc = enter_stream(s, nullptr);
  if (rc != PT_OK) return rc;
  s->has_last = false;  // these calls end with hipDeviceSynchronize: nothing is left in flight
  for (size_t off = 0; off < n && rc == PT_OK; off += chunk) {
    const size_t k = (n - off) < chunk ? (n - off) : chunk;
    const float *src[6] = {ox + off, oy + off, oz + off, dx + off, dy + off, dz + off};
    hipError_t e = hipSuccess;
    for (int a = 0; a < 6 && e == hipSuccess; ++a) e = hipMemcpy(buf + a * m, src[a], k * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = fail(PT_ERR_DEVICE, std::string("copy of rays: ") + hipGetErrorString(e)); break; }
    int32_t *dh = reinterpret_cast<int32_t *>(buf + 6 * m);
    // HitData::hits is a property of the reference's trees: both pruning modes walk those
    auto go = [&](bool exact, RaySide side) -> int {
      return exact ? launch_hits_api<PT_TRACE_EXACT>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0, side)
                   : launch_hits_api<PT_TRACE_FAST_BINARY>(s, k, buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, dh, 0, side);
    };
    if (mode == PT_TRACE_EXACT || s->dev.n_ill != 0) {
      rc = go(true, RaySide{SIDE_ALL, 0.0f});
    } else if (host_origins_within(src[0], src[1], src[2], k, near_limit(s))) {
      rc = go(false, RaySide{SIDE_ALL, 0.0f});
    } else {
      rc = go(false, RaySide{SIDE_NEAR, near_limit(s)});
      if (rc == PT_OK) rc = go(true, RaySide{SIDE_FAR, near_limit(s)});
    }
    if (rc != PT_OK) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hits + off, dh, k * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
  }
  return rc;
}

int pt_camera_rays(const PtCamera *cam, int W, int H, int device, float *dir_xyz) {
  if (!dir_xyz) return fail(PT_ERR_INVALID, "null output");
  FrameDev F;
  int rc = make_frame(cam, W, H, 0, nullptr, 0, F);
  if (rc != PT_OK) return rc;
  DeviceGuard guard;
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

// The frame loop shared by pt_render_dev and pt_render_dev_static: `batch` samples per pixel in flight
// together, in the workspace and counter block s->aw / s->acnt point at.
// `mean`: accum is the reference's Camera::pixels (a running mean of radiance * 255) instead of radiance sums.
static int render_batches(PtScene *s, FrameDev &F, const PtOptions &o, uint32_t spp_begin, uint32_t spp_count,
                          uint32_t batch, float *accum, PtCounters &c, hipStream_t st, bool mean) {
  int rc = PT_OK;
  Workspace &w = *s->aw;
  uint32_t *cnt = s->acnt;
  const bool cs = s->count_stats;
  const bool prof = s->profile;
  // longest camera ray: a corner of the viewport (the rotation keeps lengths), plus what the lens offset adds
  double dmax = std::sqrt(0.25 * (double)F.vw * F.vw + 0.25 * (double)F.neg_vh * F.neg_vh + (double)F.pin_z * F.pin_z);
  if (F.aperture > 0.0f) dmax += (double)F.aperture * std::fabs((double)F.pin_z) / (double)F.focus;
  const bool long_camera_rays = !(dmax * dmax <= 3.9);
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
    const PathFrame PF = path_frame(F);
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
      HIP_TRY(hipMemsetAsync(cnt, 0, 8 * sizeof(uint32_t), st));
      // bounce and visibility rays are normalised; camera rays are as long as the viewport makes them
      s->short_rays = depth > 0 || !long_camera_rays;
      if (n <= tail_limit(s, o.trace_mode)) {
        // few paths left: one kernel follows each of them to its end
        if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
        rc = with_mode(o.trace_mode, [&](auto MM) -> int {
          constexpr int M = decltype(MM)::value;
          return cs ? launch_tail<M, true>(s, PF, depth, n, w.q[cur], st) : launch_tail<M, false>(s, PF, depth, n, w.q[cur], st);
        });
        if (rc != PT_OK) return rc;
        if (prof) HIP_TRY(hipEventRecord(s->ev[1], st));
        HIP_TRY(hipMemcpyAsync(s->h_cnt, cnt, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (prof) {
          if ((rc = elapsed(0, 1, s->prof.tail_ms)) != PT_OK) return rc;
          s->prof.tail_launches += 1;
        }
        c.rays_closest += s->h_cnt[6];
        c.rays_shadow += (uint64_t)s->h_cnt[1] * s->dev.n_lights;
        c.bounce_passes += s->h_cnt[7];
        break;
      }
      if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
      rc = with_mode(o.trace_mode, [&](auto MM) -> int {
        constexpr int M = decltype(MM)::value;
        return cs ? launch_trace_queue<M, true>(s, PF, depth, n, w.q[cur], st) : launch_trace_queue<M, false>(s, PF, depth, n, w.q[cur], st);
      });
      if (rc != PT_OK) return rc;
      const uint32_t n_chunks = (n + 63) / 64, n_sblocks = (n_chunks + kScanSpan - 1) / kScanSpan;
      hipLaunchKernelGGL(k_scan_local, dim3(n_sblocks), dim3(kScanBlock), 0, st, n, n_chunks, w.alive, w.chunk_mask, w.chunk_base, w.block_tot);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, st, n_sblocks, w.block_tot, cnt);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(s->ev[1], st));
      rc = with_mode(o.trace_mode, [&](auto MM) -> int {
        constexpr int M = decltype(MM)::value;
        return cs ? launch_shade<M, true>(s, PF, depth, n, w.q[cur], w.q[cur ^ 1], st) : launch_shade<M, false>(s, PF, depth, n, w.q[cur], w.q[cur ^ 1], st);
      });
      if (rc != PT_OK) return rc;
      if (prof) HIP_TRY(hipEventRecord(s->ev[2], st));
      HIP_TRY(hipMemcpyAsync(s->h_cnt, cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (prof) {
        if ((rc = elapsed(0, 1, s->prof.trace_ms)) != PT_OK) return rc;
        if ((rc = elapsed(1, 2, s->prof.shade_ms)) != PT_OK) return rc;
        s->prof.trace_launches += 1;
        s->prof.shade_launches += 1;
        s->prof.trace_rays += n;
        if (plan_of(s, o.trace_mode).split) {
          if ((rc = elapsed(4, 5, s->prof.visibility_ms)) != PT_OK) return rc;
          s->prof.visibility_launches += 1;
          s->prof.visibility_rays += (uint64_t)s->h_cnt[1] * s->dev.n_lights;
        }
      }
      cur ^= 1;
      c.rays_closest += n;
      c.rays_shadow += (uint64_t)s->h_cnt[1] * s->dev.n_lights;
      c.bounce_passes += 1;
      n = s->h_cnt[0];
    }
    if (prof) HIP_TRY(hipEventRecord(s->ev[0], st));
    if (mean) hipLaunchKernelGGL(k_resolve_mean, dim3((F.P + kBlock - 1) / kBlock), dim3(kBlock), 0, st, F.P, sb, F.sample0, w.rad, accum);
    else hipLaunchKernelGGL(k_resolve, dim3((F.P + kBlock - 1) / kBlock), dim3(kBlock), 0, st, F.P, sb, w.rad, accum);
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
  return PT_OK;
}

// Workspace bytes per path in flight for this scene (both parts).
static uint64_t bytes_per_path(const PtScene *s, bool split) {
  return (uint64_t)(kWsArrays * sizeof(float)) + 2 + (split ? 12ull + 25ull * s->dev.n_lights : 0ull);
}

// Paths the static device buffer holds for this scene: 256 bytes of counters, then the workspace.
static const uint64_t kStaticRoom = PT_STATIC_SCRATCH_BYTES - 256 - 4096;  // slack for the alignment of the parts
static uint64_t static_capacity(const PtScene *s, bool split) {
  uint64_t cap = kStaticRoom / (bytes_per_path(s, split) + 1);
  cap &= ~63ull;
  // ws_*_bytes round up in places: shrink until the two parts fit
  while (cap >= 64 && ws_main_bytes(cap) + 256 + ws_shadow_bytes(cap, s->dev.n_lights, split) > kStaticRoom) cap -= 64;
  return cap;
}

static int render_dev_impl(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count,
                           uint64_t seed, const PtOptions *opt, float *accum, PtCounters *counters, hipStream_t st,
                           bool use_static, bool mean = false) {
  FrameDev F;
  int rc = make_frame(cam, W, H, seed, opt, s->dev.n_lights, F);
  if (rc != PT_OK) return rc;
  PtOptions o;
  if (opt) o = *opt; else pt_options_default(&o);
  if (mean && F.stride != 1) return fail(PT_ERR_INVALID, "the running mean is defined over consecutive passes: spp_stride must be 1");
  if (!valid_mode(o.trace_mode)) return fail(PT_ERR_INVALID, "unknown trace mode");
  o.trace_mode = effective_mode(s, o.trace_mode);
  if (o.trace_mode != PT_TRACE_EXACT && cam) {
    // the margins of the pruning walks cover ray origins up to 8 times the scene's largest coordinate;
    // every path of a frame starts at the camera (its later vertices lie on the scene)
    float far = 0;
    for (int k = 0; k < 3; ++k) far = std::fmax(far, std::fabs(cam->origin[k]));
    if (!(far <= near_limit(s))) o.trace_mode = PT_TRACE_EXACT;
  }
  const bool split = plan_of(s, o.trace_mode).split;
  PtCounters c;
  memset(&c, 0, sizeof(c));
  if (spp_count == 0) {
    if (counters) *counters = c;
    return PT_OK;
  }
  const uint64_t per_path = bytes_per_path(s, split);
  uint32_t batch = o.batch_spp;
  Workspace static_ws;
  if (use_static) {
    // everything comes out of the static device buffer: 256 bytes of counters, then the workspace
    unsigned char *base = static_scratch(s->device);
    if (!base) return fail(PT_ERR_DEVICE, "the static device buffer was not set up on this device");
    const uint64_t room = kStaticRoom;
    const uint64_t cap = static_capacity(s, split);
    if (cap < F.P) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "pt_render_dev_static: a %dx%d frame of this scene needs %llu bytes of workspace per sample "
                    "in flight, the static buffer holds %llu (at most %llu pixels)", W, H,
                    (unsigned long long)(per_path * F.P), (unsigned long long)room, (unsigned long long)cap);
      return fail(PT_ERR_UNSUPPORTED, msg);
    }
    const uint32_t fit = (uint32_t)(cap / F.P);
    if (batch == 0 || batch > fit) batch = fit;
    if (batch > spp_count) batch = spp_count;
    const size_t stride = ws_stride((size_t)batch * F.P);
    float *main_block = reinterpret_cast<float *>(base + 256);
    float *shadow_block = reinterpret_cast<float *>(base + 256 + ((ws_main_bytes(stride) + 255) & ~(size_t)255));
    rc = carve_workspace(static_ws, (size_t)batch * F.P, s->dev.n_lights, split, main_block, shadow_block);
    if (rc != PT_OK) return rc;
    s->aw = &static_ws;
    s->acnt = reinterpret_cast<uint32_t *>(base);
  } else {
    if (batch == 0) {
      // every batch ends in a long tail of nearly empty passes (half of the 44 passes of a 33-spp
      // batch of the 1080p room hold under 1 % of its rays), so fewer, larger batches are faster
      // (64 Mi -> 128 Mi paths in flight: +8 %); a quarter of the free device memory at most,
      // counting what the present workspace would give back
      uint64_t target = 192ull << 20;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t have = s->ws.cap ? ws_main_bytes(ws_stride(s->ws.cap)) + ws_shadow_bytes(ws_stride(s->ws.cap), s->dev.n_lights, s->ws.has_shadow) : 0;
        const uint64_t fit = ((uint64_t)free_b + have) / 4 / per_path;  // (a quarter of the free memory plus the workspace itself grew the batch with every call)
        if (fit < target) target = fit;
      }
      if (s->ws.cap > target) target = s->ws.cap;  // a workspace that exists is used in full
      batch = (uint32_t)(target / F.P);
    }
    if (batch > spp_count) batch = spp_count;
    if (batch < 1) batch = 1;
    while ((uint64_t)batch * F.P > (1ull << 31) && batch > 1) batch >>= 1;
    // a failed allocation is retried with half the batch
    for (;;) {
      rc = ensure_workspace(s, (size_t)batch * F.P, split);
      if (rc != PT_ERR_NOMEM || batch == 1) break;
      batch = (batch + 1) / 2;
    }
    if (rc != PT_OK) return rc;
    s->aw = &s->ws;
    s->acnt = s->d_cnt;
  }
  rc = enter_stream(s, st);
  if (rc == PT_OK && use_static) {
    // the previous user of this GPU's static buffer, whichever scene and stream it was (an event that was
    // never recorded counts as complete)
    const hipError_t e = hipStreamWaitEvent(st, g_scratch_last[s->device], 0);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
  }
  if (rc == PT_OK) rc = render_batches(s, F, o, spp_begin, spp_count, batch, accum, c, st, mean);
  if (rc == PT_OK && use_static) {
    const hipError_t e = hipEventRecord(g_scratch_last[s->device], st);
    if (e != hipSuccess) rc = fail(PT_ERR_DEVICE, std::string("hipEventRecord: ") + hipGetErrorString(e));
  }
  if (rc == PT_OK) rc = leave_stream(s, st);
  s->aw = &s->ws;
  s->acnt = s->d_cnt;
  s->short_rays = false;  // whatever render_batches left: the ray queries make no assumption about their rays
  if (rc == PT_OK && counters) *counters = c;
  return rc;
}

int pt_render_dev(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count,
                  uint64_t seed, const PtOptions *opt, float *accum, PtCounters *counters, void *stream) {
  if (!s || !accum) return fail(PT_ERR_INVALID, "null scene or accumulator");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  return render_dev_impl(s, cam, W, H, spp_begin, spp_count, seed, opt, accum, counters, (hipStream_t)stream, false);
}

int pt_render_mean_dev(PtScene *s, const PtCamera *cam, int W, int H, uint32_t pass_begin, uint32_t pass_count,
                       uint64_t seed, const PtOptions *opt, float *pixels, PtCounters *counters, void *stream) {
  if (!s || !pixels) return fail(PT_ERR_INVALID, "null scene or pixels");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  return render_dev_impl(s, cam, W, H, pass_begin, pass_count, seed, opt, pixels, counters, (hipStream_t)stream, false, true);
}

int pt_render_dev_static(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count,
                         uint64_t seed, const PtOptions *opt, float *accum, PtCounters *counters, void *stream) {
  if (!s || !accum) return fail(PT_ERR_INVALID, "null scene or accumulator");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->device));
  return render_dev_impl(s, cam, W, H, spp_begin, spp_count, seed, opt, accum, counters, (hipStream_t)stream, true);
}

uint64_t pt_static_scratch_bytes(void) { return PT_STATIC_SCRATCH_BYTES; }

uint64_t pt_static_max_pixels(const PtScene *s, int trace_mode) {
  if (!s || !valid_mode(trace_mode)) return 0;
  return static_capacity(s, plan_of(s, effective_mode(s, trace_mode)).split);
}

int pt_render(PtScene *s, const PtCamera *cam, int W, int H, uint32_t spp_begin, uint32_t spp_count, uint64_t seed,
              const PtOptions *opt, float *accum, PtCounters *counters) {
  if (!s || !accum) return fail(PT_ERR_INVALID, "null scene or accumulator");
  if (W <= 0 || H <= 0) return fail(PT_ERR_INVALID, "bad image size");
  DeviceGuard guard;
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

// ---------------------------------------------------------------- several devices, one all-reduce

// RCCL is loaded on first use (a one-device call never needs it) and by name only, so that a process
// that already holds an RCCL (PyTorch's) shares it.
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;
static std::vector<std::pair<std::vector<int>, std::vector<ncclComm_t>>> g_comms;  // per device list, kept for the process

static int rccl_load() {
  if (g_rccl.lib) return PT_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(PT_ERR_DEVICE, std::string("cannot load librccl: ") + dlerror());
  RcclApi a;
  a.lib = h;
  a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.CommInitAll || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString)
    return fail(PT_ERR_DEVICE, "librccl lacks an expected symbol");
  g_rccl = a;
  return PT_OK;
}

static int rccl_comms(const std::vector<int> &devices, std::vector<ncclComm_t> &out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  int rc = rccl_load();
  if (rc != PT_OK) return rc;
  for (auto &e : g_comms)
    if (e.first == devices) {
      out = e.second;
      return PT_OK;
    }
  std::vector<ncclComm_t> comms(devices.size());
  const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devices.size(), devices.data());
  if (r != ncclSuccess) return fail(PT_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
  g_comms.emplace_back(devices, comms);
  out = comms;
  return PT_OK;
}

int pt_shard_samples(uint32_t spp_begin, uint32_t spp_count, uint32_t spp_stride, uint32_t shard, uint32_t n_shards,
                     uint32_t *begin, uint32_t *count, uint32_t *stride) {
  if (n_shards == 0 || shard >= n_shards || !begin || !count || !stride) return fail(PT_ERR_INVALID, "bad shard or null output");
  const uint32_t st = spp_stride ? spp_stride : 1;
  *begin = spp_begin + shard * st;
  *count = spp_count > shard ? (spp_count - shard + n_shards - 1) / n_shards : 0;
  *stride = st * n_shards;
  return PT_OK;
}

int pt_render_multi(PtScene *const *scenes, int n, const PtCamera *cam, int W, int H, uint32_t spp_begin,
                    uint32_t spp_count, uint64_t seed, const PtOptions *opt, float *accum, PtCounters *counters) {
  if (!scenes || n <= 0 || !accum) return fail(PT_ERR_INVALID, "null scenes or accumulator");
  if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > (1ull << 28)) return fail(PT_ERR_INVALID, "bad image size");
  std::vector<int> devices;
  // PT_MULTI_SHARE_DEVICE=1 (a rehearsal switch for boxes with one GPU): several of the scenes may live on one
  // device.  Threads, sample dealing and the merge of the counters run as they do on several GPUs; the
  // accumulators are then added up by a kernel, because RCCL takes each device once per communicator.
  const char *share_env = std::getenv("PT_MULTI_SHARE_DEVICE");
  const bool may_share = share_env && std::atoi(share_env) != 0;
  bool shared = false;
  for (int g = 0; g < n; ++g) {
    if (!scenes[g]) return fail(PT_ERR_INVALID, "null scene");
    for (int h = 0; h < g; ++h)
      if (scenes[h] == scenes[g]) return fail(PT_ERR_INVALID, "the same scene object twice");
    for (int d : devices)
      if (d == scenes[g]->device) {
        if (!may_share) return fail(PT_ERR_INVALID, "two scenes on one device");
        shared = true;
      }
    devices.push_back(scenes[g]->device);
  }
  // PT_MULTI_FORCE_RCCL=1 (the other rehearsal switch): a single scene takes the general path too, so that a box
  // with one GPU executes the library loading, the communicator and the grouped all-reduce (over one rank).
  const char *force_env = std::getenv("PT_MULTI_FORCE_RCCL");
  const bool force_rccl = force_env && std::atoi(force_env) != 0;
  if (n == 1 && !force_rccl) return pt_render(scenes[0], cam, W, H, spp_begin, spp_count, seed, opt, accum, counters);
  PtOptions o;
  if (opt) o = *opt; else pt_options_default(&o);
  const uint32_t stride = o.spp_stride ? o.spp_stride : 1;
  DeviceGuard guard;
  const size_t count = (size_t)W * H * 3, bytes = count * sizeof(float);
  std::vector<float *> bufs((size_t)n, nullptr);
  std::vector<int> rcs((size_t)n, PT_OK);
  std::vector<std::string> errs((size_t)n);
  std::vector<PtCounters> cs((size_t)n);
  auto cleanup = [&]() {
    for (int g = 0; g < n; ++g)
      if (bufs[g]) {
        (void)hipSetDevice(devices[g]);
        (void)hipFree(bufs[g]);
      }
  };
  // one host thread per device: its share of the samples into a zeroed accumulator on that device
  auto work = [&](int g) {
    memset(&cs[g], 0, sizeof(PtCounters));
    hipError_t e = hipSetDevice(devices[g]);
    if (e == hipSuccess) e = hipMalloc(&bufs[g], bytes);
    if (e == hipSuccess) e = hipMemset(bufs[g], 0, bytes);
    if (e != hipSuccess) {
      rcs[g] = PT_ERR_DEVICE;
      errs[g] = std::string("device accumulator: ") + hipGetErrorString(e);
      return;
    }
    PtOptions og = o;
    uint32_t first = 0, mine = 0;
    (void)pt_shard_samples(spp_begin, spp_count, stride, (uint32_t)g, (uint32_t)n, &first, &mine, &og.spp_stride);
    rcs[g] = pt_render_dev(scenes[g], cam, W, H, first, mine, seed, &og, bufs[g], &cs[g], nullptr);
    if (rcs[g] != PT_OK) errs[g] = g_err;
    else if ((e = hipDeviceSynchronize()) != hipSuccess) {
      rcs[g] = PT_ERR_DEVICE;
      errs[g] = std::string("render: ") + hipGetErrorString(e);
    }
  };
  {
    std::vector<std::thread> th;
    for (int g = 1; g < n; ++g) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
  }
  for (int g = 0; g < n; ++g)
    if (rcs[g] != PT_OK) {
      cleanup();
      return fail(rcs[g], "device " + std::to_string(devices[g]) + ": " + errs[g]);
    }
  if (shared) {
    for (int g = 1; g < n; ++g) {
      if (devices[g] != devices[0]) { cleanup(); return fail(PT_ERR_INVALID, "PT_MULTI_SHARE_DEVICE: all scenes must be on one device"); }
    }
    (void)hipSetDevice(devices[0]);
    for (int g = 1; g < n; ++g)
      hipLaunchKernelGGL(k_add_into, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, count, bufs[0], bufs[g]);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { cleanup(); return fail(PT_ERR_DEVICE, std::string("merge: ") + hipGetErrorString(e)); }
  } else {
    std::vector<ncclComm_t> comms;
    int rc = rccl_comms(devices, comms);
    if (rc != PT_OK) { cleanup(); return rc; }
    ncclResult_t r = g_rccl.GroupStart();
    for (int g = 0; g < n && r == ncclSuccess; ++g) {
      (void)hipSetDevice(devices[g]);
      r = g_rccl.AllReduce(bufs[g], bufs[g], count, ncclFloat, ncclSum, comms[g], nullptr);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) { cleanup(); return fail(PT_ERR_DEVICE, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r)); }
    for (int g = 0; g < n; ++g) {
      (void)hipSetDevice(devices[g]);
      const hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { cleanup(); return fail(PT_ERR_DEVICE, std::string("all-reduce: ") + hipGetErrorString(e)); }
    }
  }
  // every device now holds the frame's sums; the first one's copy goes to the caller
  std::vector<float> host(count);
  (void)hipSetDevice(devices[0]);
  const hipError_t e = hipMemcpy(host.data(), bufs[0], bytes, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("copy of the result: ") + hipGetErrorString(e));
  for (size_t i = 0; i < count; ++i) accum[i] += host[i];
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (int g = 0; g < n; ++g) {
      counters->paths += cs[g].paths; counters->rays_closest += cs[g].rays_closest; counters->rays_shadow += cs[g].rays_shadow;
      counters->node_boxes += cs[g].node_boxes; counters->tri_tests += cs[g].tri_tests;
      counters->shadow_node_boxes += cs[g].shadow_node_boxes; counters->shadow_tri_tests += cs[g].shadow_tri_tests;
      if (cs[g].bounce_passes > counters->bounce_passes) counters->bounce_passes = cs[g].bounce_passes;
    }
  }
  return PT_OK;
}

int pt_resolve_dev(const float *sum_dev, size_t n, uint32_t spp, float *pixels_dev, int device, void *stream) {
  if (!sum_dev || !pixels_dev || spp == 0) return fail(PT_ERR_INVALID, "null array or spp == 0");
  if (n == 0) return PT_OK;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_resolve_pixels, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, (float)spp,
                     sum_dev, pixels_dev);
  HIP_TRY(hipGetLastError());
  return PT_OK;
}

int pt_resolve_display_dev(const float *sum_dev, size_t n_pixels, uint32_t spp, uint8_t *rgb8_dev, int device, void *stream) {
  if (!sum_dev || !rgb8_dev || spp == 0) return fail(PT_ERR_INVALID, "null array or spp == 0");
  if (n_pixels == 0) return PT_OK;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  const size_t n = 3 * n_pixels;
  hipLaunchKernelGGL(k_resolve_display<true>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, (float)spp,
                     sum_dev, rgb8_dev);
  HIP_TRY(hipGetLastError());
  return PT_OK;
}

int pt_display_dev(const float *pixels_dev, size_t n_pixels, uint8_t *rgb8_dev, int device, void *stream) {
  if (!pixels_dev || !rgb8_dev) return fail(PT_ERR_INVALID, "null array");
  if (n_pixels == 0) return PT_OK;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  const size_t n = 3 * n_pixels;
  hipLaunchKernelGGL(k_resolve_display<false>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, 1.0f,
                     pixels_dev, rgb8_dev);
  HIP_TRY(hipGetLastError());
  return PT_OK;
}

int pt_device_alloc(int device, size_t bytes, void **out) {
  if (!out) return fail(PT_ERR_INVALID, "null output");
  *out = nullptr;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
  HIP_TRY(hipMemset(*out, 0, bytes ? bytes : 16));
  return PT_OK;
}

int pt_device_free(int device, void *ptr) {
  if (!ptr) return PT_OK;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(ptr));
  return PT_OK;
}

int pt_device_clear(int device, void *ptr, size_t bytes, void *stream) {
  if (!ptr) return fail(PT_ERR_INVALID, "null pointer");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream));
  return PT_OK;
}

int pt_device_read(int device, void *dst_host, const void *src_dev, size_t bytes, void *stream) {
  if (!dst_host || !src_dev) return fail(PT_ERR_INVALID, "null pointer");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return PT_OK;
}

int pt_resolve(const float *sum, size_t n, uint32_t spp, float *pixels) {
  if (!sum || !pixels || spp == 0) return fail(PT_ERR_INVALID, "null array or spp == 0");
  for (size_t i = 0; i < n; ++i) pixels[i] = sum[i] / (float)spp * 255.0f;
  return PT_OK;
}

}  // extern "C"

#ifdef PT_DEBUG_ROUNDS
extern "C" int pt_debug_rounds(unsigned long long *out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_dbg), 24 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned long long z[24] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_dbg), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

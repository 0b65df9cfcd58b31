// build: hipcc -O3 --offload-arch=gfx950 gather_chain.hip -o gather_chain
// Dependent random gathers, the memory pattern of a tree walk outside LDS: every lane follows its own chain of
// records through a table; a step loads PIECES 16-byte pieces of one record of REC bytes and derives the next
// record from what it read.  Prints records per second and bytes per second (records x REC) for tables of
// several sizes, 128-byte records read as 7 pieces (the four-child node) against 64-byte records read as 4,
// at 1, 2, 4 and 8 waves per SIMD.  What it answers: is a walk at N G records/s bound by the latency of its
// dependent fetches (then more waves help and smaller records do not) or by the memory system's rate for
// scattered lines (then the reverse).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int REC, int PIECES>
__global__ void __launch_bounds__(256) chain(const uint4 *table, uint32_t n_rec, int steps, uint32_t *out) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    const uint32_t r = (uint32_t)(((uint64_t)x * n_rec) >> 32);
    const uint4 *p = table + (size_t)r * (REC / 16);
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const uint4 v = p[k];
      h += v.x ^ v.y ^ v.z ^ v.w;
    }
    acc += h;
    x = x * 1664525u + 1013904223u + h;  // the next record depends on this one's content
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int REC, int PIECES>
static void run(size_t table_bytes, int waves_per_simd) {
  const uint32_t n_rec = (uint32_t)(table_bytes / REC);
  uint4 *table;
  uint32_t *out;
  (void)hipMalloc(&table, table_bytes);
  std::vector<uint32_t> init(table_bytes / 4);
  uint32_t s = 1;
  for (auto &w : init) { s = s * 1664525u + 1013904223u; w = s >> 3; }
  (void)hipMemcpy(table, init.data(), table_bytes, hipMemcpyHostToDevice);
  const int blocks = 256 * waves_per_simd, steps = 2000;  // 256 threads = one wave per SIMD of a CU
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  chain<REC, PIECES><<<blocks, 256>>>(table, n_rec, 50, out);
  (void)hipEventRecord(e0);
  chain<REC, PIECES><<<blocks, 256>>>(table, n_rec, steps, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double recs = (double)blocks * 256 * steps;
  printf("  table %6.0f MB  record %3d B / %d pieces  %d waves/SIMD: %6.2f G records/s  %5.2f TB/s  (%.2f us per dependent step)\n",
         table_bytes / 1e6, REC, PIECES, waves_per_simd, recs / ms / 1e6, recs * REC / ms / 1e9, ms * 1e3 / steps);
  (void)hipFree(table); (void)hipFree(out);
}

int main(int argc, char **argv) {
  if (argc == 4) {  // one case only (for counter passes): table MB, record bytes (128 or 64), waves per SIMD
    const size_t mb = (size_t)atol(argv[1]);
    const int w = atoi(argv[3]);
    if (atoi(argv[2]) == 128) run<128, 7>(mb << 20, w); else run<64, 4>(mb << 20, w);
    return 0;
  }
  for (size_t mb : {8, 24, 200, 600}) {
    for (int w : {1, 2, 4, 8}) {
      run<128, 7>(mb << 20, w);
      run<64, 4>((mb << 20) / 2, w);  // the same number of records at half the size
    }
  }
  return 0;
}

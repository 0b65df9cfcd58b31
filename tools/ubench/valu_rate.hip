// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
// Issue-rate microbenchmark: scalar-per-lane v_mul_f32 / v_fma_f32 against packed v_pk_mul_f32 /
// v_pk_fma_f32 on gfx950.  Each wave runs 8 independent chains.  Prints, for 1, 2, 4 and 8 waves per SIMD,
// the SIMD time per wave instruction two ways: from the wall clock at a nominal 2.4 GHz, and in shader
// cycles counted by the waves themselves (s_memtime around the loop, median wave), which does not depend
// on the clock the chip holds under load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed, unsigned long long *cyc) {
  float a[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i + threadIdx.x}; }
  const float m = 1.0000001f;
  const f2 m2 = {m, m};
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q[2] = {}; __shared__ f4 lds[256]; lds[threadIdx.x] = f4{seed,seed,seed,seed}; __syncthreads(); const unsigned lds_addr = threadIdx.x * 16;
  asm volatile("v_cmp_lt_u32 vcc, 7, %0" : : "v"(threadIdx.x) : "vcc");
  unsigned long long mask = __ballot(threadIdx.x & 1);
  double dd[8]; const double dm = 1.0000001;
  for (int i = 0; i < 8; ++i) dd[i] = seed + i;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
      if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(m2));
      if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
      if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mask));
      if (KIND == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 9) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 13) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(mask) : "v"(a[i]), "v"(m));
      if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      if (KIND == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 16) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dm));
      if (KIND == 17) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[i]) : "v"(dm));
      if (KIND == 18) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[i]) : "v"(a[i]));
      if (KIND == 19) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 20) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 21) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 25) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
      if (KIND == 26) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 27) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(seed));
      if (KIND == 28) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(m), "v"(seed), "s"(mask));
      if (KIND == 29) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 30) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %1" : "+v"(a[i]) : "v"(m) : "vcc");
      if (KIND == 31) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 32) asm volatile("v_div_scale_f32 %0, %2, %0, %1, %1" : "+v"(a[i]) : "v"(m), "s"(mask));
      if (KIND == 33) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 1]) : "v"(lds_addr));
      if (KIND == 34) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 35) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 23) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(dd[i]) : "v"(m) : "vcc");
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)dd[i];
  s += (float)mask + q[0].x + q[1].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#include <algorithm>
#include <vector>
template <int KIND>
void run(const char *name) {
  printf("%-18s", name);
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int iters = 20000, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    float *d;
    unsigned long long *cyc;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 100, 1.0f, cyc);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 1.0f, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double inst_per_simd = (double)iters * 8 * waves_per_simd;
    printf("  w%d: %5.2f wall / %5.2f shader", waves_per_simd, ms * 1e-3 * 2.4e9 / inst_per_simd,
           (double)h[h.size() / 2] / inst_per_simd);
    hipFree(d); hipFree(cyc);
  }
  printf("\n");
}

int main() {
  printf("SIMD cycles per wave64 instruction (w = waves per SIMD; wall = kernel time x 2.4 GHz, shader = s_memtime)\n");
  run<2>("v_fma_f32"); run<0>("v_mul_f32"); run<10>("v_sub_f32"); run<1>("v_pk_mul_f32"); run<4>("v_pk_add_f32"); run<3>("v_pk_fma_f32");
  run<26>("v_cndmask e32 vcc"); run<28>("v_cndmask e64 sgpr"); run<9>("v_min_f32"); run<6>("v_max3_f32"); run<34>("v_cmp_lt_f32 vcc");
  run<13>("v_cmp_lt_f32 sgpr"); run<14>("v_mov_b32"); run<15>("v_rcp_f32"); run<29>("v_div_fmas_f32"); run<30>("v_div_scale_f32"); run<31>("v_div_fixup_f32");
  run<16>("v_mul_f64"); run<17>("v_fma_f64"); run<18>("v_cvt_f64_f32"); run<20>("v_mul_lo_u32"); run<21>("v_mul_hi_u32"); run<23>("v_mad_u64_u32");
  run<33>("ds_read_b128");
  return 0;
}

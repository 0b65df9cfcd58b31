// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
// Issue-rate microbenchmark: scalar-per-lane v_mul_f32 / v_fma_f32 against packed v_pk_mul_f32 /
// v_pk_fma_f32 on gfx950.  Each wave runs 8 independent chains; prints cycles per wave instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
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
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)dd[i];
  s += (float)mask + q[0].x + q[1].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int waves_per_simd) {
  const int iters = 20000, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
  float *d;
  hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(d, 100, 1.0f);
  hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = (double)iters * 8 * waves_per_simd;
  printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction at 2.4 GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / inst_per_simd);
  hipFree(d);
}

int main() {
  for (int w : {4}) {
    run<29>("v_div_fmas_f32", w); run<30>("v_div_scale vcc", w); run<31>("v_div_fixup_f32", w); run<33>("ds_read_b128", w); run<34>("v_cmp e32 vcc", w);
    run<24>("cndmask e64 vcc", w); run<25>("v_addc_co_u32", w); run<26>("cndmask e32 vcc", w); run<27>("cndmask e32 nodep", w); run<28>("cndmask e64 nodep", w);
    run<0>("v_mul_f32", w); run<1>("v_pk_mul_f32", w); run<2>("v_fma_f32", w); run<3>("v_pk_fma_f32", w);
    run<4>("v_pk_add_f32", w); run<5>("v_cndmask vcc", w); run<6>("v_max3_f32", w); run<7>("v_cndmask sgpr", w);
    run<8>("v_cmp->vcc", w); run<9>("v_min_f32", w); run<10>("v_sub_f32", w); run<11>("v_and_b32", w);
    run<12>("v_add_u32", w); run<13>("v_cmp->sgpr", w); run<14>("v_mov_b32", w); run<15>("v_rcp_f32", w);
    run<16>("v_mul_f64", w); run<17>("v_fma_f64", w); run<18>("v_cvt_f64_f32", w); run<19>("v_min3_f32", w);
    run<20>("v_mul_lo_u32", w); run<21>("v_mul_hi_u32", w); run<22>("v_xor_b32", w); run<23>("v_mad_u64_u32", w);
  }
  return 0;
}

// issue rate of the VALU instructions the epilogues are made of (independent chains, W waves per SIMD, wall clock)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(1024) void k(int iters, float *sink) {
  const int t = threadIdx.x;
  float x[8], y = 1.0001f, z = 0.5f;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = t + i; u[i] = t * 3 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#define ONE(i)                                                                                                         \
  if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y), "v"(z));                                 \
  if (OP == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(u[i]) : "v"(x[i]), "v"(y), "v"(z));                \
  if (OP == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(u[i]) : "v"(x[i]), "v"(y), "v"(z));                \
  if (OP == 3) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                             \
  if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(y));                           \
  if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                                \
  if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                        \
  if (OP == 7) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));    \
  if (OP == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));                                             \
  if (OP == 10) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(y), "v"(z));                \
  if (OP == 11) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(y), "v"(z)); \
  if (OP == 12) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(x[i]) : "v"(u[i]));                                     \
  if (OP == 13) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[i]) : "v"(u[i])); \
  if (OP == 14) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(y), "v"(z)); \
  if (OP == 15) asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(u[i]) : "v"(x[i]));                                     \
  if (OP == 16) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));      \
  if (OP == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double *)&x[i & 6]) : "v"(*(double *)&x[(i + 2) & 6]), "v"(*(double *)&x[(i + 4) & 6]));
      OPS(ONE)
#undef ONE
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + u[i];
  sink[blockIdx.x * 1024 + t] = s;
}
template <int OP>
void run(const char *name, int wps) {
  float *sink; (void)hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 4000;
  float best = 1e9f;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wps), 0, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-18s waves/SIMD %d : %.2f ns per instruction per SIMD\n", name, wps, best * 1e6 / (iters * 32.0 * wps));
  (void)hipFree(sink);
}
int main() {
  for (int wps = 1; wps <= 2; ++wps) {
    run<0>("v_fma_f32", wps); run<1>("v_fma_mixlo_f16", wps); run<2>("v_fma_mixhi_f16", wps); run<3>("v_pk_max_f16", wps);
    run<4>("v_cvt_pk_f16_f32", wps); run<5>("v_xor_b32", wps); run<6>("v_lshl_add_u32", wps); run<7>("v_pk_fma_f16", wps);
    run<8>("v_max_f32", wps); run<9>("v_pk_fma_f32", wps);
    run<10>("v_fma_mix_f32 lo", wps); run<11>("v_fma_mix_f32 hi", wps); run<12>("v_cvt_f32_f16", wps); run<13>("v_cvt_f32_f16 sdwa", wps);
    run<14>("mixlo f16-in", wps); run<15>("v_cvt_f16_f32", wps); run<16>("v_pack_b32_f16", wps);
  }
  return 0;
}

// How many single-issue VALU instructions hide behind one MFMA, by MFMA shape and waves per SIMD:
// every wave runs {MFMA, F x v_fma_f32} x 8 per iteration (independent accumulators / filler registers, order pinned by asm volatile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int F>
__global__ __launch_bounds__(1024) void k(int iters, float *sink, long long *clk) {
  const int t = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (t + j)); b[j] = (_Float16)(0.002f * (t - j)); }
  f32x4 c4[8]; f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
  float x[8], y = 1.0001f, z = 0.5f;
  for (int i = 0; i < 8; ++i) x[i] = t + i;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[m]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[m & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(f + m) & 7]) : "v"(y), "v"(z));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c4[i][0] + x[i];
  for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][15];
  sink[blockIdx.x * 1024 + t] = s;
  if (t == 0) clk[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int F>
void run(int wps) {
  float *sink; long long *clk;
  hipMalloc(&sink, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  const int iters = 4000;
  long long best = 1LL << 60;
  float bestms = 1e9f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, F>), dim3(256), dim3(256 * wps), 0, 0, iters, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < bestms) bestms = ms;
    long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    if (c[0] < best) best = c[0];
  }
  // cycles per MFMA per SIMD (all waves of the SIMD together issue wps MFMAs in that time)
  const double nmfma_simd = (double)iters * 8 * wps;            // MFMAs per SIMD
  const double flop = nmfma_simd * 1024 * 32768.0 * (SHAPE == 16 ? 0.5 : 1.0);
  printf("shape %2d  waves/SIMD %d  fillers/MFMA %d : %.1f ticks per MFMA per wave, %.1f ns per SIMD-MFMA, %.0f TFLOP/s, %.2f ticks/ns\n", SHAPE, wps, F,
         (double)best / iters / 8, bestms * 1e6 / nmfma_simd, flop / bestms / 1e9, best / (bestms * 1e6));
  hipFree(sink); hipFree(clk);
}
template <int SHAPE> void sweep(int wps) {
  run<SHAPE, 0>(wps); run<SHAPE, 1>(wps); run<SHAPE, 2>(wps); run<SHAPE, 3>(wps); run<SHAPE, 4>(wps); run<SHAPE, 6>(wps); run<SHAPE, 8>(wps);
}
int main() {
  for (int wps = 1; wps <= 3; ++wps) { sweep<16>(wps); sweep<32>(wps); }
  return 0;
}

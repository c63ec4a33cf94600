// What a ds_read_b128 costs beside MFMAs: every wave runs {R reads for the next step, 8 MFMAs on the operands read one step
// earlier} per step (software-pipelined by hand, two register sets), W waves per SIMD; also reads alone and MFMAs alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, bool MFMA>
__global__ __launch_bounds__(1024) void k(int iters, float *sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  for (int i = t; i < 16384; i += blockDim.x) ((float *)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f16x8 a;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (t + j));
  f16x8 b0[8], b1[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { b0[i][j] = (_Float16)(0.002f * (i + j)); b1[i][j] = b0[i][j]; }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
  const unsigned addr = (unsigned)(size_t)smem + lane * 16 + (wid & 3) * 8192;   // 1 KiB contiguous per wave-instruction
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1[i]) : "v"(addr), "n"(i * 1024));
    if (MFMA) {
#pragma unroll
      for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b0[m]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0[i]) : "v"(addr), "n"(i * 1024));
    if (MFMA) {
#pragma unroll
      for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b1[m]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + (float)b0[i][0] + (float)b1[i][1];
  sink[blockIdx.x * 1024 + t] = s;
}

template <int R, bool MFMA>
void run(int wps) {
  float *sink; (void)hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 2000;
  float best = 1e9f;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, MFMA>), dim3(256), dim3(256 * wps), 0, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double steps = 2.0 * iters;                       // steps per wave
  const double ns_step = best * 1e6 / steps;              // wall per step (all waves of a SIMD run concurrently)
  const double lds_bytes = steps * R * 1024.0 * 4 * wps;  // per CU
  printf("waves/SIMD %d  reads/step %d  mfma %d : %7.1f ns per step per wave-slot", wps, R, MFMA ? 8 : 0, ns_step);
  if (MFMA) printf("  = %5.2f ns per SIMD-MFMA", ns_step / (8.0 * wps));
  if (R) printf("  LDS %6.1f B/ns/CU", lds_bytes / (best * 1e6));
  printf("\n");
  (void)hipFree(sink);
}
int main() {
  for (int wps = 1; wps <= 3; ++wps) {
    run<0, true>(wps); run<2, true>(wps); run<4, true>(wps); run<8, true>(wps);
    run<2, false>(wps); run<4, false>(wps); run<8, false>(wps);
  }
  return 0;
}

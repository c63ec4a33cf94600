// What the matrix pipe really delivers on this box: N waves/CU of back-to-back v_mfma_f32_16x16x32_f16
// (registers only), optionally with ds_read_b128 / VALU mixed in at the K-loop's ratios.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float *sink, long long *clk) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int t = threadIdx.x;
  f16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (t + i + j));
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.002f * (t + i - j));
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  for (int i = t; i < 16384; i += 512) ((float *)smem)[i] = i;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {   // 16 ds_read_b128 per 32 MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *(const f16x8 *)(smem + ((t * 16 + i * 8192 + it * 16) & 65520));
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *(const f16x8 *)(smem + ((t * 16 + i * 4096 + it * 32) & 65520));
    }
    if (MODE >= 2) {   // ~48 VALU
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        b[i] = __builtin_elementwise_max(b[i] * a[i] + a[i + 4], z);
        b[i] = __builtin_elementwise_max(b[i] * a[i + 1] + a[i + 3], z);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  sink[blockIdx.x * 512 + t] = s;
  if (t == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int iters) {
  float *sink; long long *clk;
  hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double fl = 256.0 * 8 * iters * 32 * 16 * 16 * 32 * 2;
    printf("%-22s iters %d: %.3f ms  %.0f TFLOP/s   loop ticks %lld -> %.2f ticks/ns, %.1f ticks per 32-MFMA step per wave\n", name, iters, ms,
           fl / ms / 1e9, c[0], c[0] / (ms * 1e6), (double)c[0] / iters);
  }
}
int main(int argc, char **argv) {
  if (argc > 2 && !strcmp(argv[1], "loop")) {      // mfmabench loop <n>: n long MFMA-only launches back to back (scripts/power_trace.py)
    for (int i = 0; i < atoi(argv[2]); ++i) run<0>("mfma only (long)", 400000);
    return 0;
  }
  run<0>("mfma only", 20000);
  run<1>("mfma + 16 ds_read", 20000);
  run<2>("mfma + ds_read + valu", 20000);
  run<0>("mfma only (long)", 200000);
  return 0;
}

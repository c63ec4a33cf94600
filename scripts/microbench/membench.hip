// Per-CU load-path microbenchmark: how fast can one CU pull data global->LDS / ->VGPR
// as a function of the access shape of one wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef float f4 __attribute__((ext_vector_type(4)));

// each wave-instruction covers RPI rows x (1024/RPI) bytes; rows are `stride` bytes apart.
template <int RPI, bool DMA, int NINF>
__global__ __launch_bounds__(512) void k(const char *base, long region, long stride, int iters, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nw = blockDim.x >> 6;
  constexpr int LPR = 64 / RPI;                 // lanes per row
  const long lane_off = (long)(lane / LPR) * stride + (lane % LPR) * 16;
  // this workgroup's private region
  const char *p = base + (long)blockIdx.x * region;
  f4 acc = {0, 0, 0, 0};
  long off = (long)wid * RPI * stride;          // waves interleave over row groups
  const long step = (long)nw * RPI * stride;
  for (int it = 0; it < iters; ++it) {
    if constexpr (DMA) {
#pragma unroll
      for (int j = 0; j < NINF; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)(p + off + lane_off), (lptr_t)(smem + (wid * NINF + j) * 1024), 16, 0, 0);
        off += step; if (off >= region) off -= region;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f4 v[NINF];
#pragma unroll
      for (int j = 0; j < NINF; ++j) {
        v[j] = *(const f4 *)(p + off + lane_off);
        off += step; if (off >= region) off -= region;
      }
#pragma unroll
      for (int j = 0; j < NINF; ++j) acc += v[j];
    }
  }
  if (acc[0] == 123.456f) sink[t] = acc[1];
}

template <int RPI, bool DMA, int NINF>
void run(const char *d, long total, long stride, int nwg, int threads, const char *tag) {
  long region = (total / nwg) & ~((long)stride * 64 * 8 - 1);
  if (region < stride * 64) region = stride * 64;
  const int iters = 200;
  float *sink; hipMalloc(&sink, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<RPI, DMA, NINF>), dim3(nwg), dim3(threads), 64 * 1024, 0, d, region, stride, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)nwg * (threads / 64) * iters * NINF * 1024.0;
  printf("%-34s wg=%4d thr=%4d rows/instr=%2d stride=%6ld inflight=%2d : %7.2f TB/s  %6.1f B/clk/CU(@2.1GHz, %d CUs)\n", tag, nwg, threads, RPI,
         stride, NINF, bytes / ms / 1e9, bytes / ms / 1e6 / 2.1e3 / (nwg < 256 ? nwg : 256), nwg < 256 ? nwg : 256);
  hipFree(sink);
}

int main() {
  const long total = 2L << 30;
  char *d; hipMalloc(&d, total + (64L << 20)); hipMemset(d, 1, total + (64L << 20));
  hipFuncSetAttribute((const void *)k<1, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (long tot : {2L << 30, 64L << 20}) {
    printf("---- footprint %ld MB\n", tot >> 20);
    for (int nwg : {256, 512}) for (int thr : {256, 512}) {
      run<1, true, 8>(d, tot, 1024, nwg, thr, "DMA contiguous 1KiB");
      run<8, true, 8>(d, tot, 2048, nwg, thr, "DMA 8 rows x128B");
      run<16, true, 8>(d, tot, 2048, nwg, thr, "DMA 16 rows x64B");
      run<1, false, 8>(d, tot, 1024, nwg, thr, "LOAD contiguous 1KiB");
      run<8, false, 8>(d, tot, 2048, nwg, thr, "LOAD 8 rows x128B");
      run<16, false, 8>(d, tot, 2048, nwg, thr, "LOAD 16 rows x64B");
    }
    run<8, true, 4>(d, tot, 2048, 256, 512, "DMA 8 rows x128B");
    run<8, true, 16>(d, tot, 2048, 256, 512, "DMA 8 rows x128B");
    run<8, false, 16>(d, tot, 2048, 256, 512, "LOAD 8 rows x128B");
    run<8, true, 8>(d, tot, 512, 256, 512, "DMA 8 rows x128B");
    run<8, true, 8>(d, tot, 2176, 256, 512, "DMA 8 rows x128B");
    run<8, true, 8>(d, tot, 2048, 16, 512, "DMA 8 rows x128B (16 CUs)");
    run<8, false, 8>(d, tot, 2048, 16, 512, "LOAD 8 rows x128B (16 CUs)");
  }
  return 0;
}

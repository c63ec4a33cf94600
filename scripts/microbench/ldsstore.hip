// LDS store throughput at the occupancy of the fused dense-layer kernel (2 waves per SIMD, one 512-thread workgroup per CU):
// ds_write_b64 against ds_write_b128, 1 KiB / 512 B contiguous per wave-instruction, and a 256-byte-pitch pattern like epilogue A's
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  unsigned d[4] = {(unsigned)t, (unsigned)t * 3, (unsigned)t * 5, (unsigned)t * 7};
  unsigned base = (unsigned)(size_t)smem + wid * 4096;
  // MODE 0: b64 contiguous (lane*8); 1: b128 contiguous (lane*16); 2: b64, pixel pitch 256 B: lane (frow = l&15, fch = l>>4) -> frow*256 + fch*8
  // 3: b128, pixel pitch 256 B: frow*256 + fch*16
  unsigned addr = base + (MODE == 0 ? lane * 8 : MODE == 1 ? lane * 16 : MODE == 2 ? (lane & 15) * 256 + (lane >> 4) * 8 : (lane & 15) * 256 + (lane >> 4) * 16);
  if (MODE >= 4) {
    // epilogue A of the dense-layer kernel: lane (frow = l & 15, fch = l >> 4) stores 8 B of pixel slot s0 + 16 mi + frow, chunk 2 ni + (fch >> 1),
    // swizzled (MODE 4: chunk ^ ((slot & 7) << 1); MODE 5: chunk ^ (slot & 15), the earlier conflict-free-for-stores form)
    const int frow = lane & 15, fch = lane >> 4;
    unsigned a4[4];
    for (int mi = 0; mi < 4; ++mi) {
      const int slot = (wid & 7) * 64 + mi * 16 + frow;     // 512 slots x 256 B = the 128 KiB tile (two waves per SIMD share slots: same addresses, fine)
      a4[mi] = (unsigned)(size_t)smem + slot * 256 + (fch & 1) * 8;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int slot = (wid & 7) * 64 + mi * 16 + frow;
          const int chunk = 2 * ni + (fch >> 1);
          const unsigned ad = a4[mi] + (((MODE == 4 ? (chunk ^ ((slot & 7) << 1)) : (chunk ^ (slot & 15)))) << 4);
          asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(*(unsigned long long *)&d[0]));
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (d[0] == 0x12345678u) sink[t] = smem[t];
    return;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0 || MODE == 2) {
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(*(unsigned long long *)&d[0]), "n"(i * 512 % 4096 + (i / 8) * 32));
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(*(unsigned long long *)&d[2]), "n"(i * 512 % 4096 + (i / 8) * 32 + 64));
      } else {
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(*(__uint128_t *)&d[0]), "n"(i * 1024 % 4096 + (i / 4) * 64));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (d[0] == 0x12345678u) sink[t] = smem[t];
}
template <int MODE>
void run(const char *name, int threads) {
  float *sink; (void)hipMalloc(&sink, 4096 * 4);
  const int iters = 4000;
  float best = 1e9f;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 136 * 1024, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double bytes = (double)iters * (MODE >= 4 ? 32 * 512 : 16 * 1024) * (threads / 64);   // per CU
  printf("%-34s waves/SIMD %d : %6.1f B/ns/CU\n", name, threads / 256, bytes / (best * 1e6));
  (void)hipFree(sink);
}
int main() {
  for (int thr : {256, 512, 1024}) {
    run<0>("ds_write_b64 contiguous", thr); run<1>("ds_write_b128 contiguous", thr);
    run<2>("ds_write_b64 256-B pixel pitch", thr); run<3>("ds_write_b128 256-B pixel pitch", thr);
    run<4>("epilogue A pattern, swizzle now", thr); run<5>("epilogue A pattern, old swizzle", thr);
  }
  return 0;
}

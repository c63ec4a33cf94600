// Does s_waitcnt vmcnt(1) after {DMA A (HBM-cold), DMA B (L2-hot)} guarantee A has landed?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(64) void k(const unsigned *cold, const unsigned *hot, int iters, long cold_words, unsigned *bad, int wait_mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2048];
  const int lane = threadIdx.x;
  const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  unsigned nbad = 0;
  long off = ((long)blockIdx.x * 7919 * 4096) % cold_words;
  for (int it = 0; it < iters; ++it) {
    for (int i = lane; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned *a = cold + off + lane * 4;
    dma16(a, l0);            // A: cold
    dma16(hot + lane * 4, l0 + 1024);  // B: hot
    if (wait_mode == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned got = lds[lane * 4];
    const unsigned want = (unsigned)((off + lane * 4) & 0xffffffffu) * 2654435761u;
    if (got != want) ++nbad;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    off = (off + 1048576 + 4096 * 17) % cold_words; off &= ~3L;
  }
  if (nbad) atomicAdd(bad, nbad);
}
__global__ void fill(unsigned *p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (unsigned)(i & 0xffffffffu) * 2654435761u;
}
int main() {
  const long words = (3L << 30) / 4;
  unsigned *cold, *hot, *bad; hipMalloc(&cold, words * 4 + 65536); hipMalloc(&hot, 4096); hipMalloc(&bad, 4);
  fill<<<4096, 256>>>(cold, words + 16384); hipMemset(hot, 0, 4096);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(bad, 0, 4);
    k<<<1024, 64>>>(cold, hot, 2000, words, bad, mode);
    unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("wait_mode=%s : %u stale lanes of %ld\n", mode ? "vmcnt(1)" : "vmcnt(0)", h, 1024L * 2000 * 64);
  }
  return 0;
}

// Does the instruction offset of global_load_lds_dwordx4 apply to the LDS address as well as to the global address?  (gfx950)
// hipcc --offload-arch=gfx950 -O2 scripts/microbench/dmaoff.hip -o /tmp/dmaoff && /tmp/dmaoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *g, unsigned *out) {
  extern __shared__ unsigned smem[];
  typedef __attribute__((address_space(3))) void *lptr_t;
  for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);
  const unsigned voff = threadIdx.x * 16;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
               "global_load_lds_dword %2, %3 offset:3072\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "s"(lds0), "v"(voff), "s"(g) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = smem[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *g, *o;
  hipMalloc(&g, 16384); hipMalloc(&o, 8192);
  hipMemcpy(g, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, g, o);
  std::vector<unsigned> r(2048);
  hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
  for (int i = 0; i < 2048; i += 64) printf("lds dword %4d: %08x %08x .. %08x\n", i, r[i], r[i + 1], r[i + 63]);
  // x4 piece: global dwords 256.. (offset 1024 B); lands at LDS dword 256 if the offset applies to LDS too, at 0 otherwise
  // dword piece: 64 lanes x 4 B with voffset lane*16 (!): global dword (768 + 4*lane); lands at LDS dword 768 + lane or lane
  return 0;
}

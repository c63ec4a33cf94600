// Micro-benchmark for dense_strip.hip's 3x3 slots: one wave per SIMD (256-thread workgroups holding all of the CU's LDS),
// a loop of 12 slots per iteration; a slot = two v_mfma_f32_16x16x32_f16 on different accumulators + fillers.
//   variant bit 0: B operand from literal AGPRs (else VGPRs)      bit 1: one ds_read_b128 per slot
//   bits 2-4: number of v_fma_f32 fillers per slot (0..7)          bit 5: s_waitcnt lgkmcnt(7) in front of each pair
//   bit 6: 2 v_accvgpr_read per slot                               bit 7: accumulators in VGPRs (builtin-like), B from VGPR
// hipcc --offload-arch=gfx950 -O3 -o slotbench slotbench.hip && ./slotbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const u32x4 *src, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 40960 / 16; i += 256) ((u32x4 *)smem)[i] = src[i & 255];
  __syncthreads();
  u32x4 wa[6];
  for (int f = 0; f < 6; ++f) wa[f] = src[f * 64 + lane];
  const u32x4 b0 = src[7 * 64 + lane], b1 = src[8 * 64 + lane];
  asm volatile("v_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %1\n\tv_accvgpr_write_b32 a202, %2\n\tv_accvgpr_write_b32 a203, %3\n\t"
               "v_accvgpr_write_b32 a204, %4\n\tv_accvgpr_write_b32 a205, %5\n\tv_accvgpr_write_b32 a206, %6\n\tv_accvgpr_write_b32 a207, %7\n\ts_nop 4"
               :: "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3])
               : "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float fl[8];
  for (int i = 0; i < 8; ++i) fl[i] = (float)lane * 0.001f + i;
  float rd = 0.f;
  float ex = 1.f;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ex) : "v"(fl[0]));
  const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + lane * 16);
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if constexpr (V & 32) asm volatile("s_waitcnt lgkmcnt(7)");
      if constexpr (V & 128) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %1" : "+v"(acc[2 * s]), "+v"(acc[2 * s + 1]) : "v"(wa[s]), "v"(b0), "v"(b1));
      } else if constexpr (V & 1) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, a[200:203], %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, a[204:207], %1" : "+a"(acc[2 * s]), "+a"(acc[2 * s + 1]) : "v"(wa[s]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %1" : "+a"(acc[2 * s]), "+a"(acc[2 * s + 1]) : "v"(wa[s]), "v"(b0), "v"(b1));
      }
      if constexpr (V & 2) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wa[s]) : "v"(lp), "n"(s * 1024));
      constexpr int NF = (V >> 2) & 7;
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fl[j]) : "v"(fl[7]));
      if constexpr (V & 64) {
        float r0, r1;
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %2" : "=v"(r0), "=v"(r1) : "a"(ex));
        rd += r0 + r1;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = rd;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += fl[i];
  for (int f = 0; f < 6; ++f) s += __builtin_bit_cast(float, wa[f][0]);
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) ((long *)(out + 256 * 256))[blockIdx.x] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// one v_mfma_f32_32x32x16_f16 per slot (same MFMA time as two 16x16x32) + NF v_fma_f32 fillers (+ ds_read when DS)
template <int NF, int DS, int T>
__global__ __launch_bounds__(T) void k32(const u32x4 *src, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 40960 / 16; i += T) ((u32x4 *)smem)[i] = src[i & 255];
  __syncthreads();
  u32x4 wa[6];
  for (int f = 0; f < 6; ++f) wa[f] = src[f * 64 + lane];
  const u32x4 b0 = src[7 * 64 + lane];
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float fl[8];
  for (int i = 0; i < 8; ++i) fl[i] = (float)lane * 0.001f + i;
  const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + lane * 16);
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if constexpr (DS) asm volatile("s_waitcnt lgkmcnt(5)");
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(wa[s]), "v"(b0));
      if constexpr (DS) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wa[s]) : "v"(lp), "n"(s * 1024));
      if constexpr (NF >= 100) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NF - 100; ++j) {
          f2 &p = *(f2 *)&fl[2 * (j % 3)];
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(*(f2 *)&fl[6]));
        }
      } else {
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fl[j % 8]) : "v"(fl[7]));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += fl[i];
  for (int f = 0; f < 6; ++f) s += __builtin_bit_cast(float, wa[f][0]);
  out[blockIdx.x * T + tid] = s;
  if (tid == 0) ((long *)(out + 256 * 1024))[blockIdx.x] = t1 - t0;
}
template <int NF, int DS, int T>
void run32(const u32x4 *src, float *out, const char *what) {
  hipFuncSetAttribute((const void *)k32<NF, DS, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k32<NF, DS, T><<<256, T, 163840>>>(src, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k32<NF, DS, T><<<256, T, 163840>>>(src, out, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long> t(256);
  hipMemcpy(t.data(), out + 256 * 1024, 256 * sizeof(long), hipMemcpyDeviceToHost);
  double tk = 0;
  for (long x : t) tk += x;
  tk /= 256;
  printf("32x32x16 T=%d NF=%d DS=%d %-40s %7.2f ns per SIMD-slot (= 2 x 16x16x32)  %6.1f ticks/slot/wave  [%s]\n", T, NF, DS, what,
         ms * 1e6 / (iters * 6.0) / (T / 256), tk / (iters * 6.0), hipGetErrorString(hipGetLastError()));
}

// the 16x16x32 slot kernel again with T threads per workgroup (T / 256 waves per SIMD)
template <int NF, int DS, int T>
__global__ __launch_bounds__(T) void k16(const u32x4 *src, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 40960 / 16; i += T) ((u32x4 *)smem)[i] = src[i & 255];
  __syncthreads();
  u32x4 wa[6];
  for (int f = 0; f < 6; ++f) wa[f] = src[f * 64 + lane];
  const u32x4 b0 = src[7 * 64 + lane], b1 = src[8 * 64 + lane];
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float fl[8];
  for (int i = 0; i < 8; ++i) fl[i] = (float)lane * 0.001f + i;
  const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + lane * 16);
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if constexpr (DS) asm volatile("s_waitcnt lgkmcnt(5)");
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %1" : "+a"(acc[2 * s]), "+a"(acc[2 * s + 1]) : "v"(wa[s]), "v"(b0), "v"(b1));
      if constexpr (DS) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wa[s]) : "v"(lp), "n"(s * 1024));
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fl[j % 8]) : "v"(fl[7]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += fl[i];
  for (int f = 0; f < 6; ++f) s += __builtin_bit_cast(float, wa[f][0]);
  out[blockIdx.x * T + tid] = s;
  if (tid == 0) ((long *)(out + 256 * 1024))[blockIdx.x] = t1 - t0;
}
template <int NF, int DS, int T>
void run16(const u32x4 *src, float *out, const char *what) {
  hipFuncSetAttribute((const void *)k16<NF, DS, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k16<NF, DS, T><<<256, T, 163840>>>(src, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k16<NF, DS, T><<<256, T, 163840>>>(src, out, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long> t(256);
  hipMemcpy(t.data(), out + 256 * 1024, 256 * sizeof(long), hipMemcpyDeviceToHost);
  double tk = 0;
  for (long x : t) tk += x;
  tk /= 256;
  printf("16x16x32 T=%d NF=%d DS=%d %-40s %7.2f ns per SIMD-slot (2 MFMAs)              %6.1f ticks/slot/wave  [%s]\n", T, NF, DS, what,
         ms * 1e6 / (iters * 6.0) / (T / 256), tk / (iters * 6.0), hipGetErrorString(hipGetLastError()));
}

// one 32x32x16 MFMA + one 16-byte-per-lane global load per slot (L2-resident source, fragment-shaped: 32 pixels x 2 halves, 512-B
// pixel pitch): MODE 0 no load, 1 global_load 64-bit vaddr, 2 global_load saddr + 32-bit voffset, 3 buffer_load offen
template <int MODE, int NF>
__global__ __launch_bounds__(256) void kld(const u32x4 *src, const unsigned char *gsrc, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 wa[6];
  for (int f = 0; f < 6; ++f) wa[f] = src[f * 64 + lane];
  const u32x4 b0 = src[7 * 64 + lane];
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float fl[8];
  for (int i = 0; i < 8; ++i) fl[i] = (float)lane * 0.001f + i;
  u32x4 ld[6];
  for (int i = 0; i < 6; ++i) ld[i] = (u32x4){0, 0, 0, 0};
  const unsigned voff = (unsigned)((lane & 31) * 512 + (lane >> 5) * 64 + (blockIdx.x & 63) * 16384 + (tid >> 6) * 4096);
  const unsigned char *vp = gsrc + voff;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gsrc, 0, 1 << 22, 0x00020000);
  const long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(wa[s]), "v"(b0));
      if (s == 0 && (it & 1) == 0) {     // one load per 12 slots
      if constexpr (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(ld[s]) : "v"(vp), "n"(s * 16));
      if constexpr (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%c3" : "=v"(ld[s]) : "v"(voff), "s"(gsrc), "n"(s * 16));
      if constexpr (MODE == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%c3" : "=v"(ld[s]) : "v"(voff), "s"(rs), "n"(s * 16));
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fl[j % 8]) : "v"(fl[7]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)");
  const long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += fl[i];
  for (int f = 0; f < 6; ++f) s += __builtin_bit_cast(float, ld[f][0]);
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) ((long *)(out + 256 * 1024))[blockIdx.x] = t1 - t0;
}
template <int MODE, int NF>
void runld(const u32x4 *src, const unsigned char *gsrc, float *out, const char *what) {
  const int iters = 2000;
  hipFuncSetAttribute((const void *)kld<MODE, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kld<MODE, NF><<<256, 256, 163840>>>(src, gsrc, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kld<MODE, NF><<<256, 256, 163840>>>(src, gsrc, out, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("32x32x16 + load mode %d (%s) NF=%d: %7.2f ns per slot  [%s]\n", MODE, what, NF, ms * 1e6 / (iters * 6.0), hipGetErrorString(hipGetLastError()));
}

template <int V>
void run(const u32x4 *src, float *out, const char *what) {
  hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<V><<<256, 256, 163840>>>(src, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<V><<<256, 256, 163840>>>(src, out, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long> t(256);
  hipMemcpy(t.data(), out + 256 * 256, 256 * sizeof(long), hipMemcpyDeviceToHost);
  double tk = 0;
  for (long x : t) tk += x;
  tk /= 256;
  printf("v=%3d %-58s %7.2f ns/MFMA  %6.1f ticks/slot (2 MFMAs)  [%s]\n", V, what, ms * 1e6 / (iters * 12.0), tk / (iters * 6.0), hipGetErrorString(hipGetLastError()));
}

int main() {
  u32x4 *src; float *out;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 256 * 1024 * 4 + 4096);
  hipMemset(src, 0, 1 << 20);
  run<128>(src, out, "acc VGPR, B VGPR, no fillers");
  run<0>(src, out, "acc AGPR, B VGPR, no fillers");
  run<1>(src, out, "acc AGPR, B AGPR, no fillers");
  run<1 | 2>(src, out, "B AGPR + ds_read");
  run<1 | (2 << 2)>(src, out, "B AGPR + 2 fma");
  run<1 | (4 << 2)>(src, out, "B AGPR + 4 fma");
  run<1 | (6 << 2)>(src, out, "B AGPR + 6 fma");
  run<0 | (4 << 2)>(src, out, "B VGPR + 4 fma");
  run<0 | (6 << 2)>(src, out, "B VGPR + 6 fma");
  run<1 | 2 | (4 << 2)>(src, out, "B AGPR + ds_read + 4 fma");
  run<1 | 2 | (4 << 2) | 32>(src, out, "B AGPR + ds_read + 4 fma + waitcnt");
  run<1 | 2 | (2 << 2) | 64 | 32>(src, out, "B AGPR + ds_read + 2 fma + 2 accread + waitcnt");
  run<1 | 2 | (4 << 2) | 64 | 32>(src, out, "B AGPR + ds_read + 4 fma + 2 accread + waitcnt");
  run<0 | 2 | (4 << 2) | 64 | 32>(src, out, "B VGPR + ds_read + 4 fma + 2 accread + waitcnt");
  unsigned char *gsrc; hipMalloc(&gsrc, 1 << 22); hipMemset(gsrc, 0, 1 << 22);
  runld<0, 0>(src, gsrc, out, "no load");
  runld<1, 0>(src, gsrc, out, "global_load vaddr64");
  runld<2, 0>(src, gsrc, out, "global_load saddr+voff");
  runld<3, 0>(src, gsrc, out, "buffer_load offen");
  runld<0, 4>(src, gsrc, out, "no load");
  runld<1, 4>(src, gsrc, out, "global_load vaddr64");
  runld<2, 4>(src, gsrc, out, "global_load saddr+voff");
  runld<3, 4>(src, gsrc, out, "buffer_load offen");
  run32<0, 0, 256>(src, out, "");
  run32<2, 0, 256>(src, out, "");
  run32<4, 0, 256>(src, out, "");
  run32<6, 0, 256>(src, out, "");
  run32<8, 0, 256>(src, out, "");
  run32<104, 0, 256>(src, out, "4 v_pk_fma_f32");
  run32<106, 0, 256>(src, out, "6 v_pk_fma_f32");
  run32<108, 0, 256>(src, out, "8 v_pk_fma_f32");
  run32<10, 0, 256>(src, out, "10 v_fma");
  run32<12, 0, 256>(src, out, "12 v_fma");
  run32<4, 1, 256>(src, out, "");
  run32<6, 1, 256>(src, out, "");
  run16<0, 0, 512>(src, out, "2 waves/SIMD");
  run16<2, 0, 512>(src, out, "2 waves/SIMD");
  run16<4, 0, 512>(src, out, "2 waves/SIMD");
  run16<6, 0, 512>(src, out, "2 waves/SIMD");
  run16<8, 0, 512>(src, out, "2 waves/SIMD");
  run16<6, 1, 512>(src, out, "2 waves/SIMD");
  run32<6, 1, 512>(src, out, "2 waves/SIMD");
  run32<8, 1, 512>(src, out, "2 waves/SIMD");
  return 0;
}

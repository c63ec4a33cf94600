// The recurrent kernels' dot product for blocks whose weight columns do not fit the registers (rnn.hip, train.hip): a thread's H
// weights are split three ways - KR in registers, KL in LDS ([k/4][row][4] floats: conflict-free 16-byte reads), the rest
// streamed with 16 loads in flight - and the vector x (LDS, the same for all lanes of a wave) is not read per FMA: a lane reads
// 16 bytes per 16 k-values (lane l holds x[k0 + 4 (l mod 4) + e], e = 0..3) and the FMAs take the operand through DPP
// quad_perm:[j,j,j,j] (v_fmac_f32_dpp: lane j of the quad, broadcast to the quad) - 4 x fewer LDS instructions than one
// broadcast ds_read_b128 per four FMAs, no extra VALU work.  One accumulator, k ascending: results do not depend on KR / KL.
#pragma once
#include <hip/hip_runtime.h>

#define TN_FMA_Q(ACC, HREG, W, J)                                                                              \
  asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #J "," #J "," #J "," #J "] row_mask:0xf bank_mask:0xf"  \
               : "+v"(ACC) : "v"(HREG), "v"(W))
// 16 k-values k0 .. k0+15: WV(i) is the weight of k0 + i
#define TN_DOT16(ACC, HQ, WV)                                                                                          \
  do {                                                                                                                 \
    TN_FMA_Q(ACC, HQ.x, WV(0), 0);  TN_FMA_Q(ACC, HQ.y, WV(1), 0);  TN_FMA_Q(ACC, HQ.z, WV(2), 0);  TN_FMA_Q(ACC, HQ.w, WV(3), 0);   \
    TN_FMA_Q(ACC, HQ.x, WV(4), 1);  TN_FMA_Q(ACC, HQ.y, WV(5), 1);  TN_FMA_Q(ACC, HQ.z, WV(6), 1);  TN_FMA_Q(ACC, HQ.w, WV(7), 1);   \
    TN_FMA_Q(ACC, HQ.x, WV(8), 2);  TN_FMA_Q(ACC, HQ.y, WV(9), 2);  TN_FMA_Q(ACC, HQ.z, WV(10), 2); TN_FMA_Q(ACC, HQ.w, WV(11), 2);  \
    TN_FMA_Q(ACC, HQ.x, WV(12), 3); TN_FMA_Q(ACC, HQ.y, WV(13), 3); TN_FMA_Q(ACC, HQ.z, WV(14), 3); TN_FMA_Q(ACC, HQ.w, WV(15), 3);  \
  } while (0)

// the thread's weights KR .. KR+KL-1 into the LDS share (w_k = wcol[k * stride]; `row` of `nrows` threads)
template <int KR, int KL>
__device__ __forceinline__ void rnn_dot_fill_lds(float *wl, int nrows, int row, const float *__restrict__ wcol, long stride) {
  for (int k4 = 0; k4 < KL / 4; ++k4) {
    float4 v;
    v.x = wcol[(long)(KR + 4 * k4 + 0) * stride]; v.y = wcol[(long)(KR + 4 * k4 + 1) * stride];
    v.z = wcol[(long)(KR + 4 * k4 + 2) * stride]; v.w = wcol[(long)(KR + 4 * k4 + 3) * stride];
    *(float4 *)(wl + ((long)k4 * nrows + row) * 4) = v;
  }
}

// acc + sum over k < H of w_k x[k] (H % 16 == 0, x 16-byte aligned in LDS); lane4 = 4 * (lane & 3)
template <int KR, int KL>
__device__ __forceinline__ float rnn_dot_big(float acc, const float (&wr)[KR], const float *wl, int nrows, int row,
                                             const float *__restrict__ wcol, long stride, const float *x, int H, int lane4) {
  static_assert(KR % 16 == 0 && KL % 16 == 0, "whole 16-wide chunks of x");
#pragma unroll
  for (int k = 0; k < KR; k += 16) {
    const float4 hq = *(const float4 *)(x + k + lane4);
#define TN_WV(i) wr[k + (i)]
    TN_DOT16(acc, hq, TN_WV);
#undef TN_WV
  }
#pragma unroll
  for (int k = 0; k < KL; k += 16) {
    const float4 hq = *(const float4 *)(x + KR + k + lane4);
    float w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *(const float4 *)(wl + ((long)(k / 4 + i) * nrows + row) * 4);
      w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
#define TN_WV(i) w[i]
    TN_DOT16(acc, hq, TN_WV);
#undef TN_WV
  }
  for (int k = KR + KL; k < H; k += 16) {
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wcol[(long)(k + i) * stride];
    const float4 hq = *(const float4 *)(x + k + lane4);
#define TN_WV(i) w[i]
    TN_DOT16(acc, hq, TN_WV);
#undef TN_WV
  }
  return acc;
}

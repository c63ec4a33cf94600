// fp32 linear launcher shared by dense / rnn / gnmt translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

// One 16 x 16 tile of Y = X W^T (+ bias), fp32, by FOUR waves (wq = 0..3; all of them - and every other wave of the workgroup -
// must call: the partial tiles meet behind a __syncthreads).  Each wave takes a QUARTER of K and requests ALL its operand data
// up front (one float4 per lane and 16 k-values, 6 x 2 loads in flight) - one or two L2 round trips per wave instead of the
// twenty dependent k-tiles a 32 x 32 kernel walks - and the four partial tiles are added in wave order (deterministic).  A lane
// (r = lane & 15, q = lane >> 4) holds X[m0 + r][16 j + 4 q + e] / W[n0 + r][same k] for e = 0..3: MFMA e of chunk j contracts
// the k-values 16 j + 4 q' + e over q' = 0..3 - any bijection of k works as long as both operands use it.
// Needs ldx, ldw, K multiples of 4 and 16-byte aligned X, Wt.  m0 >= M (a workgroup with fewer tiles than wave groups): no output.
constexpr int kLatGroup = 6;
// WK4: the weights are stored k-group-major, Wt[(k / 4) * ldw + n * 4 + k % 4] with ldw = 4 * (rows of W): the 16 lanes of a
// quarter wave then read 256 contiguous bytes instead of 16 bytes from each of 16 rows 3 KiB apart.
// XK4: the same for X, X[(k / 4) * ldx + m * 4 + k % 4] with ldx = 4 * (rows of X).
template <bool WK4 = false, bool XK4 = false>
__device__ __forceinline__ void lat_tile_f32(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                             const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K,
                                             int m0, int n0, int wq, int lane, float (*red)[64][4]) {
  const int r = lane & 15, q = lane >> 4;
  const int nch = (K + 63) / 64;                     // 16-wide k chunks per wave
  const int kbeg = wq * nch * 16;
  const int xm = min(m0 + r, M - 1), wn = min(n0 + r, N - 1);
  const float *xrow = XK4 ? X + (long)q * ldx + 4 * xm : X + (long)xm * ldx + 4 * q;
  const float *wrow = WK4 ? Wt + (long)q * ldw + 4 * wn : Wt + (long)wn * ldw + 4 * q;
  const long wstep = WK4 ? (long)ldw / 4 : 1, xstep = XK4 ? (long)ldx / 4 : 1;      // floats per k in an operand's addressing
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nch; c0 += kLatGroup) {
    float4 xa[kLatGroup], wb[kLatGroup];
#pragma unroll
    for (int j = 0; j < kLatGroup; ++j) {
      const int k = kbeg + (c0 + j) * 16 + 4 * q;
      const bool ok = c0 + j < nch && k < K;        // (K % 4 == 0: a float4 is all inside or all outside)
      xa[j] = ok ? *(const float4 *)(xrow + (long)(kbeg + (c0 + j) * 16) * xstep) : make_float4(0.f, 0.f, 0.f, 0.f);
      wb[j] = ok ? *(const float4 *)(wrow + (long)(kbeg + (c0 + j) * 16) * wstep) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < kLatGroup; ++j) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j].x, wb[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j].y, wb[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j].z, wb[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j].w, wb[j].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wq][lane][e] = acc[e];
  __syncthreads();
  if (wq == 0 && m0 < M) {
    const int n = n0 + r;
    if (n < N) {
      const float bz = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + q * 4 + e;
        if (m < M) Y[(long)m * ldy + n] = ((red[0][lane][e] + red[1][lane][e]) + (red[2][lane][e] + red[3][lane][e])) + bz;
      }
    }
  }
}

int launch_linear_f32(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy,
                      int M, int N, int K, int accumulate, hipStream_t s);
// Latency-optimised form for skinny problems (few rows, K >= 128, float4-aligned operands; falls back to launch_linear_f32
// otherwise): 16 x 16 output tiles, K split over the four waves of a workgroup, every operand requested up front.
int launch_linear_f32_lat(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                          hipStream_t s);
// the weights k-group-major (see lat_tile_f32<true>): Wk4[(k / 4) * 4 N + n * 4 + k % 4], K % 4 == 0, X float4-aligned
int launch_linear_f32_lat_wk4(const float *X, int ldx, const float *Wk4, const float *bias, float *Y, int ldy, int M, int N, int K, hipStream_t s);
// (ldx < 0: X k-group-major too, with -ldx rows: X[(k / 4) * 4 (-ldx) + m * 4 + k % 4])
int launch_linear_f32_lat2(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                           int nsplit, int K2, hipStream_t s);
// Y = relu(X * asc[k] + ash[k]) W^T (+ bias): a BatchNorm + ReLU in front of the GEMM applied to the X operand while it is staged
int launch_linear_f32_bnrelu(const float *X, int ldx, const float *asc, const float *ash, const float *Wt, int ldw, const float *bias,
                             float *Y, int ldy, int M, int N, int K, int accumulate, hipStream_t s);

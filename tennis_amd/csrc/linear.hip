// fp32 linear layer  Y[M][N] = X[M][K] * W[N][K]^T + bias[N]  on the exact-f32
// matrix instruction v_mfma_f32_16x16x4_f32 (an fmaf chain in k order, so the
// result is fp32-exact in the sense of the CPU oracle up to summation order).
// Serves nn.Dense (reference models/vision/definitions.py:25,32,101,108-109),
// the i2h projections of gluon.rnn.GRU/LSTM (definitions.py:94-96) and the
// GNMT cells / tgt_proj (models/captioning/gnmt.py:369-392, SURVEY K9,K14,K15).
// 64x64 tile, BK=16, 4 waves each 32x32; generic in M, N, K with guards.
#include "common.h"
#include "linear.h"

namespace {

__global__ __launch_bounds__(256) void linear_f32_kernel(const float *__restrict__ X, int ldx,
                                                         const float *__restrict__ Wt, int ldw,
                                                         const float *__restrict__ bias, float *__restrict__ Y,
                                                         int ldy, int M, int N, int K, int accumulate) {
  __shared__ float As[64][17];
  __shared__ float Bs[64][17];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int srow = t >> 2, sk = (t & 3) * 4;
  const bool vec = ((ldx | ldw | K) & 3) == 0 && (((uintptr_t)X | (uintptr_t)Wt) & 15) == 0;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < K; k0 += 16) {
    float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    const int am = m0 + srow, bn = n0 + srow, kk = k0 + sk;
    if (vec) {
      if (am < M && kk < K) { const float4 v = *(const float4 *)(X + (long)am * ldx + kk); av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w; }
      if (bn < N && kk < K) { const float4 v = *(const float4 *)(Wt + (long)bn * ldw + kk); bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (am < M && kk + j < K) av[j] = X[(long)am * ldx + kk + j];
        if (bn < N && kk + j < K) bv[j] = Wt[(long)bn * ldw + kk + j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      As[srow][sk + j] = av[j];
      Bs[srow][sk + j] = bv[j];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kq = ks * 4 + (lane >> 4);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = As[wm * 32 + i * 16 + (lane & 15)][kq];
        b[i] = Bs[wn * 32 + i * 16 + (lane & 15)][kq];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[i=m][j=n]: lane: n = lane&15, m = (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + (lane & 15);
      if (n >= N) continue;
      const float bz = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (m < M) {
          float *dst = Y + (long)m * ldy + n;
          const float v = acc[i][j][r] + bz;
          *dst = accumulate ? (*dst + v) : v;
        }
      }
    }
}

}  // namespace

int launch_linear_f32(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy,
                      int M, int N, int K, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return TN_OK;
  const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
  hipLaunchKernelGGL(linear_f32_kernel, grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, accumulate);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

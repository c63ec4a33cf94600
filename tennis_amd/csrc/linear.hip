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

constexpr int kPD = 4;   // k-tiles of 16 in flight per thread (registers) ahead of the one being multiplied

// F = 16x16 fragments per wave in each direction: tile = (32 F) x (32 F), 4 waves in a 2 x 2 arrangement (F = 2 is
// what is launched; skinny problems go to linear_f32_skinny_kernel below).
// BN (round 4, the fine-tuning path): the X operand is transformed while it is staged, x -> relu(x * asc[k] + ash[k]) - a
// training-mode BatchNorm + ReLU in front of a 1x1 convolution never materialises its output (finetune.hip).
template <int F, bool BN = false>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float *__restrict__ X, int ldx,
                                                         const float *__restrict__ Wt, int ldw,
                                                         const float *__restrict__ bias, float *__restrict__ Y,
                                                         int ldy, int M, int N, int K, int accumulate,
                                                         const float *__restrict__ asc = nullptr, const float *__restrict__ ash = nullptr) {
  constexpr int BT = 32 * F;             // tile rows (M) = tile columns (N)
  __shared__ float As[2][BT][17];
  __shared__ float Bs[2][BT][17];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
  // staging: F = 2: every thread loads one float4 of X and one of W per k-tile; F = 1: threads 0..127 load X, 128..255 W
  const int srow = F == 2 ? t >> 2 : (t & 127) >> 2, sk = (t & 3) * 4;
  const bool doA = F == 2 || t < 128, doB = F == 2 || t >= 128;
  const bool vec = ((ldx | ldw | K) & 3) == 0 && (((uintptr_t)X | (uintptr_t)Wt) & 15) == 0;
  const int am = m0 + srow, bn = n0 + srow;
  const float *xrow = X + (long)am * ldx, *wrow = Wt + (long)bn * ldw;
  const int nk = (K + 15) / 16;

  f32x4 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // kPD k-tiles stay in flight in registers while one is multiplied; the MFMA order over k is that of a plain loop
  float av[kPD][4], bv[kPD][4];
  auto fetch = [&](int it, float *a4, float *b4) {
    const int kk = it * 16 + sk;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a4[j] = 0.f; b4[j] = 0.f; }
    if (it >= nk) return;
    if (vec) {
      if (doA && am < M && kk < K) {
        const float4 v = *(const float4 *)(xrow + kk);
        a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;
        if constexpr (BN) {
          const float4 sc = *(const float4 *)(asc + kk), sh = *(const float4 *)(ash + kk);
          a4[0] = fmaxf(fmaf(a4[0], sc.x, sh.x), 0.f); a4[1] = fmaxf(fmaf(a4[1], sc.y, sh.y), 0.f);
          a4[2] = fmaxf(fmaf(a4[2], sc.z, sh.z), 0.f); a4[3] = fmaxf(fmaf(a4[3], sc.w, sh.w), 0.f);
        }
      }
      if (doB && bn < N && kk < K) { const float4 v = *(const float4 *)(wrow + kk); b4[0] = v.x; b4[1] = v.y; b4[2] = v.z; b4[3] = v.w; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (doA && am < M && kk + j < K) {
          a4[j] = xrow[kk + j];
          if constexpr (BN) a4[j] = fmaxf(fmaf(a4[j], asc[kk + j], ash[kk + j]), 0.f);
        }
        if (doB && bn < N && kk + j < K) b4[j] = wrow[kk + j];
      }
    }
  };
#pragma unroll
  for (int p = 0; p < kPD; ++p) fetch(p, av[p], bv[p]);

  for (int it0 = 0; it0 < nk; it0 += kPD) {
#pragma unroll
    for (int p = 0; p < kPD; ++p) {
      const int it = it0 + p;
      if (it < nk) {              // block-uniform
        const int buf = it & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (doA) As[buf][srow][sk + j] = av[p][j];
          if (doB) Bs[buf][srow][sk + j] = bv[p][j];
        }
        fetch(it + kPD, av[p], bv[p]);
        __syncthreads();          // one barrier per k-tile: the other buffer is still being read by slower waves
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kq = ks * 4 + (lane >> 4);
          float a[F], b[F];
#pragma unroll
          for (int i = 0; i < F; ++i) {
            a[i] = As[buf][wm * 16 * F + i * 16 + (lane & 15)][kq];
            b[i] = Bs[buf][wn * 16 * F + i * 16 + (lane & 15)][kq];
          }
#pragma unroll
          for (int i = 0; i < F; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // D[i=m][j=n]: lane: n = lane&15, m = (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const int n = n0 + wn * 16 * F + j * 16 + (lane & 15);
      if (n >= N) continue;
      const float bz = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * F + i * 16 + (lane >> 4) * 4 + r;
        if (m < M) {
          float *dst = Y + (long)m * ldy + n;
          const float v = acc[i][j][r] + bz;
          *dst = accumulate ? (*dst + v) : v;
        }
      }
    }
}

// Skinny problems (fewer than two 64x64 tiles per CU): 32x32 tile, one 16x16 fragment per wave, BK = 32 so that a
// k-tile carries 8 MFMAs per barrier.  Same k order per output element as linear_f32_kernel (one accumulator chain).
template <bool BN = false>
__global__ __launch_bounds__(256) void linear_f32_skinny_kernel(const float *__restrict__ X, int ldx,
                                                                const float *__restrict__ Wt, int ldw,
                                                                const float *__restrict__ bias, float *__restrict__ Y,
                                                                int ldy, int M, int N, int K, int accumulate,
                                                                const float *__restrict__ asc = nullptr, const float *__restrict__ ash = nullptr) {
  __shared__ float As[2][32][33];
  __shared__ float Bs[2][32][33];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int srow = t >> 3, sk = (t & 7) * 4;
  const bool vec = ((ldx | ldw | K) & 3) == 0 && (((uintptr_t)X | (uintptr_t)Wt) & 15) == 0;
  const int am = m0 + srow, bn = n0 + srow;
  const float *xrow = X + (long)am * ldx, *wrow = Wt + (long)bn * ldw;
  const int nk = (K + 31) / 32;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int PD = 3;
  float av[PD][4], bv[PD][4];
  auto fetch = [&](int it, float *a4, float *b4) {
    const int kk = it * 32 + sk;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a4[j] = 0.f; b4[j] = 0.f; }
    if (it >= nk) return;
    if (vec) {
      if (am < M && kk < K) {
        const float4 v = *(const float4 *)(xrow + kk);
        a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;
        if constexpr (BN) {
          const float4 sc = *(const float4 *)(asc + kk), sh = *(const float4 *)(ash + kk);
          a4[0] = fmaxf(fmaf(a4[0], sc.x, sh.x), 0.f); a4[1] = fmaxf(fmaf(a4[1], sc.y, sh.y), 0.f);
          a4[2] = fmaxf(fmaf(a4[2], sc.z, sh.z), 0.f); a4[3] = fmaxf(fmaf(a4[3], sc.w, sh.w), 0.f);
        }
      }
      if (bn < N && kk < K) { const float4 v = *(const float4 *)(wrow + kk); b4[0] = v.x; b4[1] = v.y; b4[2] = v.z; b4[3] = v.w; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (am < M && kk + j < K) {
          a4[j] = xrow[kk + j];
          if constexpr (BN) a4[j] = fmaxf(fmaf(a4[j], asc[kk + j], ash[kk + j]), 0.f);
        }
        if (bn < N && kk + j < K) b4[j] = wrow[kk + j];
      }
    }
  };
#pragma unroll
  for (int p = 0; p < PD; ++p) fetch(p, av[p], bv[p]);
  for (int it0 = 0; it0 < nk; it0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int it = it0 + p;
      if (it < nk) {              // block-uniform
        const int buf = it & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          As[buf][srow][sk + j] = av[p][j];
          Bs[buf][srow][sk + j] = bv[p][j];
        }
        fetch(it + PD, av[p], bv[p]);
        __syncthreads();
        float a[8], b[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          a[ks] = As[buf][wm * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
          b[ks] = Bs[buf][wn * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks], acc, 0, 0, 0);
      }
    }
  }
  const int n = n0 + wn * 16 + (lane & 15);
  if (n < N) {
    const float bz = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
      if (m < M) {
        float *dst = Y + (long)m * ldy + n;
        const float v = acc[r] + bz;
        *dst = accumulate ? (*dst + v) : v;
      }
    }
  }
}

// Latency-bound skinny problems (the captioner's per-step gate GEMMs: 160 rows, N = 1024, K = 612 / 768): a 16 x 16 output tile per
// workgroup, its four waves each take a QUARTER of K and request ALL their operand data up front (one float4 per lane and 16
// k-values, 6 x 2 loads in flight) - one or two L2 round trips per wave instead of the twenty dependent k-tiles the 32 x 32 kernel
// walks - and the four partial tiles meet in LDS, added in wave order (deterministic).  A lane (r = lane & 15, q = lane >> 4)
// holds X[m0 + r][16 j + 4 q + e] / W[n0 + r][same k] for e = 0..3: MFMA e of chunk j contracts the k-values 16 j + 4 q' + e over
// q' = 0..3 - any bijection of k works as long as both operands use it.
#ifdef TN_LAT_STAMPS
__device__ long long g_lat_stamps[8];
#endif
__global__ __launch_bounds__(256) void linear_f32_lat_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                                             const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int Kall,
                                                             int nsplit, int K2) {
  __shared__ float red[4][64][4];
  const int n0 = blockIdx.x * 16;
  const int K = n0 >= nsplit ? K2 : Kall;            // the output columns from nsplit on only contract the first K2 k-values
#ifdef TN_LAT_STAMPS   // tuning builds only: when the first / middle / last workgroup start and end (100 MHz ticks)
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
  const int slot = wgid == 0 ? 0 : wgid == nwg / 2 ? 1 : wgid == nwg - 1 ? 2 : -1;
  if (slot >= 0 && threadIdx.x == 0) g_lat_stamps[2 * slot] = wall_clock64();
  if (slot == 0 && threadIdx.x == 0) g_lat_stamps[6] = __builtin_amdgcn_s_memtime();
#endif
  lat_tile_f32(X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, blockIdx.y * 16, n0, threadIdx.x >> 6, threadIdx.x & 63, red);
#ifdef TN_LAT_STAMPS
  if (slot >= 0 && threadIdx.x == 0) g_lat_stamps[2 * slot + 1] = wall_clock64();
  if (slot == 0 && threadIdx.x == 0) g_lat_stamps[7] = __builtin_amdgcn_s_memtime();
#endif
}

__global__ __launch_bounds__(256) void linear_f32_lat_wk4_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ Wk4,
                                                                 const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K) {
  __shared__ float red[4][64][4];
  if (ldx < 0)
    lat_tile_f32<true, true>(X, -4 * ldx, Wk4, 4 * N, bias, Y, ldy, M, N, K, blockIdx.y * 16, blockIdx.x * 16, threadIdx.x >> 6, threadIdx.x & 63, red);
  else
    lat_tile_f32<true>(X, ldx, Wk4, 4 * N, bias, Y, ldy, M, N, K, blockIdx.y * 16, blockIdx.x * 16, threadIdx.x >> 6, threadIdx.x & 63, red);
}

}  // namespace

#ifdef TN_LAT_STAMPS
extern "C" int tn_dbg_lat_stamps(long long *out) {
  TN_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lat_stamps), sizeof(long long) * 8));
  return TN_OK;
}
#endif
int launch_linear_f32_lat_wk4(const float *X, int ldx, const float *Wk4, const float *bias, float *Y, int ldy, int M, int N, int K, hipStream_t s) {
  if (M <= 0 || N <= 0) return TN_OK;
  TN_REQUIRE((ldx < 0 || (ldx & 3) == 0) && (K & 3) == 0 && (((uintptr_t)X | (uintptr_t)Wk4) & 15) == 0, "linear_f32_lat_wk4: operands must be float4-aligned");
  const dim3 grid((N + 15) / 16, (M + 15) / 16), block(256);
  hipLaunchKernelGGL(linear_f32_lat_wk4_kernel, grid, block, 0, s, X, ldx, Wk4, bias, Y, ldy, M, N, K);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_linear_f32_lat(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                          hipStream_t s) {
  return launch_linear_f32_lat2(X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, N, K, s);
}

// the same with two K ranges: the output columns n >= nsplit (a multiple of 16) contract only k < K2 (their weights beyond are
// never read) - the captioner's step GEMM computes the last cell's gates over [h0, ctx, h1_prev] and the first cell's share
// over [h0, ctx] in one launch
int launch_linear_f32_lat2(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                           int nsplit, int K2, hipStream_t s) {
  if (M <= 0 || N <= 0) return TN_OK;
  TN_REQUIRE(nsplit >= N || (nsplit % 16 == 0 && K2 > 0 && K2 <= K && K2 % 4 == 0), "linear_f32_lat2: bad column split");
  const bool vec = ((ldx | ldw | K) & 3) == 0 && (((uintptr_t)X | (uintptr_t)Wt) & 15) == 0;
  if (!vec || K < 128 || (long)((N + 15) / 16) * ((M + 15) / 16) > 4096) {
    if (nsplit >= N) return launch_linear_f32(X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, 0, s);
    if (int rc = launch_linear_f32(X, ldx, Wt, ldw, bias, Y, ldy, M, nsplit, K, 0, s)) return rc;
    return launch_linear_f32(X, ldx, Wt + (long)nsplit * ldw, ldw, bias ? bias + nsplit : nullptr, Y + nsplit, ldy, M, N - nsplit, K2, 0, s);
  }
  const dim3 grid((N + 15) / 16, (M + 15) / 16), block(256);
  hipLaunchKernelGGL(linear_f32_lat_kernel, grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, nsplit, K2);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_linear_f32(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy,
                      int M, int N, int K, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return TN_OK;
  const long big = (long)((N + 63) / 64) * ((M + 63) / 64);
  if (big >= 512) {
    const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
    hipLaunchKernelGGL((linear_f32_kernel<2, false>), grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, accumulate, (const float *)nullptr,
                       (const float *)nullptr);
  } else {   // fewer than two 64x64 tiles per CU: quarter-size tiles put four times as many CUs to work
    const dim3 grid((N + 31) / 32, (M + 31) / 32), block(256);
    hipLaunchKernelGGL(linear_f32_skinny_kernel<false>, grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, accumulate, (const float *)nullptr,
                       (const float *)nullptr);
  }
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// Y = relu(X * asc + ash) W^T (+ bias): the X operand goes through a per-column scale / shift and a ReLU while it is staged
int launch_linear_f32_bnrelu(const float *X, int ldx, const float *asc, const float *ash, const float *Wt, int ldw, const float *bias,
                             float *Y, int ldy, int M, int N, int K, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return TN_OK;
  const long big = (long)((N + 63) / 64) * ((M + 63) / 64);
  if (big >= 512) {
    const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
    hipLaunchKernelGGL((linear_f32_kernel<2, true>), grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, accumulate, asc, ash);
  } else {
    const dim3 grid((N + 31) / 32, (M + 31) / 32), block(256);
    hipLaunchKernelGGL(linear_f32_skinny_kernel<true>, grid, block, 0, s, X, ldx, Wt, ldw, bias, Y, ldy, M, N, K, accumulate, asc, ash);
  }
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

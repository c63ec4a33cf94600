// K2/K5 — fused BN+ReLU -> 1x1 convolution (+ optional 2x2 average pool) as an
// MFMA GEMM on NHWC fp16 activations.  Replaces, per DenseNet layer, the
// BatchNorm -> Activation -> Convolution(1x1) [-> Pooling(avg)] operator chain
// that gluoncv's DenseNet .features launches (reference call site
// models/vision/definitions.py:30; op inventory SURVEY §2c rows K2, K5).
//
//   Y[m][n] = sum_k relu(scale[k]*X[m][k]+shift[k]) * Wt[n][k]
//
// X is the dense block's concat buffer (row stride ldx >= K): reading the first
// K channels of every pixel IS the concat (no concat kernel).
// Tiling: 256 threads = 4 waves as 2(m) x 2(n); block tile (32*MI) x 128, BK=64;
// v_mfma_f32_16x16x32_f16 with the weight fragment as the A operand so each lane
// ends with 4 consecutive output channels of one pixel.  BN+ReLU is applied once
// per element while staging global->registers->LDS (fp32 math, one rounding);
// the next k-tile's global loads are in flight while the current tile computes.
// The epilogue transposes through LDS so every output row is one coalesced
// 256-byte store.  With POOL the four BN+ReLU'd input pixels of each pooled
// pixel are averaged before the GEMM (avgpool and 1x1 conv commute), which also
// cuts the transition GEMMs' work by 4x.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BN_TILE = 128;
constexpr int BK = 64;

template <int MI, int NB, bool POOL = false>
constexpr int smem_bytes() {
  constexpr int tiles = (POOL ? 2 : 1) * 32 * MI * 128 + NB * 128;     // POOL: the pooled operand as hi + lo tiles
  constexpr int epi = 32 * MI * (NB * 2 + 16);
  return tiles > epi ? tiles : epi;
}

// NB = output channels per workgroup (128, or 256 for the transitions with N >= 256: every column tile streams
// the pixel tile again and repeats its BN+ReLU+average, so wider tiles cut both)
// EX (exact-weights mode): w = hi + lo as two fp16 numbers, rows [hi (Kp) | lo (Kp)], Kp = K rounded up to the
// k-tile (zero-padded; api.hip::split_hi_lo_rows); the k-tile loop runs over 2 Kp, the activation side (and its BatchNorm
// constants) wrapping around after the hi half.  Round 6: also without POOL (the un-fused dense layers of map sizes no fused
// kernel tiles - the 128 x 128 block of a 512 x 512 input - in the exact-weights mode).
// ONCE (a transition with a single column tile): the activations are read exactly once, through non-temporal loads.
template <int MI, bool POOL, int NB, bool EX = false, bool ONCE = false>
__global__ __launch_bounds__(256) void conv1x1_kernel(Conv1x1Args a) {
  constexpr int BM = 32 * MI;
  constexpr int NSRC = POOL ? 4 : 1;
  constexpr int NI = NB / 32;          // 16-channel fragments per wave (a wave owns NB / 2 channels)
  constexpr int CPITCH = NB * 2 + 16;  // bytes per epilogue row: NB halfs + 8 pad
  __shared__ __attribute__((aligned(16))) unsigned char smem[smem_bytes<MI, NB, POOL>()];
  unsigned char *Xs = smem;
  // POOL (round 5): the average of four BN + ReLU'd pixels has more bits than an fp16 number holds, and rounding it was the
  // largest single rounding of the encoder on textured frames (one rounding whose result all 24 / 16 layers of the next block
  // consume through this GEMM; scripts/round_study.py: "tin").  The operand is kept as hi + lo (hi = fp16(v), lo = fp16(v - hi):
  // 22 bits) and the GEMM runs both against the same weight fragments - twice the MFMAs of a kernel that is bound by its
  // activation reads.
  unsigned char *Xl = smem + BM * 128;
  unsigned char *Ws = smem + (POOL ? 2 : 1) * BM * 128;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // POOL launches use a 1-D grid: the N/128 column tiles of one pixel tile get consecutive slots on the
  // same XCD (linear id % 8), so the activations they all read come from HBM once and from that XCD's
  // L2 for the others (the transitions read every input N/128 times otherwise)
  int mt = blockIdx.x, nt = blockIdx.y;
  if constexpr (POOL) {
    const int NT = a.N / NB, L = blockIdx.x;
    const int grp = L / (8 * NT), rem = L - grp * 8 * NT;
    mt = grp * 8 + (rem & 7);
    nt = rem >> 3;
    if (mt * BM >= a.M) return;      // padding of the last group of 8 pixel tiles
  }
  const int m0 = mt * BM;
  const int n0 = nt * NB;
  const int K = a.K;

  // staging assignment: chunk column c (8 halfs) is fixed per thread
  const int c = t & 7;
  const int r0 = t >> 3;  // 0..31

  // source row pointers for this thread's MI tile rows
  const f16 *xsrc[MI][NSRC];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m >= a.M) m = a.M - 1;
    if constexpr (POOL) {
      const int Wo = a.W >> 1, Ho = a.H >> 1;
      const int px = m % Wo;
      const int q = m / Wo;
      const int py = q % Ho;
      const int b = q / Ho;
      const long base = ((long)(b * a.H + 2 * py) * a.W + 2 * px);
      xsrc[i][0] = a.x + base * a.ldx;
      xsrc[i][1] = a.x + (base + 1) * a.ldx;
      xsrc[i][2] = a.x + (base + a.W) * a.ldx;
      xsrc[i][3] = a.x + (base + a.W + 1) * a.ldx;
    } else {
      xsrc[i][0] = a.x + (long)m * a.ldx;
    }
  }
  const int nkc = (K + BK - 1) / BK;
  const int WLD = EX ? 2 * nkc * BK : K;   // weight row pitch
  const f16 *wsrc = a.w + (long)(n0 + r0) * WLD;

  f16x8 xr[MI][NSRC];
  f16x8 wr[NI];
  float sc[8], sh[8];

  auto load_tile = [&](int kt) {
    const int kc = ((EX && kt >= nkc) ? kt - nkc : kt) * BK + c * 8;     // activation channels of this k-tile
    const int kw = kt * BK + c * 8;                                      // weight columns
    if (kc < K) {
      const float4 s0 = *(const float4 *)(a.scale + kc);
      const float4 s1 = *(const float4 *)(a.scale + kc + 4);
      const float4 t0 = *(const float4 *)(a.shift + kc);
      const float4 t1 = *(const float4 *)(a.shift + kc + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w;
      sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w;
      sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
          // a transition with ONE column tile reads every activation exactly once: non-temporal loads (no L2 / MALL
          // allocation) stream 17 % faster there (92 -> 76 us for the first transition); with two column tiles the second
          // one lives on the lines the first one left in L2, and the hint costs 3 us.  (Compile-time: under a run-time
          // condition the compiler merges the two loads into a plain one.)
          if constexpr (ONCE) xr[i][s] = __builtin_nontemporal_load((const f16x8 *)(xsrc[i][s] + kc));
          else xr[i][s] = *(const f16x8 *)(xsrc[i][s] + kc);
        }
#pragma unroll
      for (int i = 0; i < NI; ++i) wr[i] = *(const f16x8 *)(wsrc + (long)(32 * i) * WLD + kw);
    }
  };

  auto store_tile = [&](int kt) {
    const int kc = ((EX && kt >= nkc) ? kt - nkc : kt) * BK + c * 8;
    const bool kv = kc < K;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f16x8 v;
      [[maybe_unused]] f16x8 vl;
      if (kv) {
        if constexpr (POOL) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) acc += fmaxf(fmaf((float)xr[i][s][j], sc[j], sh[j]), 0.f);
            const float m = 0.25f * acc;
            v[j] = (f16)m;
            vl[j] = (f16)(m - (float)v[j]);
          }
        } else {
          v = a.clamp ? clamp8(xr[i][0], sc, sh) : bn_relu8(xr[i][0], sc, sh);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (f16)0.f; vl[j] = (f16)0.f; }
      }
      *(f16x8 *)(Xs + swz<128>(r0 + 32 * i, c)) = v;
      if constexpr (POOL) *(f16x8 *)(Xl + swz<128>(r0 + 32 * i, c)) = vl;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f16x8 v = wr[i];
      if (!kv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
      }
      *(f16x8 *)(Ws + swz<128>(r0 + 32 * i, c)) = v;
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = EX ? 2 * nkc : nkc;
  const int frow = lane & 15;
  const int fch = lane >> 4;

  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile(kt);
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (((EX && kt >= nkc) ? kt - nkc : kt) * BK + ks * 32 < K) {
        f16x8 xb[MI], wa[NI];
        [[maybe_unused]] f16x8 xl[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          xb[mi] = *(const f16x8 *)(Xs + swz<128>(wm * 16 * MI + mi * 16 + frow, ks * 4 + fch));
          if constexpr (POOL) xl[mi] = *(const f16x8 *)(Xl + swz<128>(wm * 16 * MI + mi * 16 + frow, ks * 4 + fch));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wa[ni] = *(const f16x8 *)(Ws + swz<128>(wn * (NB / 2) + ni * 16 + frow, ks * 4 + fch));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb[mi], acc[ni][mi], 0, 0, 0);
        if constexpr (POOL) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xl[mi], acc[ni][mi], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: D[i=n][j=m]; lane holds n = (lane>>4)*4 + r, m = lane&15
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = wm * 16 * MI + mi * 16 + frow;
      const int n = wn * (NB / 2) + ni * 16 + fch * 4;
      f16x4 h;
      // (bias: the shift of a BatchNorm folded behind the convolution is added BEFORE the one rounding to fp16 - added to
      // the rounded value by the consumer, every pixel of a channel would carry the same rounding offset)
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bv = *(const float4 *)(a.bias + n0 + n);
      const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (f16)(acc[ni][mi][r] + b4[r]);
      *(f16x4 *)(smem + m * CPITCH + n * 2) = h;
      if (a.y32 && m0 + m < a.M)      // (the last transition: the un-rounded result for the head, 16 B per lane)
        *(float4 *)(a.y32 + (long)(m0 + m) * a.ld32 + n0 + n) =
            make_float4(acc[ni][mi][0] + b4[0], acc[ni][mi][1] + b4[1], acc[ni][mi][2] + b4[2], acc[ni][mi][3] + b4[3]);
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI * NB / 64; ++i) {
    const int id = t + 256 * i;
    const int row = id / (NB / 8), ch = id % (NB / 8);
    const int m = m0 + row;
    if (m < a.M) {
      const uint4 v = *(const uint4 *)(smem + row * CPITCH + ch * 16);
      *(uint4 *)(a.y + (long)m * a.ldy + a.yoff + n0 + ch * 8) = v;
    }
  }
}

}  // namespace

int launch_conv1x1(const Conv1x1Args &a, hipStream_t s) {
  TN_REQUIRE(a.K % 32 == 0 && a.N % BN_TILE == 0, "conv1x1: K%32 or N%128");
  TN_REQUIRE(a.ldx % 8 == 0 && a.ldy % 8 == 0 && a.yoff % 8 == 0, "conv1x1: strides must be multiples of 8");
  const dim3 block(256);
  static const bool no_ws = getenv("TN_NO_TRANS_WS") != nullptr;      // A/B runs
  if (a.wfrag && !no_ws && trans_ws_supported(a)) return launch_trans_ws(a, s);
  if (a.pool) {
    static const bool narrow = getenv("TN_TRANS_NARROW") != nullptr;   // A/B runs: 128-channel tiles everywhere
    const bool wide = a.N % 256 == 0 && !narrow;
    const int mtiles = (a.M + 63) / 64, NT = a.N / (wide ? 256 : BN_TILE);
    const dim3 grid(((mtiles + 7) / 8) * 8 * NT);
    if (a.exact) {
      TN_REQUIRE(a.K % 64 == 0, "conv1x1: the exact-weights mode needs K % 64 == 0");
      if (wide) hipLaunchKernelGGL((conv1x1_kernel<2, true, 256, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv1x1_kernel<2, true, 128, true>), grid, block, 0, s, a);
    } else {
      // non-temporal activation loads: one column tile (every byte is read once) AND a buffer too large to be found in
      // the 256 MB MALL.  Measured at 256 frames in one launch: transition 1 (411 MB) 92 -> 76 us; transition 2 (205 MB,
      // partly served by the MALL after the layers that wrote it) 52 -> 53.5 us with the hint.  The encoder runs two
      // half batches side by side (2 x 205 MB for transition 1): threshold 128 MB, +0.5 ... 1 % end to end against no
      // hint (same-box A/B; 64 MB and 256 MB are in between)
      static const bool no_nt = getenv("TN_TRANS_NO_NT") != nullptr;   // A/B runs
      static const size_t nt_mb = getenv("TN_TRANS_NT_MB") ? (size_t)atoi(getenv("TN_TRANS_NT_MB")) : 128;   // tuning hook
      const bool stream = NT == 1 && !no_nt && (size_t)a.M * 4 * a.ldx * sizeof(f16) > (nt_mb << 20);
      if (wide && stream) hipLaunchKernelGGL((conv1x1_kernel<2, true, 256, false, true>), grid, block, 0, s, a);
      else if (wide) hipLaunchKernelGGL((conv1x1_kernel<2, true, 256>), grid, block, 0, s, a);
      else if (stream) hipLaunchKernelGGL((conv1x1_kernel<2, true, 128, false, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv1x1_kernel<2, true, 128>), grid, block, 0, s, a);
    }
  } else if (a.M >= 128 * 512) {
    const dim3 grid((a.M + 127) / 128, a.N / BN_TILE);
    if (a.exact) hipLaunchKernelGGL((conv1x1_kernel<4, false, 128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv1x1_kernel<4, false, 128>), grid, block, 0, s, a);
  } else if (a.M >= 64 * 512) {
    const dim3 grid((a.M + 63) / 64, a.N / BN_TILE);
    if (a.exact) hipLaunchKernelGGL((conv1x1_kernel<2, false, 128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv1x1_kernel<2, false, 128>), grid, block, 0, s, a);
  } else {
    const dim3 grid((a.M + 31) / 32, a.N / BN_TILE);
    if (a.exact) hipLaunchKernelGGL((conv1x1_kernel<1, false, 128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv1x1_kernel<1, false, 128>), grid, block, 0, s, a);
  }
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// Whole-frame dense blocks (14x14, 7x7), two layers per pass over the activations.
//
// In these blocks one workgroup owns one frame and walks all layers of the block inside one launch
// (dense_layer_big.hip, CHAIN).  Its K loop is bound by streaming the frame's concat buffer — and layer l+1
// streams exactly the channels layer l just streamed, plus 32.  Here the two share ONE pass: every activation
// fragment read from the LDS ring is normalised twice (BN1 of layer l, BN1 of layer l+1) and multiplied into two
// accumulator sets with the two layers' 1x1 weights.  Layer l is then finished as usual (BN2+ReLU -> LDS tile ->
// 3x3 -> 32 new channels, stored to the concat buffer); those 32 channels, still sitting in the LDS output row
// buffer, are the one k-step layer l+1 is missing: BN1 of layer l+1 on them, its last 32 weight columns straight
// from global memory, 8 more MFMAs per pixel fragment — and layer l+1 is finished the same way.
// Channels are accumulated in the same order as in the layer-at-a-time kernels, so the result is bit-identical.
//
// LDS: two ring stages of [activations | W1 of layer l | W1 of layer l+1] alias the bottleneck tile; BN tables
// of both layers behind them.  The next pair's first stage and tables are requested during the last store.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void *lptr_t;

template <int W, int BM>
struct PG {
  static constexpr int WP = W + 2, TR = W + 2, NSLOT = TR * WP;
  static constexpr int NF16 = (W * WP + 15) / 16, MAXU = (NF16 + 7) / 8;
  static constexpr int RSLOT = WP + 16 * NF16 + WP + 2;
  static constexpr int TSLOT = NSLOT > RSLOT ? NSLOT : RSLOT;
  static constexpr int TILE_BYTES = TSLOT * 256;
  static constexpr int BK = 64, ROWB = 128;
  static constexpr int XS = BM * ROWB, WS = 128 * ROWB, STAGE = XS + 2 * WS, RING = 2 * STAGE;
  static constexpr int XPIECES = XS / 1024, WPIECES = WS / 1024, PIECES = STAGE / 1024, PPW = PIECES / 8;
  static constexpr int W3RING = TILE_BYTES;
  static constexpr int TAB = (TILE_BYTES + 16384 > RING) ? TILE_BYTES + 16384 : RING;
  static constexpr int TABL = 1024 + 8192;                 // per layer: s2|t2 (256 floats), s1[1024]|t1[1024]
  static constexpr int LDS_BYTES = TAB + 2 * TABL;
  static constexpr int MIW = BM / 128;
  static_assert(PIECES % 8 == 0 && LDS_BYTES <= 160 * 1024 && BM >= W * W && BM % 128 == 0, "geometry");
  static_assert(16 * NF16 * 80 <= STAGE, "the output row buffer must stay inside ring slot 0");
};

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ (row & 7); }

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {   // see dense_layer_big.hip
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int W, int BM>
__global__ __launch_bounds__(512) void dense_block_pair_kernel(DenseLayerArgs a) {
  using G = PG<W, BM>;
  constexpr int WP = G::WP, TR = G::TR, MIW = G::MIW, PPW = G::PPW, BK = G::BK, ROWB = G::ROWB;
  constexpr int NF16 = G::NF16, MAXU = G::MAXU, MA = W * W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem;
  unsigned char *ring3 = smem + G::W3RING;      // 3x3 weight ring (2 x 8 KiB)
  unsigned char *obuf = smem;                   // output row buffer (80-B pitch), aliases the tile / ring slot 0
  const int ldc = a.ldc, K0 = a.K, nlayers = a.nchain;
  const int img = blockIdx.x;
  f16 *const fbase = a.buf + (long)img * MA * ldc;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);
  bool primed = false;

  for (int layer = 0; layer < nlayers; layer += 2) {
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));              // keeps the body's address arithmetic inside the loop (VGPR budget)
    const int t = t_, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int frow = lane & 15, fch = lane >> 4;
    const DenseLayerDev dA = a.chain[layer], dB = a.chain[layer + 1];
    const int K = K0 + 32 * layer, K1 = K + 32, nk = (K + BK - 1) / BK;
    float *tab2A = (float *)(smem + G::TAB), *tab1A = tab2A + 256;
    float *tab2B = (float *)(smem + G::TAB + G::TABL), *tab1B = tab2B + 256;

    // per-lane DMA sources of this wave's pieces, pointing at channel stage 0
    const f16 *src[PPW];
    auto set_src = [&](const f16 *wA, int kA, const f16 *wB, int kB) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int piece = wid * PPW + j, prow = lane >> 3, p = lane & 7;
        if (piece < G::XPIECES) {
          const int row = piece * 8 + prow, m = row < MA ? row : MA - 1;
          src[j] = fbase + (long)m * ldc + swz128(row, p) * 8;
        } else if (piece < G::XPIECES + G::WPIECES) {
          const int row = (piece - G::XPIECES) * 8 + prow;
          src[j] = wA + (long)row * kA + swz128(row, p) * 8;
        } else {
          const int row = (piece - G::XPIECES - G::WPIECES) * 8 + prow;
          src[j] = wB + (long)row * kB + swz128(row, p) * 8;
        }
      }
    };
    auto issue_piece = [&](int j, int slot, int stage) {
      dma16(src[j] + stage * BK, lds0 + slot * G::STAGE + (wid * PPW + j) * 1024);
    };
    if (!primed) {
      set_src(dA.w1, K, dB.w1, K1);
#pragma unroll
      for (int j = 0; j < PPW; ++j) issue_piece(j, 1, 0);       // the i-th stage lives in slot (1 + i) & 1
      for (int i = t; i < K1; i += 512) {
        if (i < K) { tab1A[i] = dA.s1[i]; tab1A[1024 + i] = dA.t1[i]; }
        tab1B[i] = dB.s1[i]; tab1B[1024 + i] = dB.t1[i];
      }
      if (t < 256) {
        tab2A[t] = t < 128 ? dA.s2[t] : dA.t2[t - 128];
        tab2B[t] = t < 128 ? dB.s2[t] : dB.t2[t - 128];
      }
      __syncthreads();
    }
    primed = false;
    const f16x8 *w3A = (const f16x8 *)dA.w3p + 72 * 64 + t, *w3B = (const f16x8 *)dB.w3p + 72 * 64 + t;
    f16x8 wq[3] = {w3A[0], w3A[512], w3A[2 * 512]};

    f32x4 acc0[8][MIW], acc1[8][MIW];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        acc0[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc1[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }

    // ============ shared K loop over channels [0, K): both layers' bottleneck GEMMs ============
    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt<0>();                          // stage kt (the only one in flight) has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool refill = kt + 1 < nk;
      const int slot = (1 + kt) & 1, rslot = slot ^ 1;
      const unsigned char *Xs = smem + slot * G::STAGE;
      // pieces [lo, hi) of stage kt+1 go out behind a block of MFMAs (compile-time ranges: src[] stays in registers)
      auto refill_range = [&](auto lot, auto hit) {
        if (refill) {
#pragma unroll
          for (int j = decltype(lot)::value; j < decltype(hit)::value; ++j) issue_piece(j, rslot, kt + 1);
        }
      };
      bool issued[4] = {false, false, false, false};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (kt * BK + ks * 32 < K) {
          const int kb = kt * BK + ks * 32 + fch * 8;
          f16x8 xraw[MIW];
#pragma unroll
          for (int mi = 0; mi < MIW; ++mi) {
            const int row = wid * (BM / 8) + mi * 16 + frow;
            xraw[mi] = *(const f16x8 *)(Xs + row * ROWB + (swz128(row, ks * 4 + fch) << 4));
          }
#pragma unroll
          for (int L = 0; L < 2; ++L) {
            const float *tb = L ? tab1B : tab1A;
            const unsigned char *Ws = Xs + G::XS + L * G::WS;
            const float4 s0 = *(const float4 *)(tb + kb), s1 = *(const float4 *)(tb + kb + 4);
            const float4 t0 = *(const float4 *)(tb + 1024 + kb), t1 = *(const float4 *)(tb + 1024 + kb + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            f16x8 wa[8];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
              const int row = ni * 16 + frow;
              wa[ni] = *(const f16x8 *)(Ws + row * ROWB + (swz128(row, ks * 4 + fch) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
              const f16x8 xb = bn_relu8_mix(xraw[mi], sc, sh);
#pragma unroll
              for (int ni = 0; ni < 8; ++ni) {
                if (L == 0) acc0[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb, acc0[ni][mi], 0, 0, 0);
                else acc1[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb, acc1[ni][mi], 0, 0, 0);
              }
            }
            if (ks == 0 && L == 0) { refill_range(std::integral_constant<int, 0>{}, std::integral_constant<int, PPW / 4>{}); issued[0] = true; }
            if (ks == 0 && L == 1) { refill_range(std::integral_constant<int, PPW / 4>{}, std::integral_constant<int, PPW / 2>{}); issued[1] = true; }
            if (ks == 1 && L == 0) { refill_range(std::integral_constant<int, PPW / 2>{}, std::integral_constant<int, 3 * PPW / 4>{}); issued[2] = true; }
            if (ks == 1 && L == 1) { refill_range(std::integral_constant<int, 3 * PPW / 4>{}, std::integral_constant<int, PPW>{}); issued[3] = true; }
          }
        }
      }
      // blocks skipped by the K guard (last stage of a K % 64 == 32 layer never refills: it is the last stage)
      (void)issued;
    }
    __syncthreads();   // nobody reads the ring any more: the tile that aliases it may be written

    // ============ finish one layer: BN2+ReLU -> tile, 3x3 -> output row buffer ============
    auto finish = [&](f32x4 (&acc)[8][MIW], const float *tab2, const f16x8 *w3) {
      {   // zero padding of the tile: pad columns of every row, the rows above and below the frame
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        if (t < TR * 32) {
          const int tr = t >> 5, side = (t >> 4) & 1, ch = t & 15;
          *(uint4 *)(tile + (tr * WP + side * (WP - 1)) * 256 + ch * 16) = z4;
        }
        for (int idx = t; idx < WP * 16; idx += 512) {
          *(uint4 *)(tile + idx * 16) = z4;
          *(uint4 *)(tile + (TR - 1) * WP * 256 + idx * 16) = z4;
        }
      }
      {   // epilogue A
        unsigned char *dst[MIW];
        int sl15[MIW];
        bool ok[MIW];
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
          const int m = wid * (BM / 8) + mi * 16 + frow;
          const int rr = m / W, x = m - rr * W;
          const int slot = (rr + 1) * WP + x + 1;
          dst[mi] = tile + slot * 256 + (fch & 1) * 8;
          sl15[mi] = slot & 15;
          ok[mi] = m < MA;
        }
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          const float4 sv = *(const float4 *)(tab2 + ni * 16 + fch * 4);
          const float4 tv = *(const float4 *)(tab2 + 128 + ni * 16 + fch * 4);
          const int chunk = ni * 2 + (fch >> 1);
#pragma unroll
          for (int mi = 0; mi < MIW; ++mi) {
            const f16x4 hv = bn_relu4_from_f32(acc[ni][mi], sv, tv);
            if (ok[mi]) *(f16x4 *)(dst[mi] + ((chunk ^ sl15[mi]) << 4)) = hv;
          }
        }
      }
      *(f16x8 *)(ring3 + t * 16) = wq[0];
      wq[0] = w3[3 * 512];
      __syncthreads();
      // phase B (see dense_layer_big.hip): 16-slot fragments, whole K per wave
      const int wpos = (wid & 3) * 2 + (wid >> 2);
      const int u0 = (wpos * NF16) >> 3, u1 = ((wpos + 1) * NF16) >> 3;
      const int px = lane & 15, kg = lane >> 4;
      f32x4 bacc[MAXU][2];
#pragma unroll
      for (int j = 0; j < MAXU; ++j) {
        bacc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bacc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      auto phase_b = [&](auto nfr_tag) {
        constexpr int NFR = decltype(nfr_tag)::value;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) {
            *(f16x8 *)(ring3 + ((tap + 1) & 1) * 8192 + t * 16) = wq[(tap + 1) % 3];
            if (tap + 4 < 9) wq[(tap + 1) % 3] = w3[(tap + 4) * 512];
          }
          const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
          const int off = WP + dy * WP + dx + px + 16 * u0;
          const unsigned char *wring = ring3 + (tap & 1) * 8192 + lane * 16;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            f16x8 wf0 = {0, 0, 0, 0, 0, 0, 0, 0}, wf1 = wf0;
            if constexpr (NFR > 0) {
              wf0 = *(const f16x8 *)(wring + (kk * 2) * 1024);
              wf1 = *(const f16x8 *)(wring + (kk * 2 + 1) * 1024);
            }
            const int chunk = kk * 4 + kg;
            f16x8 xf[NFR > 0 ? NFR : 1];
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
              const int slot = off + 16 * j;
              xf[j] = *(const f16x8 *)(tile + slot * 256 + ((chunk ^ (slot & 15)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
              bacc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0, xf[j], bacc[j][0], 0, 0, 0);
              bacc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1, xf[j], bacc[j][1], 0, 0, 0);
            }
          }
          __syncthreads();
        }
      };
      if (u1 - u0 == MAXU) phase_b(std::integral_constant<int, MAXU>{});
      else phase_b(std::integral_constant<int, MAXU - 1>{});
#pragma unroll
      for (int j = 0; j < MAXU; ++j) {
        if (u0 + j < u1) {
          const int srel = 16 * (u0 + j) + px;
#pragma unroll
          for (int nf = 0; nf < 2; ++nf) {
            f16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (f16)bacc[j][nf][r];
            *(f16x4 *)(obuf + srel * 80 + (nf * 16 + kg * 4) * 2) = hv;
          }
        }
      }
      __syncthreads();
    };
    auto copy_out = [&](int kout) {             // output row buffer -> channels [kout, kout+32) of the frame
      f16 *ybase = fbase + kout;
      for (int id = t; id < MA * 4; id += 512) {
        const int pi = id >> 2, c = id & 3;
        const int r = pi / W, x = pi - r * W;
        const uint4 v = *(const uint4 *)(obuf + (r * WP + x + 1) * 80 + c * 16);
        *(uint4 *)(ybase + (long)pi * ldc + c * 8) = v;
      }
    };

    // ---- layer l ----
    finish(acc0, tab2A, w3A);
    wq[0] = w3B[0]; wq[1] = w3B[512]; wq[2] = w3B[2 * 512];      // layer l+1's first taps
    {   // layer l+1's missing k-step: the 32 channels layer l just produced, from the output row buffer
      const int kb = K + fch * 8;
      const float4 s0 = *(const float4 *)(tab1B + kb), s1 = *(const float4 *)(tab1B + kb + 4);
      const float4 t0 = *(const float4 *)(tab1B + 1024 + kb), t1 = *(const float4 *)(tab1B + 1024 + kb + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      f16x8 wa[8];
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) wa[ni] = *(const f16x8 *)(dB.w1 + (long)(ni * 16 + frow) * K1 + kb);
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        int m = wid * (BM / 8) + mi * 16 + frow;
        if (m >= MA) m = MA - 1;
        const int rr = m / W, x = m - rr * W;
        const f16x8 xraw = *(const f16x8 *)(obuf + (rr * WP + x + 1) * 80 + fch * 16);
        const f16x8 xb = bn_relu8_mix(xraw, sc, sh);
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc1[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb, acc1[ni][mi], 0, 0, 0);
      }
    }
    copy_out(K);
    __syncthreads();   // every wave has read the row buffer: the tile may be written again
    // ---- layer l+1 ----
    finish(acc1, tab2B, w3B);
    float nxs[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, nxt[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, nx2[2] = {0.f, 0.f};
    const bool more = layer + 2 < nlayers;
    if (more) {        // next pair: first stage and the BN tables on their way during the store
      const DenseLayerDev eA = a.chain[layer + 2], eB = a.chain[layer + 3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = t + 512 * i;
        if (idx < K + 64) { nxs[0][i] = eA.s1[idx]; nxt[0][i] = eA.t1[idx]; }
        if (idx < K + 96) { nxs[1][i] = eB.s1[idx]; nxt[1][i] = eB.t1[idx]; }
      }
      if (t < 256) {
        nx2[0] = t < 128 ? eA.s2[t] : eA.t2[t - 128];
        nx2[1] = t < 128 ? eB.s2[t] : eB.t2[t - 128];
      }
      set_src(eA.w1, K + 64, eB.w1, K + 96);
#pragma unroll
      for (int j = 0; j < PPW; ++j) issue_piece(j, 1, 0);
      primed = true;
    }
    copy_out(K1);
    if (more) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = t + 512 * i;
        if (idx < K + 64) { tab1A[idx] = nxs[0][i]; tab1A[1024 + idx] = nxt[0][i]; }
        if (idx < K + 96) { tab1B[idx] = nxs[1][i]; tab1B[1024 + idx] = nxt[1][i]; }
      }
      if (t < 256) { tab2A[t] = nx2[0]; tab2B[t] = nx2[1]; }
    }
    wait_vmcnt<0>();
    __syncthreads();
  }
}

template <int W, int BM>
int launch_pair(const DenseLayerArgs &a, hipStream_t s) {
  using G = PG<W, BM>;
  static bool attr_set = false;
  if (!attr_set) {
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_block_pair_kernel<W, BM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL((dense_block_pair_kernel<W, BM>), dim3(a.B), dim3(512), G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

}  // namespace

bool dense_block_pair_supported(const DenseLayerArgs &a) {
  return a.nchain >= 2 && (a.nchain & 1) == 0 && a.chain && a.H == a.W && (a.H == 14 || a.H == 7) && a.K >= 128 &&
         a.K % 32 == 0;
}

int launch_dense_block_pair(const DenseLayerArgs &a, hipStream_t s) {
  TN_REQUIRE(dense_block_pair_supported(a), "dense_block_pair: needs an even chain of whole-frame layers (14x14 / 7x7)");
  const int klast = a.K + 32 * (a.nchain - 1);
  TN_REQUIRE(klast <= 1024 && a.ldc % 8 == 0 && klast + 32 <= a.ldc, "dense_block_pair: bad channel geometry");
  return a.H == 14 ? launch_pair<14, 256>(a, s) : launch_pair<7, 128>(a, s);
}

// Fused DenseNet dense layer, second generation (SURVEY §7 H1, §2c rows K2+K3+K4):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// replacing gluoncv's BatchNorm-Activation-Conv1x1-BatchNorm-Activation-Conv3x3-Concat chain (reference call site
// models/vision/definitions.py:30).  Same tiling as dense_layer_big.hip (a workgroup of 8 waves owns ROUT full image
// rows, the 128-channel bottleneck tile lives in LDS, persistent tiles / chained layers) but the activations no
// longer pass through LDS on their way to the bottleneck GEMM:
//
//   * X (activations) go global -> VGPR in MFMA-fragment shape (lane = pixel row x 16-byte channel chunk; the two
//     halves of a 128-byte line are fetched by consecutive k-steps) through a register ring two stages deep.
//     The ring does not alias the bottleneck tile, so the loads of the NEXT tile / layer are issued while phase B
//     of the current one runs: HBM streams during the phases that used to leave it idle.
//   * Only the 1x1 weights (shared by the 8 waves) are staged in LDS: global -> VGPR at the start of a stage ->
//     ds_write at its end, two slots (one in the 3x3-weight ring's place, one in the dead tile), one barrier per stage.
//   * Every vector-memory access of the steady state is issued through inline asm, invisible to hipcc's waitcnt
//     pass (which answers loop-carried loads behind conditional issue with s_waitcnt vmcnt(0): measured), and every
//     consumer sits behind a hand-counted s_waitcnt vmcnt(N) that names the registers it releases.  The issue
//     order is kept STATIC (loads past the end of the K range or without a next tile are issued anyway, at a
//     lane-invariant address) so that N is a compile-time constant: per tile the queue is
//         X0 W0 X1 | W1 X2 | W2 X3 | ...          (requested by the previous tile's last stages | stage 0 | 1 ...)
//     i.e. strictly alternating X(j), W(j): stage kt waits for X(kt) with W(kt) X(kt+1) younger (N = GRP) and for
//     W(kt+1) with X(kt+2) younger (N = XL).  Waits only COUNT loads that are younger than the target: the (asm)
//     stores in the queue can make a wait stricter, never weaker.
//   * The 32 new channels leave the phase-B accumulators as one 16-byte store per lane (third packed 3x3 layout,
//     api.hip pack_conv3x3): no LDS row buffer, no barrier between phase B and the store.
#include <type_traits>

#include "common.h"

namespace {

template <int W, int ROUT, int BM, int KS>
struct DL2Geom {
  static constexpr int WP = W + 2;
  static constexpr int TR = ROUT + 2;
  static constexpr int NSLOT = TR * WP;
  static constexpr int NF = (ROUT * WP + 31) / 32;
  static constexpr int NF16 = (ROUT * WP + 15) / 16;         // 16-slot output fragments of phase B
  static constexpr int MAXU = (NF16 + 7) / 8;                // ... per wave
  static constexpr int RSLOT = WP + 32 * NF + WP + 2;        // highest slot phase B touches + 1
  static constexpr int TSLOT = NSLOT > RSLOT ? NSLOT : RSLOT;
  static constexpr int TILE_BYTES = TSLOT * 256;
  static constexpr int BK = 32 * KS;                         // channels per 1x1-weight stage
  static constexpr int ROWB = BK * 2;                        // bytes per staged weight row
  static constexpr int WS = 128 * ROWB;                      // one weight stage: 8 or 16 KiB
  static constexpr int CPR = ROWB / 16;                      // 16-byte chunks per staged row
  static constexpr int W3RING = TILE_BYTES;                  // 2 x 8 KiB ring of 3x3 weights = weight slot A (even stages)
  static constexpr int WSLOT_B = 0;                          // weight slot B (odd stages): start of the dead tile
  static constexpr int TAB = TILE_BYTES + 16384;
  static constexpr int TAB2 = TAB;                           // s2[128], t2[128]
  static constexpr int TAB1 = TAB + 1024;                    // s1[K], t1[K]  (K <= 1024)
  static constexpr int LDS_BYTES = TAB + 1024 + 8192;
  static constexpr bool NSPLIT = (BM == 64);                 // 7x7: waves split 4 (pixel rows) x 2 (bottleneck channel halves)
  static constexpr int NI = NSPLIT ? 4 : 8;                  // 16-channel weight fragments per wave
  static constexpr int MIW = NSPLIT ? 1 : BM / 128;          // 16-row pixel fragments per wave
  static constexpr int DX = 2;                               // X stages in flight (register ring slots)
  static_assert(WS <= 16384, "a weight stage must fit the 3x3 ring's place");
  static_assert(WS <= TILE_BYTES, "a weight stage must fit the dead tile");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
  static_assert(BM >= TR * W && (BM % 128 == 0 || BM == 64), "phase A tile too small");
  static_assert(WS % (512 * 16) == 0, "a weight stage is KS 16-byte pieces per thread");
};

template <int BK>
__device__ __forceinline__ int wswz(int row, int chunk) {
  if constexpr (BK == 64) return chunk ^ (row & 7);
  else return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

__device__ __forceinline__ void lds_barrier() {     // LDS hand-off between waves; leaves vector-memory loads in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// 16-byte global store hidden from hipcc's waitcnt bookkeeping (see the header); s_nop: cdna_hip_programming.md 5.7
__device__ __forceinline__ void store16_hidden(void *p, f16x8 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// 16-byte load dst <- [sbase + voff + IMM], hidden from hipcc: consumers sit behind vm_wait<N>(dst) / vm_tie(dst).
// s_nop 4: a freshly written SGPR base must age five states before a VMEM instruction reads it (5.7 item 2).
template <int IMM>
__device__ __forceinline__ void gload16(f16x8 &dst, unsigned voff, const void *sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait(f16x8 &r, bool drain = false) {     // at most N younger loads stay in flight; r is valid afterwards
  (void)drain;
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}
__device__ __forceinline__ void vm_tie(f16x8 &r) {      // orders r's consumers behind the preceding vm_wait
  asm volatile("" : "+v"(r)::"memory");
}
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void *)(((unsigned long long)hi << 32) | lo);
}

template <int W, int ROUT, int BM, int KS, bool CHAIN>
__global__ __launch_bounds__(512) void dense_layer2_kernel(DenseLayerArgs a) {
  using G = DL2Geom<W, ROUT, BM, KS>;
  constexpr int WP = G::WP, TR = G::TR, MIW = G::MIW, NI = G::NI, BK = G::BK, ROWB = G::ROWB, CPR = G::CPR, DX = G::DX;
  constexpr int XL = KS * MIW, WL = KS, GRP = XL + WL;   // loads per X stage, per W stage (per wave), per (X, W) pair
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem;
  unsigned char *ring = smem + G::W3RING;
  float *tab2 = (float *)(smem + G::TAB2);
  float *tab1 = (float *)(smem + G::TAB1);

  const int H = a.H, ldc = a.ldc;
  const bool dbg_drain = (a.variant >> 16) & 1;   // debug: every counted wait drains the queue
#define DL2_STAMP(i) do { if (a.ts && threadIdx.x == 0) a.ts[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  DL2_STAMP(0);
  const int tiles_per_img = H / ROUT;
  struct TileAt { int img, r0, MA, top_pad; const f16 *xbase; };
  auto tile_at = [&](int vb) {     // all row-tiles of a frame share (virtual) blockIdx % 8 (the XCD) when B % 8 == 0
    int img, tix;
    if ((a.B & 7) == 0) {
      const int j = vb >> 3;
      img = (j / tiles_per_img) * 8 + (vb & 7);
      tix = j % tiles_per_img;
    } else {
      img = vb / tiles_per_img;
      tix = vb % tiles_per_img;
    }
    TileAt ta;
    ta.img = img;
    ta.r0 = tix * ROUT;
    const int rlo = ta.r0 > 0 ? ta.r0 - 1 : 0;
    const int rhi = (ta.r0 + ROUT < H) ? ta.r0 + ROUT + 1 : H;
    ta.MA = (rhi - rlo) * W;
    ta.top_pad = (ta.r0 == 0) ? 1 : 0;
    ta.xbase = a.buf + ((long)img * H * W + (long)rlo * W) * ldc;
    return ta;
  };
  static_assert(!CHAIN || ROUT == W, "layer chaining needs whole-frame tiles");

  // ---- state that lives across tiles / layers: the X register ring and the 1x1-weight stages in flight.  The ring
  // runs THROUGH tile / layer boundaries: the last stages of a K loop already request the first stages of what the
  // workgroup does next, so stage kt of a tile lives in slot (kt + ph) % 2 with a phase ph that advances by nk. ----
  f16x8 xq[DX][KS][MIW];
  f16x8 wr[KS];
  bool primed = false;     // the ring holds X(0), X(1) of this tile, its W(0) sits in weight slot A
  int ph = 0;

  const int nvb = a.B * tiles_per_img;
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  const TileAt ta = tile_at(vb);
  const int img = ta.img, r0 = ta.r0, MA = ta.MA, top_pad = ta.top_pad;
  const f16 *xbase = ta.xbase;
  const int nlayers = CHAIN ? a.nchain : 1;
  const int K0 = a.K;
  for (int layer = 0; layer < nlayers; ++layer) {
  int t_ = threadIdx.x;
  asm volatile("" : "+v"(t_));     // laundered per layer: keeps the address arithmetic of the body out of the outer loops
  const int t = t_;
  const int lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  if constexpr (CHAIN) {
    const DenseLayerDev d = a.chain[layer];
    a.K = K0 + 32 * layer;
    a.s1 = d.s1; a.t1 = d.t1; a.w1 = d.w1; a.s2 = d.s2; a.t2 = d.t2; a.w3p = d.w3p;
  }
  const int K = a.K;
  const int nkr = (K + BK - 1) / BK;          // stages that carry channels
  const int nk = nkr;                         // (the ring runs through ONE tile boundary: nkr >= DX, checked by the launcher)
  const int frow = lane & 15, fch = lane >> 4;
  const bool REBAL = !G::NSPLIT && MIW == 4 && (TR * W + 15) / 16 == 28;   // 28x28: 28 real fragments dealt 4/3 over the wave pairs
  const int mrow0 = G::NSPLIT ? (wid & 3) * 16 : REBAL ? (wid < 4 ? wid * 64 : 256 + (wid - 4) * 48) : wid * (BM / 8);
  const int nfw = (REBAL && wid >= 4) ? 3 : MIW;
  const int nch0 = G::NSPLIT ? (wid >> 2) * 64 : 0;

  // ---- what comes next for this workgroup: its next tile (same layer) or the next layer of its frame, as DELTAS
  // to this tile's values (a select between two captured variables inside the nested lambdas below sends both
  // through scratch memory: measured; a select between a value and zero does not) ----
  bool have_next = false;
  long dXb = 0, dWb = 0;        // byte distance of the next tile's first activation row / of the next layer's 1x1 weights
  int dK = 0, dMA = 0, nx_nkr = 0;
  DenseLayerDev nx_dev;
  if constexpr (!CHAIN) {
    if (vb + (int)gridDim.x < nvb) {
      const TileAt nx = tile_at(vb + gridDim.x);
      dXb = (const char *)nx.xbase - (const char *)xbase;
      dMA = nx.MA - MA;
      nx_nkr = nkr;
      have_next = true;
    }
  } else if (layer + 1 < nlayers) {
    nx_dev = a.chain[layer + 1];
    dWb = (const char *)nx_dev.w1 - (const char *)a.w1;
    dK = 32;
    nx_nkr = (K + 32 + BK - 1) / BK;
    have_next = true;         // same frame: its first three stages (channels < 3 BK <= 192) exist since the block's first layer
  }
  const char *const xbase_c = (const char *)xbase, *const w1_c = (const char *)a.w1;

  // Stage g of the X stream / the weight stream, counted THROUGH the end of this tile: g < nk is this tile's stage g,
  // nk <= g is stage g - nk of what comes next (or, without a successor, a request every lane sends to the first
  // bytes of the tile: the queue keeps its static shape, the registers are tied off after phase B).
  auto load_x1 = [&](f16x8 &dst0, f16x8 &dst1, int g, int mi) {
    const bool cur = g < nk;
    const int q = cur ? g : g - nk;
    const bool real = (cur & (q < nkr)) | (!cur & have_next & (q < nx_nkr));    // (stages past the channels: dummy requests)
    const void *base = uniform_ptr(xbase_c + (cur ? 0L : dXb) + (real ? (long)q * (BK * 2) : 0L));
    const int row = mrow0 + mi * 16 + frow, ma = MA + (cur ? 0 : dMA);
    const int m = row < ma ? row : ma - 1;       // rows past the tile re-read its last row (cache hits, results unused)
    const unsigned vo = real ? (unsigned)((m * ldc + fch * 8) * 2) : 0u;
    gload16<0>(dst0, vo, base);
    if constexpr (KS == 2) gload16<64>(dst1, vo, base);
  };
  auto load_x = [&](int g, auto st) {
    constexpr int s = decltype(st)::value;
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) load_x1(xq[s][0][mi], xq[s][KS - 1][mi], g, mi);
  };
  // 1x1-weight stage ([128][kv] fp16) -> registers wr[s]: thread t carries KS 16-byte pieces
  auto load_w = [&](int g) {
    const bool cur = g < nk;
    const int q = cur ? g : g - nk;
    const bool real = (cur & (q < nkr)) | (!cur & have_next & (q < nx_nkr));
    const int kv = K + (cur ? 0 : dK);
    const void *base = uniform_ptr(w1_c + (cur ? 0L : dWb) + (real ? (long)q * (BK * 2) : 0L));
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int c = t + 512 * j, row = c / CPR, p = c % CPR;
      gload16<0>(wr[j], real ? (unsigned)((row * kv + p * 8) * 2) : 0u, base);
    }
  };
  auto write_w = [&](unsigned char *slot) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int c = t + 512 * j, row = c / CPR, p = c % CPR;
      *(f16x8 *)(slot + row * ROWB + (wswz<BK>(row, p) << 4)) = wr[j];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  if (!primed) {
    // cold start (first tile of the launch): the queue's first three entries X0 W0 X1, then the BN tables
    // (ordinary loads: their wait drains the three, once per launch)
    ph = 0;
    load_x(0, I0{});
    load_w(0);
    load_x(1, I1{});
    for (int i = t; i < K; i += 512) {
      tab1[i] = a.s1[i];
      tab1[1024 + i] = a.t1[i];
    }
    if (t < 128) {
      tab2[t] = a.s2[t];
      tab2[128 + t] = a.t2[t];
    }
    vm_wait<XL>(wr[0], dbg_drain);      // W0 has landed (X1 may still be in flight)
    if constexpr (KS == 2) vm_tie(wr[1]);
    write_w(smem + G::W3RING);
  }
  primed = false;
  const char *w3b = (const char *)a.w3p + (size_t)2 * 72 * 64 * 8 * 2;   // third packed layout (api.hip pack_conv3x3)
  const unsigned w3off = (unsigned)t * 16u;
  f16x8 wq[3];
  auto load_q = [&](f16x8 &dst, int tap) { gload16<0>(dst, w3off, uniform_ptr(w3b + tap * 8192)); };

  // ======================= phase A: bottleneck = conv1x1(relu(bn1(x))) =======================
  f32x4 acc[NI][MIW];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  DL2_STAMP(1);
  // stage kt (ring slot s = (kt + ph) % 2): wait for X(kt) -> barrier (W(kt) visible, the other weight slot free) ->
  // request W(kt+1) -> per pixel fragment: BN1+ReLU, refill its ring register with X(kt+2), 8 MFMAs -> wait for
  // W(kt+1), ds_write it
  auto kstage = [&](int kt, auto st) {
    constexpr int s = decltype(st)::value;
    // X(kt) has landed: W(kt) X(kt+1) are younger
    vm_wait<GRP>(xq[s][0][0], dbg_drain);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi)
        if (ks + mi > 0) vm_tie(xq[s][ks][mi]);
    lds_barrier();
    load_w(kt + 1);
    const unsigned char *Ws = smem + ((kt & 1) ? G::WSLOT_B : G::W3RING);
    f16x8 xb[KS][MIW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k0 = kt * BK + ks * 32;
      // dead: past the channels.  Only the second half of the last stage when K % 64 == 32 (the launcher sends
      // layers with fewer than DX stages to the first-generation kernel): the first half is straight-line code
      const bool live = (KS == 1 || ks == 0) ? true : k0 < K;
      f16x8 wa[NI];
      float sc[8], sh[8];
      if (live) {
        const int kb = k0 + fch * 8;
        const float4 sa = *(const float4 *)(tab1 + kb), sb = *(const float4 *)(tab1 + kb + 4);
        const float4 ta_ = *(const float4 *)(tab1 + 1024 + kb), tb = *(const float4 *)(tab1 + 1024 + kb + 4);
        sc[0] = sa.x; sc[1] = sa.y; sc[2] = sa.z; sc[3] = sa.w; sc[4] = sb.x; sc[5] = sb.y; sc[6] = sb.z; sc[7] = sb.w;
        sh[0] = ta_.x; sh[1] = ta_.y; sh[2] = ta_.z; sh[3] = ta_.w; sh[4] = tb.x; sh[5] = tb.y; sh[6] = tb.z; sh[7] = tb.w;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int row = nch0 + ni * 16 + frow;
          wa[ni] = *(const f16x8 *)(Ws + row * ROWB + (wswz<BK>(row, ks * 4 + fch) << 4));
        }
      }
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        // BN1 + ReLU (each element exactly once: the wave owns all 128 channels of its rows)
        if (live) xb[ks][mi] = bn_relu8_mix(xq[s][ks][mi], sc, sh);
        // the ring register is free again: refill it with stage kt + 2 (KS == 2: both halves of fragment mi go out
        // together behind the second half's BN - queue order mi-major)
        if (ks == KS - 1) load_x1(xq[s][0][mi], xq[s][KS - 1][mi], kt + DX, mi);
        // wave-uniform: fragments past the tile's rows are skipped - where that saves real work (28x28: 28 of 32
        // fragments, 14x14: 13 of 16); at 56x56 (504 of 512 rows) the straight-line code is worth more
        constexpr bool SKIP = (TR * W + 15) / 16 < BM / 16;
        if (live & (!SKIP | ((mrow0 + mi * 16 < MA) & (mi < nfw)))) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb[ks][mi], acc[ni][mi], 0, 0, 0);
        }
      }
    }
    // W(kt+1) has landed: X(kt+2) is younger
    vm_wait<XL>(wr[0], dbg_drain);
    if constexpr (KS == 2) vm_tie(wr[1]);
    if (kt + 1 < nk) write_w(smem + (((kt + 1) & 1) ? G::WSLOT_B : G::W3RING));
  };
  for (int v = 0; v < ph + nk; v += 2) {      // v = kt + ph: the slot is v % 2
    if (v >= ph && v < ph + nk) kstage(v - ph, I0{});
    if (v + 1 >= ph && v + 1 < ph + nk) kstage(v + 1 - ph, I1{});
  }
  // the ring now holds X(0), X(1) of what comes next in slots (ph + nk + 0..1) % 2, its W(0) in wr
  ph = (ph + nk) & 1;
  load_q(wq[0], 0);   // the first three taps of the 3x3 weights: their latency hides behind epilogue A
  load_q(wq[1], 1);
  load_q(wq[2], 2);
  lds_barrier();   // every wave is done reading the weight slots; the tile may now be written

  DL2_STAMP(2);
  // ---- zero padding of the tile: the two pad columns of every row, out-of-image halo rows ----
  {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    if (t < TR * 32) {
      const int tr = t >> 5, side = (t >> 4) & 1, ch = t & 15;
      *(uint4 *)(tile + (tr * WP + side * (WP - 1)) * 256 + ch * 16) = z4;
    }
    if (top_pad)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + idx * 16) = z4;
    if (r0 + ROUT >= H)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + (TR - 1) * WP * 256 + idx * 16) = z4;
  }
  // ---- epilogue A: BN2 + ReLU, fp16, scatter into the tile ----
  // D[i=n][j=m]: lane holds channels n = ni*16 + fch*4 + r of pixel row m = .. + frow
  {
    unsigned char *dst[MIW];
    int sl15[MIW];
    bool ok[MIW];
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
      const int m = mrow0 + mi * 16 + frow;
      const int rr = m / W, x = m - rr * W;
      const int slot = (rr + top_pad) * WP + x + 1;
      dst[mi] = tile + slot * 256 + (fch & 1) * 8;
      sl15[mi] = slot & 15;
      ok[mi] = m < MA && mi < nfw;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float4 sv = *(const float4 *)(tab2 + nch0 + ni * 16 + fch * 4);
      const float4 tv = *(const float4 *)(tab2 + 128 + nch0 + ni * 16 + fch * 4);
      const int chunk = (nch0 >> 3) + ni * 2 + (fch >> 1);
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        const f16x4 hv = bn_relu4_from_f32(acc[ni][mi], sv, tv);
        if (ok[mi]) *(f16x4 *)(dst[mi] + ((chunk ^ sl15[mi]) << 4)) = hv;
      }
    }
  }
  vm_wait<2>(wq[0], dbg_drain);        // q0 has landed (q1 q2 younger)
  *(f16x8 *)(ring + t * 16) = wq[0];   // tap 0 -> ring[0]
  load_q(wq[0], 3);                    // request tap 3
  lds_barrier();
  DL2_STAMP(3);

  // ======================= phase B: y = conv3x3(tile) ========================================
  // 16-slot output fragments, v_mfma_f32_16x16x32_f16 with the packed weights as the A operand; a wave owns whole
  // fragments over the full K = 9 x 128; the packed 3x3 weights stream registers -> LDS through the 2 x 8 KiB ring,
  // one tap per barrier (queue during phase B: q1 q2 q3 | q4 | q5 | ... : two younger requests behind each needed tap).
  constexpr int NF16 = G::NF16, MAXU = G::MAXU;
  const int wpos = (wid & 3) * 2 + (wid >> 2);
  const int u0 = (wpos * NF16) >> 3, u1 = ((wpos + 1) * NF16) >> 3;   // this wave's fragments [u0, u1)
  const int px = lane & 15, kg = lane >> 4;
  f32x4 bacc[MAXU > 0 ? MAXU : 1][2];
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
        if (tap == 0) vm_wait<2>(wq[1], dbg_drain);
        if (tap == 1) vm_wait<2>(wq[2], dbg_drain);
        if (tap == 2) vm_wait<2>(wq[0], dbg_drain);
        if (tap == 3) vm_wait<2>(wq[1], dbg_drain);
        if (tap == 4) vm_wait<2>(wq[2], dbg_drain);
        if (tap == 5) vm_wait<2>(wq[0], dbg_drain);
        if (tap == 6) vm_wait<1>(wq[1], dbg_drain);
        if (tap == 7) vm_wait<0>(wq[2], dbg_drain);
        *(f16x8 *)(ring + ((tap + 1) & 1) * 8192 + t * 16) = wq[(tap + 1) % 3];
        if (tap + 4 < 9) load_q(wq[(tap + 1) % 3], tap + 4);
      }
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int off = WP + dy * WP + dx + px + 16 * u0;
      const unsigned char *wring = ring + (tap & 1) * 8192 + lane * 16;
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
      lds_barrier();
    }
  };
  if (u1 - u0 == MAXU) phase_b(std::integral_constant<int, MAXU>{});
  else phase_b(std::integral_constant<int, MAXU - 1>{});
  DL2_STAMP(4);

  // ---- tap 7 waited for everything in flight, the ring included.  Tie its registers here: this is where the requests
  // without a consumer (no next tile) end, and no register of the ring may be handed to another value before. ----
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int s = 0; s < DX; ++s)
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) vm_tie(xq[s][ks][mi]);
    vm_tie(wr[ks]);
  }
  // ---- the tile and the 3x3 ring are dead: W(0) of what comes next goes to weight slot A, the next layer's tables to LDS ----
  if (have_next) {
    write_w(smem + G::W3RING);
    primed = true;
  }
  if constexpr (CHAIN) {
    if (have_next) {
      for (int i = t; i < K + 32; i += 512) {
        tab1[i] = nx_dev.s1[i];
        tab1[1024 + i] = nx_dev.t1[i];
      }
      if (t < 128) {
        tab2[t] = nx_dev.s2[t];
        tab2[128 + t] = nx_dev.t2[t];
      }
    }
  }
  // ---- output: lane (px, kg) holds channels 8 kg .. 8 kg + 7 of its pixel: one 16-byte store (the concat) ----
  {
    f16 *ybase = a.buf + ((long)img * H * W + (long)r0 * W) * ldc + K + kg * 8;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      if (u0 + j < u1) {
        const int srel = 16 * (u0 + j) + px;           // slot relative to the first output slot: r * WP + x + 1
        const int r = srel / WP, x = srel - r * WP - 1;
        if (x >= 0 && x < W && r < ROUT) {
          f16x8 hv;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hv[q] = (f16)bacc[j][0][q];
            hv[4 + q] = (f16)bacc[j][1][q];
          }
          store16_hidden(ybase + (long)(r * W + x) * ldc, hv);
        }
      }
    }
  }
  DL2_STAMP(5);
  if constexpr (CHAIN) {
    // this layer's stores have completed (stores count in vmcnt on gfx9) for every wave before the next layer
    // loads them as its newest channels
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  }   // layer
  }   // tile
}

template <int W, int ROUT, int BM, int KS, bool CHAIN = false>
int launch_geom2(const DenseLayerArgs &a, hipStream_t s) {
  using G = DL2Geom<W, ROUT, BM, KS>;
  static bool attr_set = false;
  if (!attr_set) {
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_layer2_kernel<W, ROUT, BM, KS, CHAIN>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    TN_HIP_CHECK(hipGetDevice(&dev));
    TN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
    if (getenv("TN_PERSIST_WGS")) ncu = atoi(getenv("TN_PERSIST_WGS"));   // tuning hook
  }
  const int nvb = a.B * (a.H / ROUT);
  const bool persist = !CHAIN && !(a.variant & 16) && nvb > ncu && ncu > 0;
  const dim3 grid(persist ? ncu : nvb), block(512);
  hipLaunchKernelGGL((dense_layer2_kernel<W, ROUT, BM, KS, CHAIN>), grid, block, G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

}  // namespace

// the X ring runs through one tile boundary: a K loop needs at least two stages (of 32 channels at 56x56 / 28x28, 64 below)
bool dense_layer_v2_supported(int H, int W, int K) {
  if (H != W) return false;
  if (H == 56 || H == 28) return K >= 64;
  if (H == 14 || H == 7) return K >= 128;
  return false;
}

int launch_dense_layer_v2(const DenseLayerArgs &a, hipStream_t s) {
  const int klast = a.K + 32 * (a.nchain > 0 ? a.nchain - 1 : 0);
  TN_REQUIRE(a.K % 32 == 0 && klast <= 1024 && a.ldc % 8 == 0 && klast + 32 <= a.ldc, "dense_layer: bad channel geometry");
  TN_REQUIRE(dense_layer_v2_supported(a.H, a.W, a.K), "dense_layer_v2: K loop shorter than the register ring");
  if (a.nchain > 0) {
    TN_REQUIRE(a.H == a.W && (a.H == 14 || a.H == 7) && a.chain, "dense_layer: layer chaining needs whole-frame tiles (14x14, 7x7)");
    if (a.H == 14) return launch_geom2<14, 14, 256, 2, true>(a, s);
    return launch_geom2<7, 7, 64, 2, true>(a, s);
  }
  if (a.H == 56 && a.W == 56) return launch_geom2<56, 7, 512, 1>(a, s);
  if (a.H == 28 && a.W == 28) return launch_geom2<28, 14, 512, 1>(a, s);
  if (a.H == 14 && a.W == 14) return launch_geom2<14, 14, 256, 2>(a, s);
  if (a.H == 7 && a.W == 7) return launch_geom2<7, 7, 64, 2>(a, s);
  TN_REQUIRE(false, "dense_layer: unsupported spatial size");
}

// Fused DenseNet dense layer (SURVEY §7 H1, §2c rows K2+K3+K4):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// one launch per layer, replacing gluoncv's BatchNorm-Activation-Conv1x1-BatchNorm-
// Activation-Conv3x3-Concat chain (reference call site models/vision/definitions.py:30).
// The 128-channel bottleneck never goes to HBM.  A workgroup (8 waves) owns ROUT full
// image rows and computes the bottleneck of those rows plus one halo row above/below.
//
// Phase A (bottleneck GEMM, M = (ROUT+2)*W rows, N = 128, K): raw activations and the
//   1x1 weights stream global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round
//   trip) through a 3-stage ring with counted vmcnt waits and ONE raw s_barrier per
//   k-tile, so two stages are always in flight behind the MFMAs.  The DMA destination is
//   lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//   and again on the fragment read.  BN1+ReLU (fp32 math, one rounding) is applied to the
//   pixel fragment after the ds_read; each wave owns all 128 output channels of its rows,
//   so every element is transformed exactly once.  v_mfma_f32_16x16x32_f16 with the
//   weight fragment as the A operand: a lane ends with 4 consecutive channels of a pixel.
// Epilogue A: BN2+ReLU in fp32, fp16, scatter into an LDS tile of (ROUT+2) x (W+2) pixel
//   slots of 256 B with the zero padding materialised (the conv pads after the activation).
// Phase B: the 3x3 is a constant-offset walk over the flattened tile with
//   v_mfma_f32_32x32x16_f16 — no bounds logic; K = 1152 is split by channel halves over
//   wave pairs (partials meet in LDS); the packed 3x3 weights stream through a 2 x 8 KiB
//   LDS ring, one tap per barrier, three taps of prefetch in registers.
// Writing the 32 output channels at channel offset K of the same buffer IS the concat.
#include <type_traits>

#include "common.h"

#ifndef TN_EXP
#define TN_EXP 0   // timing experiments only (scripts/kbench.py --lib): bit 0 skip phase B, 1 skip K loop, 2 skip epilogue A, 3 skip the store (note: skipping a consumer lets the compiler drop its producer's MFMAs too), 4 K loop streams only (no LDS reads / MFMA), 5 no refill inside the K loop (compute on whatever the ring holds)
#endif

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int W, int ROUT, int BM, int BK>
struct DLGeom {
  static constexpr int WP = W + 2;
  static constexpr int TR = ROUT + 2;
  static constexpr int NSLOT = TR * WP;
  static constexpr int NF = (ROUT * WP + 31) / 32;           // 32-slot output fragments
  static constexpr int NF16 = (ROUT * WP + 15) / 16;         // 16-slot output fragments of phase B
  static constexpr int MAXU = (NF16 + 7) / 8;                // ... per wave
  static constexpr int RSLOT = WP + 32 * NF + WP + 2;        // highest slot phase B touches + 1
  static constexpr int TSLOT = NSLOT > RSLOT ? NSLOT : RSLOT;
  static constexpr int TILE_BYTES = TSLOT * 256;
  static constexpr int ROWB = BK * 2;                        // bytes per staged row
  static constexpr int XS = BM * ROWB, WS = 128 * ROWB, STAGE = XS + WS;
  static constexpr int NST = (BM == 64) ? 4 : 3;             // ring slots (7x7: four, for the software-pipelined K loop)
  static constexpr int PIECES = STAGE / 1024, PPW = PIECES / 8, XPIECES = XS / 1024;
  static constexpr int RPP = 1024 / ROWB;                    // rows per 1-KiB DMA piece
  static constexpr int RING_A = NST * STAGE;
  static constexpr int KMAX = W == 56 ? 256 : W == 28 ? 512 : 1024;   // input channels a layer of this block can have
  static constexpr int TABS = 1024 + 8 * KMAX;               // s2[128] t2[128] | s1[KMAX] t1[KMAX]
  static constexpr int W3RING = TILE_BYTES;                  // slots 0 and 1 of the 3 x 8 KiB ring of 3x3 weights
  static constexpr int W3SLOT2 = TILE_BYTES + 16384;         // slot 2
  // The BatchNorm tables.  56x56 / 28x28: in the slots between the tile's last row and the highest slot a (discarded)
  // phase-B fragment still reads - nothing is ever written there, the K-loop ring ends below them, and it is what lets the
  // third weight slot fit into 160 KB.  14x14 / 7x7 (K-loop ring longer than tile + weight slots): behind the ring.
  static constexpr bool TABS_IN_SLACK = (TSLOT - NSLOT) * 256 >= TABS && NSLOT * 256 >= RING_A;
  static constexpr int TAIL = (TILE_BYTES + 24576 > RING_A) ? TILE_BYTES + 24576 : RING_A;
  // a pixel slot that takes the stores of rows past the tile: the last slot phase B can touch when the tables are not in the
  // slack (only discarded fragments read it), else one extra slot's worth of bytes behind everything
  static constexpr int DUMP_SLOT = TABS_IN_SLACK ? (TAIL + 255) / 256 : TSLOT - 1;
  static constexpr int TAB2 = TABS_IN_SLACK ? NSLOT * 256 : TAIL;
  static constexpr int TAB1 = TAB2 + 1024;
  static constexpr int LDS_BYTES = TABS_IN_SLACK ? (DUMP_SLOT + 1) * 256 : TAIL + TABS;
  // BM = 64 (7x7 frames, 63 tile rows): the waves split the tile 4 (pixel rows) x 2 (bottleneck channel halves)
  // instead of 8 x 1, so no wave spends MFMAs and weight-fragment reads on rows past the frame
  static constexpr bool NSPLIT = (BM == 64);
  static constexpr int NI = NSPLIT ? 4 : 8;                  // 16-channel weight fragments per wave
  static constexpr int MIW = NSPLIT ? 1 : BM / 128;          // 16-row pixel fragments per wave
  static_assert(PIECES % 8 == 0, "DMA pieces must divide over 8 waves");
  static_assert(DUMP_SLOT >= NSLOT, "the dump slot must not be a pixel or padding slot of the tile");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
  static_assert(16 * NF16 * 80 <= TILE_BYTES, "output row buffer does not fit");
  static_assert(BM >= TR * W && (BM % 128 == 0 || BM == 64), "phase A tile too small");
  static_assert(BK == 32 || BK == 64, "BK");
};

// swizzle of the 16-B chunk index inside a staged row (conflict-free ds_read_b128 of a
// 16x32 MFMA operand fragment): 128-B rows: chunk ^ (row & 7); 64-B rows: chunk ^ g(row>>2)
template <int BK>
__device__ __forceinline__ int stage_swz(int row, int chunk) {
  if constexpr (BK == 64) return chunk ^ (row & 7);
  else return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

// swizzle of the 16-B chunk (8 bottleneck channels) inside a 256-B pixel slot of the tile: chunk ^ ((slot & 7) << 1).
// A phase-B fragment read is 16 consecutive slots x one chunk per 16-lane k-group, and gfx950 serves a ds_read_b128 in
// the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... - eight lanes of k-group g and eight of g+1, on
// complementary halves of the slot run.  With chunk ^ (slot & 15) (round 1) a run that starts on an odd slot - every
// tap with dx = +-1 - put two lane pairs of each group on the same banks (2-way: 8 LDS cycles per read instead of 4,
// SQ_LDS_BANK_CONFLICT = 27 % of SQ_LDS_IDX_ACTIVE).  Leaving bit 0 of the chunk alone and folding slot bits 0-2 into
// chunk bits 1-3 is conflict-free for every start slot, k-step and lane group (exhaustive search over the GF(2)-linear
// maps, scripts/lds_swizzle_search.py); the price is paid by epilogue A's ds_write_b64: a 16-lane store group then covers
// only four of the eight chunk positions.
// ALL16 = the first form, `chunk ^ (slot & 15)`: 2-way on the epilogue's stores (153 B/ns/CU in scripts/microbench/ldsstore.hip
// against 76 for the second form, 195 for contiguous stores) but two lane pairs per group collide in the dx = +-1 taps.
// Measured per geometry (r2-g, phase stamps and rocprof on one box): at 56x56 the first form wins (epilogue A 5 270 -> 3 970
// cycles per tile, phase B 11 310 -> 12 160, kernel -1.7 %) and at 28x28 too (-1.1 %); the chained 14x14 block is 0.6 %
// faster with the second form, the 7x7 block does not care.
template <bool ALL16>
__device__ __forceinline__ int tile_swz(int slot) { return ALL16 ? (slot & 15) : ((slot & 7) << 1); }

// One LDS-DMA piece: 64 lanes x 16 B, global (per-lane address) -> LDS (wave-uniform base
// + lane*16).  Issued through inline asm on purpose: hipcc treats the builtin as an LDS
// store it must order against every later ds_read and inserts s_waitcnt vmcnt(0) in front
// of the fragment reads, which drains the whole ring each k-tile.  Hidden in asm, the only
// waits are the counted ones below (cdna_hip_programming.md 5.7; M0 saved/restored).
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
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

__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// CHAIN (whole-frame tiles only): the workgroup runs a.nchain consecutive layers of the block on its frame
// inside one launch; layer l reads channels [0, K0 + 32 l) -- its own earlier outputs included, which the
// same CU's L1 sees coherently once the stores have been waited for -- with parameters from a.chain[l].
// EX (exact-weights mode): every weight is the sum of two fp16 numbers, w = hi + lo (lo = fp16(w - hi):
// 22 bits of the fp32 weight).  The 1x1 weights come as [128][2 Kp] = [hi | lo] (Kp = K rounded up to BK) and the K
// loop simply runs over 2 Kp "channels", the activation side wrapping around after the hi half; the 3x3 weights
// come as a second packed image and phase B walks 18 taps.  Activations, BatchNorm and accumulation are unchanged.
template <int W, int ROUT, int BM, int BK, int PP, bool CHAIN, bool EX = false>
__global__ __launch_bounds__(512) void dense_layer_kernel(DenseLayerArgs a) {
  static_assert(!EX || PP == 2, "the exact-weights mode exists for the default K loop only");
  using G = DLGeom<W, ROUT, BM, BK>;
  constexpr int WP = G::WP, TR = G::TR, MIW = G::MIW, NI = G::NI;
  static_assert(PP != 4 || (G::NSPLIT && BK == 64), "the software-pipelined K loop is built for the 7x7 geometry");
  constexpr int ROWB = G::ROWB, PPW = G::PPW, RPP = G::RPP, CPR = ROWB / 16;  // chunks per row
#ifdef TN_SWZ16_ALL
  constexpr bool SWZ16 = TN_SWZ16_ALL != 0;
#else
  constexpr bool SWZ16 = W >= 28;                // tile swizzle form, see tile_swz
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem;                   // bottleneck tile (aliases the DMA ring)
  unsigned char *ring = smem + G::W3RING;       // 3x3 weight ring: slots 0, 1 here, slot 2 at W3SLOT2
  auto w3slot = [&](int i) { return i == 2 ? smem + G::W3SLOT2 : ring + i * 8192; };
  float *tab2 = (float *)(smem + G::TAB2);
  float *tab1 = (float *)(smem + G::TAB1);

  const int t = threadIdx.x;
  const int H = a.H, ldc = a.ldc;
#define DL_STAMP(i) do { if (a.ts && t == 0) a.ts[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  DL_STAMP(0);
  // all row-tiles of a frame share (virtual) blockIdx % 8 (the XCD) when B % 8 == 0, so the halo rows two
  // neighbouring tiles both read are L2 hits on that XCD instead of second HBM fetches
  const int tiles_per_img = H / ROUT;
  struct TileAt { int img, r0, MA, top_pad; const f16 *xbase; };
  auto tile_at = [&](int vb) {
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
    ta.r0 = tix * ROUT;                                             // first output row
    const int rlo = ta.r0 > 0 ? ta.r0 - 1 : 0;                      // first computed bottleneck row
    const int rhi = (ta.r0 + ROUT < H) ? ta.r0 + ROUT + 1 : H;      // one past the last
    ta.MA = (rhi - rlo) * W;                                        // real phase-A rows
    ta.top_pad = (ta.r0 == 0) ? 1 : 0;                              // tile row 0 lies above the image
    ta.xbase = a.buf + ((long)img * H * W + (long)rlo * W) * ldc;
    return ta;
  };
  static_assert(!CHAIN || ROUT == W, "layer chaining needs whole-frame tiles");
  // Persistent tiles (flat K loops, one layer per launch): the grid is one workgroup per CU and each walks
  // the tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ...  Right after a tile's phase B (ring region free
  // again) it already requests the first two stages of its NEXT tile, so that tile's cold-start latency and
  // this tile's store drain overlap instead of adding up once per tile.
  const int nvb = a.B * tiles_per_img;
  bool primed = false;      // the first two stages (and the tables) of the next tile / layer have already been requested
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  const TileAt ta = tile_at(vb);
  const int img = ta.img, r0 = ta.r0, MA = ta.MA, top_pad = ta.top_pad;
  const f16 *xbase = ta.xbase;
  const int nlayers = CHAIN ? a.nchain : 1;
  const int K0 = a.K;
  for (int layer = 0; layer < nlayers; ++layer) {
  // per-layer copies of the thread coordinates, laundered so that the compiler does not hoist every
  // address computation of the body out of the tile / layer loops (that costs ~80 VGPRs and spills)
  int t_ = threadIdx.x;
  asm volatile("" : "+v"(t_));
  const int t = t_;
  const int lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  if constexpr (CHAIN) {
    const DenseLayerDev d = a.chain[layer];
    a.K = K0 + 32 * layer;
    a.s1 = d.s1; a.t1 = d.t1; a.w1 = d.w1; a.s2 = d.s2; a.t2 = d.t2; a.w3p = d.w3p;
  }
  const int K = a.K;
  // ======================= phase A: bottleneck = conv1x1(relu(bn1(x))) =======================
  const int frow = lane & 15, fch = lane >> 4;
  // REBAL (28x28: 28 real fragments in a 32-fragment tile): waves 0-3 own 4 fragments, waves 4-7 own 3, so every
  // SIMD (waves w, w+4) carries 7 instead of 8 / 8 / 8 / 4
  const bool REBAL = !G::NSPLIT && (PP == 0 || PP == 2) && MIW == 4 && (TR * W + 15) / 16 == 28 && !(a.variant & 128);   // bit 7: off, for A/B runs
  const int mrow0 = G::NSPLIT ? (wid & 3) * 16 : REBAL ? (wid < 4 ? wid * 64 : 256 + (wid - 4) * 48) : wid * (BM / 8);   // first tile row of this wave
  const int nfw = (REBAL && wid >= 4) ? 3 : MIW;                    // fragments this wave owns
  const int nch0 = G::NSPLIT ? (wid >> 2) * 64 : 0;                // first bottleneck channel of this wave
  f32x4 acc[NI][MIW];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);   // LDS byte address of smem
  const int nkc = (K + BK - 1) / BK;          // k-tiles that carry channels
  const int nk = EX ? 2 * nkc : nkc;          // k-tiles of the loop (EX: hi pass, then lo pass over the same activations)
  const int w1ld = EX ? 2 * nkc * BK : K;     // row pitch of the 1x1 weights
  const f16x8 *w3 = (const f16x8 *)a.w3p + 72 * 64 + t;   // second half of the packed buffer: the 16x16x32 layout
  // packed 3x3 fragments of tap i (EX: taps 9..17 are the lo image, two more layouts further on)
  auto w3tap = [&](int i) { return (EX && i >= 9) ? w3[2 * 72 * 64 + (i - 9) * 512] : w3[i * 512]; };
  f16x8 wq[3];
  // flat K loops: per-lane DMA source pointers of this wave's PPW pieces (advance by BK halfs per k-tile)
  const f16 *src[PPW];
  // Which 1-KiB pieces of a stage a wave loads.  OWNX (every geometry but the 4 x 2 wave split of 7x7): first the pieces
  // that hold the wave's OWN pixel rows, then its share of the weight pieces - a wave's fragment reads of the activations
  // then depend on its own DMA only (its counted vmcnt orders them, no barrier), and the slot of those rows is refilled by
  // the wave that read them.  At 28x28 (REBAL) waves 4-7 own three pieces each and take one of the four pieces past the
  // tile's rows as their fourth.
  constexpr bool OWNX = !G::NSPLIT && PP != 4 && BK == 32 && G::XPIECES % 8 == 0 && (G::XPIECES / 8) * RPP == BM / 8;   // (14x14, 64-channel stages: measured 2 % slower)
  constexpr int XPW = G::XPIECES / 8;
  auto piece_of = [&](int j) {
    if constexpr (!OWNX) return wid * PPW + j;
    else {
      if (j >= XPW) return G::XPIECES + wid * (PPW - XPW) + (j - XPW);
      if (REBAL) return wid < 4 ? 4 * wid + j : (j < 3 ? 16 + 3 * (wid - 4) + j : 28 + (wid - 4));
      return wid * XPW + j;
    }
  };
  auto is_x = [&](int j) { return OWNX ? j < XPW : wid * PPW + j < G::XPIECES; };
  auto set_src = [&](const f16 *xb, int ma, const f16 *w1p, int kv) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = piece_of(j);                   // wave-uniform
      const int prow = lane / CPR, p = lane % CPR;     // row inside the piece, linear chunk position
      if (piece < G::XPIECES) {
        const int row = piece * RPP + prow;
        const int m = row < ma ? row : ma - 1;
        src[j] = xb + (long)m * ldc + stage_swz<BK>(row, p) * 8;
      } else {
        const int row = (piece - G::XPIECES) * RPP + prow;
        src[j] = w1p + (long)row * kv + stage_swz<BK>(row, p) * 8;
      }
    }
  };
  // after k-tile sidx went out the source pointers move on by BK; EX: the activation pieces wrap around to channel 0
  // behind the last k-tile of the hi pass (the weight pieces walk on into the lo half of their rows)
  auto advance = [&](int j, int sidx) {
    src[j] += BK;
    if constexpr (EX) {
      if (sidx == nkc - 1 && is_x(j)) src[j] -= nkc * BK;
    }
  };
  auto issue_pieces = [&](int st, int sidx, auto j0t, auto j1t) {
#pragma unroll
    for (int j = decltype(j0t)::value; j < decltype(j1t)::value; ++j) {
      dma16(src[j], lds0 + st * G::STAGE + piece_of(j) * 1024);
      advance(j, sidx);
    }
  };
  auto issue = [&](int st, int sidx) { issue_pieces(st, sidx, std::integral_constant<int, 0>{}, std::integral_constant<int, PPW>{}); };
  if constexpr (PP == 4) {
  // ---------------------------------------------------------------------------------------------------------
  // Software-pipelined K loop of the 7x7 geometry (64-channel stages).  The flat loop walks, once per k-step and
  // behind the stage's barrier, a chain LDS reads -> BN+ReLU -> MFMAs with 4 MFMAs per wave at its end (measured:
  // 1 800 cycles per stage for 128 cycles of MFMA work).  Here the operands of k-step j+1 are read while k-step j
  // computes, THROUGH the stage boundary: the barrier that publishes stage kt+1 sits in the middle of stage kt (after
  // its first k-step), and the reads of (kt+1, ks 0) go out before (kt, ks 1) computes.  Four ring slots, stage kt in
  // slot (1 + kt) % 4: stage kt+3 is requested at that barrier into the slot of stage kt-1 and has two stages of time
  // to land.  Measured per 64-channel stage (K = 768): 1 810 cycles flat, 1 150 this way; with the request going to
  // the slot of stage kt instead (three slots; every wave holds (kt, ks 1) in registers by then) 1 450, with five
  // slots and three stages in flight 1 470; the same loop at 14x14 (8 + 2 fragments per k-step) is 14 % SLOWER than
  // the flat one there - its K loop is bound by the activation stream, not by the operand chain.
  // ---------------------------------------------------------------------------------------------------------
  struct Opnd { f16x8 wa[NI]; f16x8 x[MIW]; float sc[8], sh[8]; };
  auto read_opnd = [&](Opnd &o, int kt, int ks) {
    const unsigned char *Xs = smem + ((1 + kt) & 3) * G::STAGE;
    const unsigned char *Ws = Xs + G::XS;
    const int kb = kt * BK + ks * 32 + fch * 8;
    const float4 s0 = *(const float4 *)(tab1 + kb), s1 = *(const float4 *)(tab1 + kb + 4);
    const float4 t0 = *(const float4 *)(tab1 + G::KMAX + kb), t1 = *(const float4 *)(tab1 + G::KMAX + kb + 4);
    o.sc[0] = s0.x; o.sc[1] = s0.y; o.sc[2] = s0.z; o.sc[3] = s0.w; o.sc[4] = s1.x; o.sc[5] = s1.y; o.sc[6] = s1.z; o.sc[7] = s1.w;
    o.sh[0] = t0.x; o.sh[1] = t0.y; o.sh[2] = t0.z; o.sh[3] = t0.w; o.sh[4] = t1.x; o.sh[5] = t1.y; o.sh[6] = t1.z; o.sh[7] = t1.w;
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
      const int xrow = mrow0 + mi * 16 + frow;
      o.x[mi] = *(const f16x8 *)(Xs + xrow * ROWB + (stage_swz<BK>(xrow, ks * 4 + fch) << 4));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = nch0 + ni * 16 + frow;
      o.wa[ni] = *(const f16x8 *)(Ws + row * ROWB + (stage_swz<BK>(row, ks * 4 + fch) << 4));
    }
  };
  auto compute = [&](const Opnd &o) {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
      if (mrow0 + mi * 16 < MA) {                  // wave-uniform: fragments past the frame's rows are skipped
        const f16x8 xb = clamp8_pk(o.x[mi], o.sc, o.sh);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.wa[ni], xb, acc[ni][mi], 0, 0, 0);
      }
    }
  };
  auto issue4 = [&](int kstage) {      // k-tile kstage -> slot (1 + kstage) % 4
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      dma16(src[j], lds0 + ((1 + kstage) & 3) * G::STAGE + (wid * PPW + j) * 1024);
      src[j] += BK;
    }
  };
  if (!primed) {
    set_src(xbase, MA, a.w1, K);
    issue4(0);
    if (nk > 1) issue4(1);
    if (nk > 2) issue4(2);
    for (int i = t; i < K; i += 512) {
      tab1[i] = a.s1[i];
      tab1[G::KMAX + i] = a.t1[i];
    }
    if (t < 128) {
      tab2[t] = a.s2[t];
      tab2[128 + t] = a.t2[t];
    }
  }
  primed = false;
  wq[0] = w3[0];
  wq[1] = w3[512];
  wq[2] = w3[2 * 512];
  // stage 0 has landed; stages 1, 2 and the three taps above are younger (a cold start has drained everything at its
  // table loads already)
  if (nk > 2) wait_vmcnt<2 * PPW + 3>(); else wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // table writes of a cold start
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  DL_STAMP(1);
  Opnd o0, o1;
  read_opnd(o0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const bool live1 = kt * BK + 32 < K;          // the stage's second k-step carries channels (K % 64 == 32: not in the last one)
    if (live1) read_opnd(o1, kt, 1);
    compute(o0);
    if (kt + 1 < nk) {
      // publish stage kt+1: own pieces landed (stage kt+2 may still be in flight), then the barrier
      if (kt + 2 < nk) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();
      if (kt + 3 < nk) issue4(kt + 3);
      read_opnd(o0, kt + 1, 0);
    }
    if (live1) compute(o1);
  }
  __syncthreads();   // every wave is done reading the DMA ring; the tile may now be written
  } else {
  constexpr bool SPREAD = (PP == 2);   // refill pieces interleaved with the MFMA groups instead of up front
  if (!primed) {
    set_src(xbase, MA, a.w1, w1ld);
    issue(1, 0);            // stage kt lives in slot (1 + kt) % 3: slot 0 is where the previous tile's
    if (nk > 1) issue(2, 1);   // output row buffer sits while the next tile's first stages are requested
    // BN tables -> LDS (ordinary loads; their wait also covers the two DMA stages above)
    for (int i = t; i < K; i += 512) {
      tab1[i] = a.s1[i];
      tab1[G::KMAX + i] = a.t1[i];
    }
    if (t < 128) {
      tab2[t] = a.s2[t];
      tab2[128 + t] = a.t2[t];
    }
    __syncthreads();   // tables visible to every wave (also drains the first two DMA stages)
  }
  primed = false;

  // request the first three taps of the 3x3 weights already now (one 16-B piece per thread per
  // tap): their latency hides behind the whole K loop instead of stalling epilogue A
  wq[0] = w3[0];
  wq[1] = w3[512];
  wq[2] = w3[2 * 512];

  // Fragments this wave computes (wave-uniform): the first nf of its MIW pixel fragments - those inside the tile's rows and,
  // at 28x28, inside the wave's share.  The K loop is instantiated per count so that a stage is one straight-line block.
  int nf = 0;
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi) nf += (mrow0 + mi * 16 < MA && mi < nfw) ? 1 : 0;
  // One k-tile: wait until stage kt has landed (YOUNGER = stages issued after it that may still be in flight), barrier,
  // compute, refilling the slot everyone just finished with.  Per 32-channel k-step ALL operand reads go out first
  // (BatchNorm constants, the first pixel fragment, the 1x1 weight fragments, the other pixel fragments); then fragment by
  // fragment 8 MFMAs, with BN+ReLU of the NEXT fragment (12 VALU instructions) issued in the gaps behind the first four of
  // them and this fragment's share of the refill DMA behind the last.  Left to itself hipcc emits read -> s_waitcnt
  // lgkmcnt(0) -> 12 VALU -> 8 MFMA per fragment: five exposed LDS round trips per stage and the matrix pipe idle
  // under every BatchNorm (measured: 2 025 cycles per stage with no DMA at all, for 1 088 cycles of MFMAs).
  auto kloop = [&](auto nf_tag) {
  constexpr int NF = decltype(nf_tag)::value;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  int st = 1;
  auto ktile = [&](int kt, auto younger_tag) {
    constexpr int YOUNGER = decltype(younger_tag)::value;
    const unsigned char *Xs = smem + st * G::STAGE;
    const unsigned char *Ws = Xs + G::XS;
    auto kx = [&](int ks) { return ((EX && kt >= nkc) ? kt - nkc : kt) * BK + ks * 32; };   // first activation channel of a k-step
    float4 ps0, ps1, pt0, pt1;                      // BatchNorm constants and pixel fragments of the stage's first k-step
    u32x4 px[NF > 0 ? NF : 1];
    auto read_tab_x = [&](int ks, float4 &s0, float4 &s1, float4 &t0, float4 &t1, u32x4 *xr) {
      const int kb = kx(ks) + fch * 8;
      s0 = *(const float4 *)(tab1 + kb); s1 = *(const float4 *)(tab1 + kb + 4);
      t0 = *(const float4 *)(tab1 + G::KMAX + kb); t1 = *(const float4 *)(tab1 + G::KMAX + kb + 4);
#pragma unroll
      for (int mi = 0; mi < NF; ++mi) {
        const int row = mrow0 + mi * 16 + frow;
        xr[mi] = *(const u32x4 *)(Xs + row * ROWB + (stage_swz<BK>(row, ks * 4 + fch) << 4));
      }
    };
    wait_vmcnt<YOUNGER * PPW>();
    // OWNX: this wave's pixel rows of the stage came through its own DMA pieces, which the wait above has retired: their
    // fragments (and the constants) are requested BEFORE the barrier, so that behind it only the weight fragments are
    // waited for (with all sixteen reads of all eight waves behind the barrier the matrix pipes idle until the LDS has
    // served the lot)
    constexpr bool PRE = OWNX && NF > 0 && !(TN_EXP & 16);
    if constexpr (PRE) {
      if (kx(0) < K) read_tab_x(0, ps0, ps1, pt0, pt1, px);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt == 0) DL_STAMP(1);
    const bool refill = !(TN_EXP & 32) && kt + 2 < nk;
    const int rslot = st >= 1 ? st - 1 : 2;         // slot (kt+2)%3, free since everyone passed the barrier
    if (!SPREAD && refill) issue(rslot, kt + 2);
    constexpr int NG = (BK / 32) * MIW;             // refill groups of a stage: one per (k-step, fragment)
    auto refill_group = [&](int gi) {               // pieces [gi*PPW/NG, (gi+1)*PPW/NG) of the refill
      if constexpr (SPREAD) {
        if (refill) {
#pragma unroll
          for (int j = 0; j < PPW; ++j)
            if (j >= gi * PPW / NG && j < (gi + 1) * PPW / NG) {
              dma16(src[j], lds0 + rslot * G::STAGE + piece_of(j) * 1024);
              advance(j, kt + 2);
            }
        }
      }
    };
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      if (TN_EXP & 16) {
        if (SPREAD && refill && ks == 0) issue(rslot, kt + 2);
      } else
      if (kx(ks) < K) {
        if constexpr (NF > 0) {
          float4 s0, s1, t0, t1;
          u32x4 xraw[NF];
          f16x8 wa[NI];
          if (PRE && ks == 0) {
            s0 = ps0; s1 = ps1; t0 = pt0; t1 = pt1;
#pragma unroll
            for (int mi = 0; mi < NF; ++mi) xraw[mi] = px[mi];
          } else {
            read_tab_x(ks, s0, s1, t0, t1, xraw);
            __builtin_amdgcn_sched_barrier(0);    // constants and pixel fragments first: BN(0) does not wait for the weights
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int row = nch0 + ni * 16 + frow;
            wa[ni] = *(const f16x8 *)(Ws + row * ROWB + (stage_swz<BK>(row, ks * 4 + fch) << 4));
            if (ni < 2) __builtin_amdgcn_sched_barrier(0);    // (the first MFMAs wait for these)
          }
          __builtin_amdgcn_sched_barrier(0);
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
          u32x4 xb, xn;
#pragma unroll
          for (int j = 0; j < 4; ++j) xb[j] = clamp2_pk(xraw[0][j], sc[2 * j], sc[2 * j + 1], sh[2 * j], sh[2 * j + 1]);
#pragma unroll
          for (int mi = 0; mi < NF; ++mi) {
            const f16x8 xf = __builtin_bit_cast(f16x8, xb);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xf, acc[ni][mi], 0, 0, 0);
              if (mi + 1 < NF && ni < 4) {
                xn[ni] = clamp2_pk(xraw[mi + 1][ni], sc[2 * ni], sc[2 * ni + 1], sh[2 * ni], sh[2 * ni + 1]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            refill_group(ks * MIW + mi);
            xb = xn;
          }
        }
#pragma unroll
        for (int mi = NF; mi < MIW; ++mi) refill_group(ks * MIW + mi);     // (fragments this wave does not compute)
      } else if constexpr (EX && SPREAD) {
        // a k-step past the channels in the MIDDLE of the loop (last k-tile of the hi pass when K % 64 == 32): its
        // share of the refill goes out all the same
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) refill_group(ks * MIW + mi);
      }
    }
    st = st == 2 ? 0 : st + 1;
  };
  {
    int kt = (TN_EXP & 2) ? nk : 0;
    for (; kt + 1 < nk; ++kt) ktile(kt, std::integral_constant<int, 1>{});   // steady state: one younger stage in flight
    for (; kt < nk; ++kt) ktile(kt, std::integral_constant<int, 0>{});       // drain
  }
  };
  if constexpr (MIW == 4) {
    if (nf == 4) kloop(std::integral_constant<int, 4>{});
    else if (nf == 3) kloop(std::integral_constant<int, 3>{});
    else if (nf == 2) kloop(std::integral_constant<int, 2>{});
    else if (nf == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  } else if constexpr (MIW == 3) {
    if (nf == 3) kloop(std::integral_constant<int, 3>{});
    else if (nf == 2) kloop(std::integral_constant<int, 2>{});
    else if (nf == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  } else if constexpr (MIW == 2) {
    if (nf == 2) kloop(std::integral_constant<int, 2>{});
    else if (nf == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  } else {
    static_assert(MIW == 1, "K loop dispatch: unknown fragment count per wave");
    if (nf == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  }
  __syncthreads();   // every wave is done reading the DMA ring; the tile may now be written

  }

  DL_STAMP(2);

  // ---- zero padding of the tile: the two pad columns of every row, out-of-image halo rows ----
  {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int i = t; i < TR * 32; i += 512) {      // (TR = 18 at 16 x 16: more pad cells than threads)
      const int tr = i >> 5, side = (i >> 4) & 1, ch = i & 15;
      *(uint4 *)(tile + (tr * WP + side * (WP - 1)) * 256 + ch * 16) = z4;
    }
    if (top_pad)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + idx * 16) = z4;
    if (r0 + ROUT >= H)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + (TR - 1) * WP * 256 + idx * 16) = z4;
  }
  DL_STAMP(3);
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
      sl15[mi] = tile_swz<SWZ16>(slot);
      ok[mi] = m < MA && mi < nfw;
    }
    // rows past the tile (edge tiles, the last wave): their stores go to a slot of the tile's slack instead of being masked
    // out (no exec juggling per store); the table constants of channel group ni+1 are requested before group ni is worked on
    // (hipcc waits for each pair of reads right behind issuing it: eight exposed LDS round trips per wave)
    unsigned char *wdst[MIW];
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) wdst[mi] = ok[mi] ? dst[mi] : tile + G::DUMP_SLOT * 256;
    // the store address of channel group ni is the one of group 0 with bits 5..7 flipped by ni: the group's chunk
    // (nch0 >> 3) + 2 ni + (fch >> 1) has ni in bits 1..2(3) and nothing else there (so + is ^ and the swizzle's XOR commutes), and
    // the slot base (256-byte aligned tile, 256-byte slots, + 0 / 8) has zeros in bits 4..7 - one v_xor per store instead
    // of an or + a shift-add
    const int chunk0 = (nch0 >> 3) + (fch >> 1);
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;   // (an integer XOR on a generic pointer would end in flat stores)
    unsigned wb0[MIW];
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) wb0[mi] = (unsigned)(size_t)(lds_bytes)(wdst[mi] + ((chunk0 ^ sl15[mi]) << 4));
    float4 sv = *(const float4 *)(tab2 + nch0 + fch * 4), tv = *(const float4 *)(tab2 + 128 + nch0 + fch * 4);
#pragma unroll
    for (int ni = 0; ni < ((TN_EXP & 4) ? 0 : NI); ++ni) {
      float4 sn = sv, tn = tv;
      if (ni + 1 < NI) {
        sn = *(const float4 *)(tab2 + nch0 + (ni + 1) * 16 + fch * 4);
        tn = *(const float4 *)(tab2 + 128 + nch0 + (ni + 1) * 16 + fch * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      [[maybe_unused]] const int chunk = (nch0 >> 3) + ni * 2 + (fch >> 1);   // channels n>>3
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        const f16x4 hv = bn_relu4_from_f32(acc[ni][mi], sv, tv);
        *(__attribute__((address_space(3))) f16x4 *)(size_t)(wb0[mi] ^ (unsigned)(ni << 5)) = hv;
      }
      __builtin_amdgcn_sched_barrier(0);
      sv = sn;
      tv = tn;
    }
  }
  // 3x3 weights: taps 0 and 1 -> ring slots 0 and 1 (published by the barrier below); the registers go on to taps 3, 4
  *(f16x8 *)(w3slot(0) + t * 16) = wq[0];
  *(f16x8 *)(w3slot(1) + t * 16) = wq[1];
  wq[0] = w3tap(3);
  wq[1] = w3tap(4);
  __syncthreads();
  DL_STAMP(4);

  // ======================= phase B: y = conv3x3(tile) ========================================
  // 16-slot output fragments, v_mfma_f32_16x16x32_f16 with the packed weights as the A operand (a lane
  // ends with 4 consecutive output channels of its pixel).  A wave owns whole fragments over the full
  // K = 9 x 128, so there are no partial sums to combine; the fragments are dealt out so that the two
  // waves of a SIMD (w, w+4) together get NF16/4 of them.
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

  // NFR = fragments this wave really owns (wave-uniform); the loop body is branch-free so the
  // compiler can run the LDS reads ahead of the MFMAs
  // Operands double-buffered by hand: the fragments of step (tap, kk + 1) are requested BEFORE the MFMAs of step
  // (tap, kk) - through the tap boundary for the pixel fragments (the tile does not change during phase B), behind the
  // tap's barrier for the weight fragments (their ring slot is published by it).  Left to itself hipcc issues every
  // group of ds_reads directly in front of the MFMAs that need them (s_waitcnt lgkmcnt(1) behind four reads just
  // issued): 72 exposed LDS round trips per tile, phase B 12 900 cycles for 8 060 cycles of MFMA work on the
  // busiest SIMD.  sched_barrier keeps the compiler from sinking the prefetch below the MFMAs again.
  auto phase_b = [&](auto nfr_tag) {
    constexpr int NFR = decltype(nfr_tag)::value;
    constexpr int NX = NFR > 0 ? NFR : 1;
    constexpr int NT = EX ? 18 : 9;      // EX: the nine taps once more with the lo image of the weights
    f16x8 xb[2][NX], wb[2][2];
    auto load_x = [&](f16x8 *xf, int tap, int kk) {
      const int dy = (tap % 9) / 3 - 1, dx = (tap % 9) - ((tap % 9) / 3) * 3 - 1;
      const int off = WP + dy * WP + dx + px + 16 * u0;   // slot of this lane's pixel, fragment u0
      const int chunk = kk * 4 + kg;
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const int slot = off + 16 * j;
        xf[j] = *(const f16x8 *)(tile + slot * 256 + ((chunk ^ tile_swz<SWZ16>(slot)) << 4));
      }
    };
    auto load_w = [&](f16x8 *wf, int tap, int kk) {
      const unsigned char *wring = w3slot(tap % 3) + lane * 16;
      if constexpr (NFR > 0) {
        wf[0] = *(const f16x8 *)(wring + (kk * 2) * 1024);
        wf[1] = *(const f16x8 *)(wring + (kk * 2 + 1) * 1024);
      } else {
        wf[0] = wf[1] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    };
    load_x(xb[0], 0, 0);
    load_w(wb[0], 0, 0);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;       // (four steps per tap: the buffers line up again at every tap)
        if (kk < 3) {
          load_x(xb[nxt], tap, kk + 1);
          load_w(wb[nxt], tap, kk + 1);
        } else if (tap + 1 < NT) {
          load_x(xb[nxt], tap + 1, 0);
          load_w(wb[nxt], tap + 1, 0);               // slot (tap+1) % 3: written in tap-1, published by this tap's barrier
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NFR; ++j) {
          bacc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[cur][0], xb[cur][j], bacc[j][0], 0, 0, 0);
          bacc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[cur][1], xb[cur][j], bacc[j][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 1) {
          // The tap's one barrier sits in its MIDDLE, where nothing is waited for behind it (the operands of the next
          // step are on their way already).  Everybody is done with tap-1, so its slot (tap+2) % 3 takes tap+2's weights
          // now; the NEXT tap's barrier publishes them, one and a half taps before their first read.  Three taps of
          // prefetch in registers, rotating with static indices.
          __syncthreads();
          if (tap + 2 < NT) {
            *(f16x8 *)(w3slot((tap + 2) % 3) + t * 16) = wq[(tap + 2) % 3];
            if (tap + 5 < NT) wq[(tap + 2) % 3] = w3tap(tap + 5);
          }
        }
      }
    }
    __syncthreads();   // every wave is past its last fragment read: the tile and the ring may be overwritten
  };
  if (TN_EXP & 1) {
  } else if (u1 - u0 == MAXU) phase_b(std::integral_constant<int, MAXU>{});
  else phase_b(std::integral_constant<int, MAXU - 1>{});

  DL_STAMP(5);
  if (TN_EXP & 8) return;   // (experiments never chain)
  // ---- fp16, through an LDS row buffer (80-B pitch per slot: 64 B of data, pitch chosen against
  // ds_write_b64 bank conflicts) so that the global store is 16 B per lane with 4 lanes covering one
  // pixel's 32 channels contiguously.  The buffer aliases the tile: every wave is past the last tap ----
  float nxs[2] = {0.f, 0.f}, nxt[2] = {0.f, 0.f}, nx2 = 0.f;   // CHAIN: the next layer's BN tables on their way
  int nxK = 0;
  if constexpr (PP == 0 || PP == 2 || PP == 4) {
    if constexpr (!CHAIN) {
      if (vb + (int)gridDim.x < nvb) {    // the tile and the 3x3 ring are dead: request the next tile's first stages
        const TileAt nx = tile_at(vb + gridDim.x);
        set_src(nx.xbase, nx.MA, a.w1, w1ld);
        issue(1, 0);
        if (nk > 1) issue(2, 1);
        if constexpr (PP == 4) {
          if (nk > 2) issue(3, 2);
        }
        primed = true;
      }
    } else if (layer + 1 < nlayers) {     // same frame, next layer: its first 128 input channels exist already
      const DenseLayerDev dn = a.chain[layer + 1];
      nxK = K + 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = t + 512 * i;
        if (idx < nxK) { nxs[i] = dn.s1[idx]; nxt[i] = dn.t1[idx]; }
      }
      if (t < 256) nx2 = t < 128 ? dn.s2[t] : dn.t2[t - 128];
      // the next layer's first two k-tiles (its own weight pitch; a chained layer has at least four k-tiles, so
      // the activation pointers do not wrap yet)
      const int nxc = (nxK + BK - 1) / BK;
      set_src(xbase, MA, dn.w1, EX ? 2 * nxc * BK : nxK);
#pragma unroll
      for (int st = 1; st <= (PP == 4 ? 3 : 2); ++st)       // (the pipelined loop keeps three k-tiles requested: slots 1, 2, 3)
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          dma16(src[j], lds0 + st * G::STAGE + piece_of(j) * 1024);
          src[j] += BK;
        }
      primed = true;
    }
  }
  unsigned char *obuf = smem;
#pragma unroll
  for (int j = 0; j < MAXU; ++j) {
    if (u0 + j < u1) {
      const int srel = 16 * (u0 + j) + px;           // slot relative to the first output slot
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
  {
    f16 *ybase = a.buf + ((long)img * H * W + (long)r0 * W) * ldc + K;
    for (int id = t; id < ROUT * W * 4; id += 512) {
      const int pi = id >> 2, c = id & 3;
      const int r = pi / W, x = pi - r * W;
      const uint4 v = *(const uint4 *)(obuf + (r * WP + x + 1) * 80 + c * 16);
      *(uint4 *)(ybase + (long)pi * ldc + c * 8) = v;
    }
  }
  DL_STAMP(6);
  if constexpr (CHAIN) {
    if (nxK) {            // the next layer's tables (nobody reads the old ones any more)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = t + 512 * i;
        if (idx < nxK) { tab1[idx] = nxs[i]; tab1[G::KMAX + idx] = nxt[i]; }
      }
      if (t < 256) tab2[t] = nx2;
    }
    wait_vmcnt<0>();      // this layer's stores have completed (stores count in vmcnt on gfx9) ...
    __syncthreads();      // ... for every wave, before the next layer's loads of the same frame
  }
  }   // layer
  }   // tile
}

template <int W, int ROUT, int BM, int BK, int PP, bool CHAIN = false, bool EX = false>
int launch_geom(const DenseLayerArgs &a, hipStream_t s) {
  using G = DLGeom<W, ROUT, BM, BK>;
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_layer_kernel<W, ROUT, BM, BK, PP, CHAIN, EX>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)));
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    TN_HIP_CHECK(hipGetDevice(&dev));
    TN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
    if (getenv("TN_PERSIST_WGS")) ncu = atoi(getenv("TN_PERSIST_WGS"));   // tuning hook
  }
  TN_REQUIRE(a.K + 32 * (a.nchain > 0 ? a.nchain - 1 : 0) <= G::KMAX, "dense_layer: more input channels than this block's table space holds");
  const int nvb = a.B * (a.H / ROUT);
  // one workgroup per CU walking its tiles (flat K loops, single layer); otherwise one workgroup per tile
  const bool persist = !CHAIN && (PP == 0 || PP == 2) && !(a.variant & 16) && nvb > ncu && ncu > 0;
  const dim3 grid(persist ? ncu : nvb), block(512);
  hipLaunchKernelGGL((dense_layer_kernel<W, ROUT, BM, BK, PP, CHAIN, EX>), grid, block, G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

}  // namespace

// (32 x 32 and 16 x 16: the third and fourth block of a 512 x 512 input)
bool dense_layer_big_supported(int H, int W) { return H == W && (H == 56 || H == 28 || H == 14 || H == 7 || H == 64 || H == 32 || H == 16); }
// input channels a layer of that map size may have (the BatchNorm tables' share of LDS, DLGeom::KMAX): an input size other
// than 224 can put a 56x56 map into the second block (448: K up to 480), where the tile kernel has no room for it
int dense_layer_big_kmax(int W) { return W == 56 ? 256 : W == 28 ? 512 : 1024; }

int launch_dense_layer_big(const DenseLayerArgs &a, hipStream_t s) {
  const int klast = a.K + 32 * (a.nchain > 0 ? a.nchain - 1 : 0);
  TN_REQUIRE(a.K % 32 == 0 && klast <= 1024 && a.ldc % 8 == 0 && klast + 32 <= a.ldc, "dense_layer: bad channel geometry");
  if (a.exact) {     // hi + lo weights: the default K loop only
    if (a.nchain > 0) {
      TN_REQUIRE(a.H == a.W && (a.H == 14 || a.H == 7 || a.H == 16) && a.chain, "dense_layer: layer chaining needs whole-frame tiles (16x16, 14x14, 7x7)");
      if (a.H == 14) return launch_geom<14, 14, 256, 64, 2, true, true>(a, s);
      if (a.H == 16) return launch_geom<16, 16, 384, 32, 2, true, true>(a, s);
      return launch_geom<7, 7, 64, 64, 2, true, true>(a, s);
    }
    if (a.H == 56 && a.W == 56) return launch_geom<56, 7, 512, 32, 2, false, true>(a, s);
    if (a.H == 28 && a.W == 28) return launch_geom<28, 14, 512, 32, 2, false, true>(a, s);
    if (a.H == 14 && a.W == 14) return launch_geom<14, 14, 256, 64, 2, false, true>(a, s);
    if (a.H == 7 && a.W == 7) return launch_geom<7, 7, 128, 64, 2, false, true>(a, s);
    // (round 6: the maps of a 512 x 512 input)
    if (a.H == 64 && a.W == 64) return launch_geom<64, 4, 384, 32, 2, false, true>(a, s);
    if (a.H == 32 && a.W == 32) return launch_geom<32, 8, 384, 32, 2, false, true>(a, s);
    if (a.H == 16 && a.W == 16) return launch_geom<16, 16, 384, 32, 2, false, true>(a, s);
    TN_REQUIRE(false, "dense_layer: unsupported spatial size");
  }
  if (a.nchain > 0) {
    TN_REQUIRE(a.H == a.W && (a.H == 14 || a.H == 7 || a.H == 16) && a.chain, "dense_layer: layer chaining needs whole-frame tiles (16x16, 14x14, 7x7)");
    if (a.H == 14) return launch_geom<14, 14, 256, 64, 2, true>(a, s);
    if (a.H == 16) return launch_geom<16, 16, 384, 32, 2, true>(a, s);
    // 7x7: 4 x 2 wave split over a 64-row tile (variant bit 5: the 8 x 1 split over 128 rows, for A/B runs)
    // bit 9: the flat K loop instead of the software-pipelined one (A/B runs)
    if (a.variant & 32) return launch_geom<7, 7, 128, 64, 2, true>(a, s);
    return (a.variant & 512) ? launch_geom<7, 7, 64, 64, 2, true>(a, s) : launch_geom<7, 7, 64, 64, 4, true>(a, s);
  }
  // K-loop flavour (tuning hook, variant bit 3): default -> flat loop with the refill spread over the MFMA groups (measured
  // best); bit 3 -> flat with the refill up front.  (The ping-pong variants of round 1, bit 2, measured the same as the
  // flat loop and are gone.)
  const int pp = (a.variant & 8) ? 0 : 2;
#define TN_GEOM(W_, R_, BM_, BK_) (pp == 0 ? launch_geom<W_, R_, BM_, BK_, 0>(a, s) : launch_geom<W_, R_, BM_, BK_, 2>(a, s))
  if (a.H == 56 && a.W == 56) return TN_GEOM(56, 7, 512, 32);
  if (a.H == 28 && a.W == 28) return TN_GEOM(28, 14, 512, 32);
  if (a.H == 14 && a.W == 14) return TN_GEOM(14, 14, 256, 64);
  if (a.H == 64 && a.W == 64) return TN_GEOM(64, 4, 384, 32);      // (second block of a 512 x 512 input past K = 320: 1.5 x halo rows)
  if (a.H == 32 && a.W == 32) return TN_GEOM(32, 8, 384, 32);
  if (a.H == 16 && a.W == 16) return TN_GEOM(16, 16, 384, 32);
  if (a.H == 7 && a.W == 7) {
    // default: the 4 x 2 wave split with the software-pipelined K loop, as the chained launch (bit 9: the older
    // 8 x 1 split over a 128-row tile with the K-loop flavours of bits 2-3)
    if (!(a.variant & 512) && pp == 2) return launch_geom<7, 7, 64, 64, 4>(a, s);
    return TN_GEOM(7, 7, 128, 64);
  }
#undef TN_GEOM
  TN_REQUIRE(false, "dense_layer: unsupported spatial size");
}

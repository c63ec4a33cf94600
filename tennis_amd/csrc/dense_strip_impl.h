// Fused DenseNet dense layer, "strip-streaming" form (round 3; 56x56 and 28x28 blocks, K <= 320):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// (reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_dense_layer: BatchNorm-Activation-
// Conv1x1-BatchNorm-Activation-Conv3x3-Concat).  dense_layer_big.hip owns ROUT image rows per 8-wave workgroup and runs its
// phases (K loop, epilogue, 3x3, store) one after the other behind barriers, with the bottleneck tile in LDS aliasing the
// K-loop ring.  This kernel is built the other way round:
//
// * a workgroup = one frame, 4 waves = ONE wave per SIMD with the whole 512-entry register file; no barrier after the
//   prologue.  A wave owns a pair of adjacent 14-pixel column strips (2 x 16 slots with their halo columns = the N = 32 of
//   one v_mfma_f32_32x32x16_f16) and walks DOWN its rows one image row at a time.  One wave per SIMD issues one instruction
//   per ~4 cycles: a 32-cycle 32x32x16 MFMA hides ~6 other instructions, a 16-cycle 16x16x32 only one
//   (scripts/microbench/slotbench.hip) - and this layer needs ~3.5 element-wise / LDS / load instructions per 16x16x32's worth
//   of MFMA work, so the kernel is built on the 32x32 shape.
// * the 128-channel bottleneck never touches LDS: with the weights as the A operand, the accumulator layout of the 1x1 GEMM
//   (lane = slot l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) IS the B-operand layout of the 3x3's MFMAs once the 3x3
//   weights are packed with the matching permutation of their input channels (chained MFMAs: BN2 + ReLU + fp16 pack are
//   lane-local).  The sliding window of three bottleneck rows lives in 96 literal accumulator registers.
// * the 3x3 convolution applies the three kernel columns to the SAME input fragment into three accumulator sets which are
//   combined at the end by two DPP row shifts: out[x] = acc[dx=0][x] + acc[dx=-1][x-1] + acc[dx=+1][x+1].  The shifts stay
//   inside the 16-lane row = inside one strip (outputs of the halo slots are never stored): no cross-fragment carry.
// * the layer's 1x1 weights (K x 128, as A fragments) and all nine taps of the 3x3 weights (72 KB) are resident in LDS for
//   the whole launch; activations go HBM -> registers directly in fragment shape (64 B per lane per 64-channel super-step;
//   the two lanes of a pixel cover one 128-B line) through a register ring that holds one whole row and is refilled with
//   the next row as it is read out.
// * BN2 costs no element-wise multiply-add: its scale is folded into the 1x1 weights on the host (before the fp16 rounding:
//   the fp16 model is DEFINED that way, weights.as_fp16_model) and its shift enters the GEMM through one extra 16-channel
//   k-step whose pixel fragment is the constant (1, 1, m, 0, ...): weight columns shift_hi, shift_lo (two fp16 numbers = 22
//   bits of the fp32 shift) and 1; m = -60000 in the lanes of padding columns, so that their ReLU'd bottleneck is 0 as
//   the convolution's zero padding demands.  What is left of epilogue A is convert + ReLU + the window write.
// * the schedule is pinned by hand: the body is a sequence of SLOTS - one MFMA followed by its share of everything else -
//   with a scheduling barrier behind each.  1x1 slots carry BN1 + ReLU of the next k-step, the weight-fragment and constant
//   reads, the ring refill and the DPP epilogue of the PREVIOUS output row; 3x3 slots carry the weight-fragment reload, the
//   BN2 epilogue of the row just computed (its window row is only needed by the last third of the slots) and the first
//   operands of the next row.  Nothing runs outside an MFMA's shadow in the steady state.
//
// No halo recompute in y at 56x56 beyond one row per wave (29 / 28); 16 / 14 in x.
#include <array>
#include <type_traits>

#pragma once
#include "common.h"

#ifndef TN_DS_EXP
#define TN_DS_EXP 0   // timing experiments only (results wrong): bit 0 no activation loads inside the row loop, bit 1 no 3x3 phase, bit 2 no 1x1 phase, bit 3 / 4 no weight-fragment reads in the 3x3 / 1x1 phase
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

constexpr int kW3Bytes = 3 * 8 * 3 * 1024;   // [dy][k16-step][dx] fragments of 1 KiB

template <int W, int KS>
struct DSGeom {
  // A frame = NPAIR strip pairs (28 columns each; the last pair of a width that is no multiple of 28 is partly empty) x NCHUNK
  // row chunks of ROWS output rows = NITEM work items, one per wave, four per workgroup: 56 x 56 and 28 x 28 frames are one
  // workgroup (2 x 2 and 1 x 4 items); the 128 x 128 / 64 x 64 maps of a 512 x 512 input take 5 x 4 = 20 / 3 x 4 = 12 items =
  // 5 / 3 workgroups per frame.  A chunk recomputes the bottleneck row above and below it.
  static constexpr int NPAIR = (W + 27) / 28;    // strip pairs per frame
  static constexpr int ROWS = W == 56 ? 28 : W == 28 ? 7 : W / 4;     // output rows per wave
  static constexpr int NCHUNK = W / ROWS;
  static constexpr int NITEM = NPAIR * NCHUNK, WGS = NITEM / 4;       // workgroups per frame
  static constexpr int KQ = 2 * KS;              // 16-channel k-steps
  static constexpr int NSU = (KS + 1) / 2;       // 64-channel super-steps (the last one is half when KS is odd)
  static constexpr int W1OFF = kW3Bytes;                   // KQ + 1 k-steps of 4 fragments: the last one carries BN2's shift
  static constexpr int T1OFF = W1OFF + (KQ + 1) * 4096;    // a1[K] | b1[K] (fp16: BN1 as v_pk_fma_f16, csrc/calib_host.hip)
  static constexpr int LDS_BYTES = T1OFF + KS * 32 * 4;
  static_assert(W % ROWS == 0 && NITEM % 4 == 0 && ROWS >= 4, "strip geometry");
  static_assert(NSU <= 5, "the activation ring holds five super-steps");
  static_assert(LDS_BYTES <= 160 * 1024, "weights do not fit LDS");
};

// The bottleneck window lives in LITERAL accumulator registers a[160:255] (three rows x eight 16-channel k-steps x one
// 4-register MFMA B operand): hipcc's MFMA builtin takes A / B from VGPRs only, and a window held in compiler-allocated AGPR
// values gets its live ranges split and copied through VGPRs (measured in the first version: ~200 extra v_accvgpr moves per
// row group, some of them directly in front of the asm MFMA that reads the register two cycles later - a hazard hipcc cannot
// see).  Every slot of the kernel body names all 96 registers as clobbered, which keeps compiler values out of them;
// scripts/audit_strip_isa.py checks the ISA for strays (cdna_hip_programming.md 5.7 item 4).
#define TN_WIN_BASE 160
#define TN_WIN_CLOBBER                                                                                                              \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175",   \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191",   \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207",   \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223",   \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239",   \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
constexpr int win_reg(int prow, int t) { return TN_WIN_BASE + 32 * prow + 4 * t; }
// keeps compiler values that are live here out of the window registers (no instruction)
#define TN_WIN_FENCE() asm volatile("" ::: TN_WIN_CLOBBER)

// four packed VGPRs -> window tuple (physical row PROW, k-step T).  v_accvgpr_write -> MFMA operand read needs two wait states:
// the schedule puts at least eight slots between the write of a tuple and the first MFMA that reads it
template <int PROW, int T>
__device__ __forceinline__ void win_write(const unsigned v0, const unsigned v1, const unsigned v2, const unsigned v3) {
  constexpr int B = win_reg(PROW, T);
  asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
               :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(B), "n"(B + 1), "n"(B + 2), "n"(B + 3) : TN_WIN_CLOBBER);
}
template <int R>
__device__ __forceinline__ void win_zero_reg() {
  asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "n"(R) : TN_WIN_CLOBBER);
}
// a bottleneck row above / below the image: zeros
template <int PROW>
__device__ __forceinline__ void win_zero() {
  [&]<int... I>(std::integer_sequence<int, I...>) { (win_zero_reg<win_reg(PROW, 0) + I>(), ...); }(std::make_integer_sequence<int, 32>{});
  asm volatile("s_nop 1");
}
// one 3x3 weight fragment against window tuple (PROW, T)
template <bool FIRST, int PROW, int T>
__device__ __forceinline__ void mfma32_win(f32x16 &d, const u32x4 a) {
  constexpr int B0 = win_reg(PROW, T);
  if constexpr (FIRST)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&a"(d) : "v"(a), "n"(B0), "n"(B0 + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+a"(d) : "v"(a), "n"(B0), "n"(B0 + 3));
}
// The LAST k-step of a 3x3 phase: its three MFMAs (one per kernel column) and the wait states their results need before
// anything but an MFMA of the same chain may read them, in ONE statement.  hipcc knows nothing about the latency of an asm
// MFMA: where the accumulators change registers at a control-flow join it put v_accvgpr_mov copies four instructions behind
// the last MFMA, in front of the wait states the consumer carried (round 3, the 128 x 128 geometry: registers 14 / 15 of every
// accumulator - the last pass of the MFMA - copied too early; which instantiations get such copies is the register
// allocator's choice).  With the wait inside the producing statement no copy can come between.
template <int PROW, int T>
__device__ __forceinline__ void mfma32_win_last3(f32x16 &d0, f32x16 &d1, f32x16 &d2, const u32x4 a0, const u32x4 a1, const u32x4 a2) {
  constexpr int B0 = win_reg(PROW, T);
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, a[%c6:%c7], %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, a[%c6:%c7], %1\n\t"
               "v_mfma_f32_32x32x16_f16 %2, %5, a[%c6:%c7], %2\n\ts_nop 15\n\ts_nop 3"
               : "+a"(d0), "+a"(d1), "+a"(d2) : "v"(a0), "v"(a1), "v"(a2), "n"(B0), "n"(B0 + 3));
}
__device__ __forceinline__ f32x16 mfma32(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

#define TN_INL __attribute__((always_inline))
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  [&]<int... I>(std::integer_sequence<int, I...>) TN_INL { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int V>
using ic = std::integral_constant<int, V>;

#define TN_SB() __builtin_amdgcn_sched_barrier(0)

// One LDS-DMA piece: 64 lanes x 16 B, global (per-lane address) -> LDS (wave-uniform base in M0 + lane * 16).  Inline asm:
// hipcc treats the builtin as an LDS store and orders every later ds_read behind a vmcnt(0) (dense_layer_big.hip).
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// ---- the static schedule ----
struct PItem { int kind, q, j; };     // kind: 0 none, 1 C(q), 2 BN(q).j, 3 LD(u = q).i = j
template <int KS>
struct PList { PItem it[7 * 2 * KS + 8]; int n; };
template <int KS>
constexpr PList<KS> make_pl() {
  constexpr int KQ = 2 * KS, NSU = (KS + 1) / 2;
  constexpr bool ODD = (KS & 1) != 0;
  PList<KS> l{};
  int n = 0;
  for (int c = 0; c < 3 && c < KQ; ++c) l.it[n++] = PItem{1, c, 0};
  for (int q = 0; q < KQ; ++q) {
    for (int j = 0; j < 4; ++j) l.it[n++] = PItem{2, q, j};
    if (q + 3 < KQ) l.it[n++] = PItem{1, q + 3, 0};
    const int u = q >> 2;
    const bool half = ODD && u == NSU - 1;
    if (half ? (q & 3) == 1 : (q & 3) == 3)
      for (int i = 0; i < (half ? 2 : 4); ++i) l.it[n++] = PItem{3, u, i};
  }
  l.n = n;
  return l;
}
template <int KS>
inline constexpr PList<KS> kPL = make_pl<KS>();
template <int KS>
constexpr PItem pl_at(int idx) { return idx >= 0 && idx < kPL<KS>.n ? kPL<KS>.it[idx] : PItem{0, 0, 0}; }
template <int KS>
constexpr int pl_len() { return kPL<KS>.n; }
constexpr int kPLB = 30;                 // pipeline items that run in the previous row's 3x3 phase (slots 42 - 71)
constexpr int kXN = 8;                   // pixel-fragment buffers
struct ASlot { int pl0, npl, epb0, nepb; };
template <int N> struct ASched { ASlot s[N]; };
// 1x1 slot i (k-step i / 4): pipeline items [pl0, pl0 + npl) and epilogue B items [epb0, epb0 + nepb).  The items left are
// spread evenly over the slots left; BN(q) has to be complete when k-step q starts, so the pipeline goes first whenever it is
// needed within the current k-step, otherwise the epilogue B items (24) are used up first.
template <int KS>
constexpr auto make_a_sched() {
  constexpr int KQ = 2 * KS, NA = 4 * (KQ + 1), NPL = pl_len<KS>();
  ASched<NA> r{};
  int bn3[KQ + 1] = {};                                   // index of BN(q).3 in the list
  for (int i = 0; i < NPL; ++i)
    if (kPL<KS>.it[i].kind == 2 && kPL<KS>.it[i].j == 3) bn3[kPL<KS>.it[i].q] = i;
  int pos = NPL < kPLB ? NPL : kPLB, epb = 0;
  for (int i = 0; i < NA; ++i) {
    const int qn = i / 4 + 1, rem = 3 - i % 4;          // next k-step, slots left in this one behind slot i
    const int need_end = qn < KQ ? bn3[qn] + 1 : 0;
    const int left = NA - i, total = (NPL - pos) + (24 - epb);
    int quota = (total + left - 1) / left;
    int must = need_end - rem - pos;                    // items the pipeline has to run in this slot not to fall behind
    if (must < 0) must = 0;
    if (must > NPL - pos) must = NPL - pos;
    int npl = must;
    if (quota < npl) quota = npl;
    int ne = 24 - epb < quota - npl ? 24 - epb : quota - npl;
    const int more = NPL - pos - npl < quota - npl - ne ? NPL - pos - npl : quota - npl - ne;
    npl += more;
    r.s[i] = ASlot{pos, npl, epb, ne};
    pos += npl;
    epb += ne;
  }
  return r;
}
template <int KS>
constexpr bool a_sched_complete() {        // every epilogue B item and every pipeline item has a slot
  constexpr int KQ = 2 * KS, NA = 4 * (KQ + 1);
  constexpr auto sa = make_a_sched<KS>();
  int npl = pl_len<KS>() < kPLB ? pl_len<KS>() : kPLB, nepb = 0;
  for (int i = 0; i < NA; ++i) { npl += sa.s[i].npl; nepb += sa.s[i].nepb; }
  return npl == pl_len<KS>() && nepb == 24;
}

template <int W, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_strip_kernel(DenseStripArgs a) {
  using G = DSGeom<W, KS>;
  constexpr int H = W, NSU = G::NSU, K = KS * 32, KQ = G::KQ, ROWS = G::ROWS;
  constexpr bool ODD = (KS & 1) != 0;
  constexpr int XN = kXN, NPL = pl_len<KS>(), PLB = NPL < kPLB ? NPL : kPLB;
  static_assert(a_sched_complete<KS>(), "1x1 slot schedule leaves work unassigned");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  if (a.ts && tid == 0) {
    a.ts[(size_t)blockIdx.x * 128 + 127] = __builtin_amdgcn_s_memtime();
    a.ts[(size_t)blockIdx.x * 128 + 126] = __builtin_amdgcn_s_memrealtime();     // 100 MHz
  }

  // ---- prologue: the layer's weights and tables -> LDS (once per launch) ----
  // LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no registers): every piece of the 72 + 4 (KQ + 1) KiB is in
  // flight at once and ONE wait follows.  Through registers (load, ds_write, 4 - 6 pieces per round trip) the copy took
  // 9 800 cycles at K = 320 - 13 % of a 28x28 launch, whose waves only have eight rows each to amortise it over.
  {
    typedef __attribute__((address_space(3))) void *lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    constexpr int P3 = kW3Bytes / 1024, P1 = (KQ + 1) * 4;
    const unsigned char *g3 = (const unsigned char *)a.w3s + ln * 16, *g1 = (const unsigned char *)a.w1s + ln * 16;
    for (int p = wv; p < P3; p += 4) dma16(g3 + p * 1024, lds0 + p * 1024);
    for (int p = wv; p < P1; p += 4) dma16(g1 + p * 1024, lds0 + G::W1OFF + p * 1024);
    f16 *t1 = (f16 *)(smem + G::T1OFF);      // (the constants ARE fp16 numbers: bn_relu_fold_fp16)
    for (int i = tid; i < K; i += 256) {
      t1[i] = (f16)a.s1[i];
      t1[K + i] = (f16)a.t1[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int item = (int)(blockIdx.x % G::WGS) * 4 + wid;
  const int pair = item % G::NPAIR, part = item / G::NPAIR;
  const int r_lo = part * ROWS, r_hi = r_lo + ROWS;
  const int x = 14 * (2 * pair + (n >> 4)) - 1 + (n & 15);
  const bool xvalid = x >= 0 && x < W;
  const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
  const int ldc = a.ldc;
  const unsigned rowpitch = (unsigned)W * ldc * 2;
  unsigned char *fb = (unsigned char *)(a.buf + (size_t)(blockIdx.x / G::WGS) * H * W * ldc);
  const unsigned colb = (unsigned)xc * ldc * 2 + 64 * h;     // full super-steps: 64 B per lane
  const unsigned colh = (unsigned)xc * ldc * 2 + 32 * h;     // the trailing half super-step: 32 B per lane
  const bool store_ok = (n & 15) >= 1 && (n & 15) <= 14 && xvalid;
  const unsigned outb = (unsigned)xc * ldc * 2 + K * 2 + 32 * h;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(fb, 0, (int)((unsigned)H * rowpitch), 0x00020000);

  const unsigned char *w1l = smem + G::W1OFF + lane * 16;
  const unsigned char *w3l = smem + lane * 16;
  const f16 *tab1 = (const f16 *)(smem + G::T1OFF);
  // the pixel fragment of the shift k-step: (1, 1, mask, 0, 0, 0, 0, 0) in the lanes that hold k = 0 .. 7
  const u32x4 xb_shift = {h == 0 ? 0x3c003c00u : 0u, (h == 0 && !xvalid) ? 0x0000fb53u : 0u, 0u, 0u};   // fp16 1.0 = 0x3c00, -60000 = 0xfb53

  // ================= state that lives across slots =================
  // every LDS read is issued at least two k-steps (6 - 8 slots, >= 200 cycles) ahead of its consumer: with one wave per SIMD
  // nothing else covers an exposed LDS round trip
  u32x4 ring[5][4];      // activation ring [super-step][k-step]: 16 B per lane = 8 channels of the lane's pixel; holds one row
  f32x16 acc[4];         // 1x1 accumulators [32-channel block]
  u32x4 wa[2][4];        // 1x1 weight fragments [k-step parity][block] (a register is reloaded for k-step + 2 behind its MFMA)
  u32x4 xb[XN];          // BN1 + ReLU'd pixel fragments [k-step % XN]: the BN pipeline runs up to XN - 2 k-steps ahead
  u32x4 cs[3], ct[3];    // BN1 constants [k-step % 3]: a / b of the lane's eight channels, packed halves
  u32x4 w3f[2][3];       // 3x3 weight fragments [step parity][dx] (reloaded for step + 2 behind their MFMA)
  f32x16 bacc[3];        // 3x3 accumulators [dx]
  unsigned e_pk[4];
  float o_c[2], o_l[2], o_r[2];
  unsigned o_pk[8];

  auto rowbase = [&](int y) TN_INL {
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return fb + (unsigned)yc * rowpitch;
  };
  // ---------------- the BN pipeline: an ordered list of items per bottleneck row ----------------
  //   C(q)    the two ds_read_b128 of k-step q's BN1 constants
  //   BN(q).j BN1 + ReLU of dword j of k-step q's pixel fragment (2 VALU instructions: v_pk_fma_f16, v_pk_max_f16)
  //   LD(u).i one 16-byte activation load of the NEXT row into ring slot u, behind the last BN item that read the slot
  // in the order C0 C1 C2 | BN(0).0-3 C3 | BN(1).0-3 C4 | ... ; the list is consumed one item per slot, first by the spare slots
  // of the previous row's 3x3 phase (PLB items), then by the row's own 1x1 slots (see make_a_sched)
  auto ld_item = [&](auto u_tag, auto i_tag, int y) TN_INL {
    constexpr int U = decltype(u_tag)::value, I = decltype(i_tag)::value;
    constexpr bool HALF = ODD && U == NSU - 1;
    if ((TN_DS_EXP & 1) && y > r_lo + 1) return;
    if constexpr (HALF) ring[U][I] = *(const u32x4 *)(rowbase(y) + colh + 128 * U + 16 * I);
    else ring[U][I] = *(const u32x4 *)(rowbase(y) + colb + 128 * U + 16 * I);
  };
  auto consts_item = [&](auto q_tag) TN_INL {
    constexpr int Q = decltype(q_tag)::value;
    constexpr int U = Q >> 2, I = Q & 3;
    constexpr bool HALF = ODD && U == NSU - 1;
    const int c0 = (HALF ? 64 * U + 16 * h : 64 * U + 32 * h) + 8 * I;
    cs[Q % 3] = *(const u32x4 *)(tab1 + c0);
    ct[Q % 3] = *(const u32x4 *)(tab1 + K + c0);
  };
  auto bn_item = [&](auto q_tag, auto j_tag) TN_INL {
    constexpr int Q = decltype(q_tag)::value, J = decltype(j_tag)::value;
    const unsigned in = ring[Q >> 2][Q & 3][J];
    const unsigned sc = cs[Q % 3][J], sh = ct[Q % 3][J];
    unsigned o;      // clamp(x, lo, hi): BN1 + ReLU without arithmetic or rounding (calib_host.hip::bn_relu_clamp_fold; one statement: between two, hipcc pads the dependency with an s_nop)
    asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3" : "=&v"(o) : "v"(in), "v"(sc), "v"(sh));
    xb[Q % XN][J] = o;
  };
  // pipeline item IDX of the row ybn
  auto pl_item = [&](auto idx_tag, int ybn) TN_INL {
    constexpr PItem it = pl_at<KS>(decltype(idx_tag)::value);
    if constexpr (it.kind == 1) consts_item(ic<it.q>{});
    else if constexpr (it.kind == 2) bn_item(ic<it.q>{}, ic<it.j>{});
    else if constexpr (it.kind == 3) ld_item(ic<it.q>{}, ic<it.j>{}, ybn + 1);
  };
  auto wa_item = [&](auto q_tag, auto mb_tag) TN_INL {
    constexpr int Q = decltype(q_tag)::value, MB = decltype(mb_tag)::value;
    if ((TN_DS_EXP & 16) && Q >= 2) return;     // (timing experiment: the 1x1 phase keeps reusing its first eight weight fragments)
    wa[Q & 1][MB] = *(const u32x4 *)(w1l + (Q * 4 + MB) * 1024);
  };
  auto wa_group = [&](auto q_tag) TN_INL { static_for<4>([&](auto mb_tag) TN_INL { wa_item(q_tag, mb_tag); }); };
  // what the 3x3 phase of the previous row would have done for this row (first rows of a wave)
  auto prologue_exposed = [&](int ybn) TN_INL {
    wa_group(ic<0>{});
    wa_group(ic<1>{});
    static_for<PLB>([&](auto i_tag) TN_INL { pl_item(i_tag, ybn); });
    TN_SB();
  };

  // ---- epilogue A: acc (= BN2 applied) -> ReLU, one rounding to fp16, lane-local pack -> window row PROW; 40 items: per window
  // tuple T (k-step of the 3x3: accumulators 8 (T & 1) .. + 7 of block T >> 1) four convert + ReLU items of two values each and
  // the window write ----
  auto epa_item = [&](auto prow_tag, auto e_tag) TN_INL {
    constexpr int PROW = decltype(prow_tag)::value, E = decltype(e_tag)::value;
    constexpr int T = E / 5, I = E % 5, MB = T >> 1, R0 = 8 * (T & 1);
    unsigned (&epk)[4] = e_pk;             // (asm operands alone do not capture in a generic lambda)
    f32x16 (&accr)[4] = acc;
    if constexpr (I < 4) {
      const float a0 = accr[MB][R0 + 2 * I], a1 = accr[MB][R0 + 2 * I + 1];
      asm("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=v"(epk[I]) : "v"(a0), "v"(a1));
    } else {
      win_write<PROW, T>(epk[0], epk[1], epk[2], epk[3]);
    }
  };
  auto epilogue_a_exposed = [&](auto prow_tag) TN_INL {
    static_for<40>([&](auto e_tag) TN_INL { epa_item(prow_tag, e_tag); });
    TN_SB();
  };
  // ---- epilogue B: 24 items; output dword P = out channels 16 h + 2 P, + 1 of the lane's pixel: out[x] = acc[dx=0][x] +
  // acc[dx=-1][x-1] + acc[dx=+1][x+1], fp16; 16 B stored behind every fourth dword.  `off`: byte offset of the lane's 32 B (halo
  // lanes / no previous row: past the descriptor's range - the hardware drops the store, no branch) ----
  auto epb_item = [&](auto i_tag, unsigned off) TN_INL {
    constexpr int I = decltype(i_tag)::value, P = I / 3, PART = I % 3;
    if constexpr (PART == 0) {
      o_c[0] = bacc[1][2 * P]; o_c[1] = bacc[1][2 * P + 1];
      o_l[0] = bacc[0][2 * P]; o_l[1] = bacc[0][2 * P + 1];
    } else if constexpr (PART == 1) {
      o_c[0] += dpp_f32<0x111>(o_l[0]);     // row_shr:1: lane x reads lane x - 1
      o_c[1] += dpp_f32<0x111>(o_l[1]);
      o_r[0] = bacc[2][2 * P]; o_r[1] = bacc[2][2 * P + 1];
    } else {
      o_c[0] += dpp_f32<0x101>(o_r[0]);     // row_shl:1: lane x reads lane x + 1
      o_c[1] += dpp_f32<0x101>(o_r[1]);
      const h2_t p = {(f16)o_c[0], (f16)o_c[1]};
      o_pk[P] = __builtin_bit_cast(unsigned, p);
      if constexpr (P == 3 || P == 7) {
        const u32x4 o = {o_pk[P - 3], o_pk[P - 2], o_pk[P - 1], o_pk[P]};
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, off + (P == 7 ? 16 : 0), 0, 0);
      }
    }
  };
  auto out_offset = [&](int yo, bool valid) TN_INL { return (valid && store_ok) ? outb + (unsigned)yo * rowpitch : 0x80000000u; };
  // (belt and braces: the 3x3 phase's last statement already carries these wait states, mfma32_win_last3)
  auto bacc_ready = [&]() TN_INL {   // an asm MFMA's result may be read by anything but the next MFMA of its chain only 18+ wait states after issue
    f32x16 (&b)[3] = bacc;
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]));
  };
  auto epilogue_b_exposed = [&](int yo) TN_INL {
    bacc_ready();
    const unsigned off = out_offset(yo, true);
    static_for<24>([&](auto i_tag) TN_INL { epb_item(i_tag, off); TN_SB(); });
  };
  auto w3_item = [&](auto s_tag, auto dx_tag) TN_INL {
    constexpr int S = decltype(s_tag)::value, DX = decltype(dx_tag)::value;
    if ((TN_DS_EXP & 8) && S >= 2) return;      // (timing experiment: the 3x3 phase keeps reusing its first six weight fragments)
    w3f[S & 1][DX] = *(const u32x4 *)(w3l + (S * 3 + DX) * 1024);
  };

  // ================= 1x1 phase of bottleneck row yb (its first PLB pipeline items have run) =================
  // a slot: the MFMA, the reload of its weight register for k-step + 2, and ONE item: the next of the row's BN pipeline (when the
  // pipeline would otherwise fall behind the MFMAs) or the next of the previous output row's epilogue B (make_a_sched)
  auto phase_a = [&](int yb, int yo_prev, bool prev_valid) TN_INL {
    constexpr auto SA = make_a_sched<KS>();
    bacc_ready();
    const unsigned off_prev = out_offset(yo_prev, prev_valid);
    static_for<KQ + 1>([&](auto q_tag) TN_INL {
      constexpr int Q = decltype(q_tag)::value;
      static_for<4>([&](auto mb_tag) TN_INL {
        constexpr int MB = decltype(mb_tag)::value, SL = 4 * Q + MB;
        if constexpr (MB == 0) {     // one wait for the four weight fragments of the k-step (requested two k-steps ago); inputs only:
          u32x4 (&w)[4] = wa[Q & 1];   // an output would draw hipcc's asm boundary pad (s_nop) in front of the MFMA
          asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
        }
        if constexpr (Q == 0) {
          f32x16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          acc[MB] = mfma32(wa[0][MB], xb[0], z);
        } else if constexpr (Q == KQ) {
          acc[MB] = mfma32(wa[Q & 1][MB], xb_shift, acc[MB]);
        } else {
          acc[MB] = mfma32(wa[Q & 1][MB], xb[Q % XN], acc[MB]);
        }
        if constexpr (Q + 2 < KQ + 1) wa_item(ic<Q + 2>{}, mb_tag);
        constexpr ASlot sl = SA.s[SL];
        static_for<sl.npl>([&](auto k_tag) TN_INL { pl_item(ic<sl.pl0 + decltype(k_tag)::value>{}, yb); });
        static_for<sl.nepb>([&](auto k_tag) TN_INL { epb_item(ic<sl.epb0 + decltype(k_tag)::value>{}, off_prev); });
        TN_WIN_FENCE();
        TN_SB();
      });
    });
  };

  // ================= 3x3 phase of one output row: window rows (ROT + 1) % 3, (ROT + 2) % 3, ROT (the new one) =================
  // slot (dy, k-step, dx): the MFMA, the reload of its weight register for step + 2, and one item: slots 0 - 39 epilogue A of the
  // new row into window row ROT (first needed by slot 48), 40 / 41 the next row's first weight fragments, 42 - 71 the first PLB
  // items of the next row's BN pipeline
  auto phase_b = [&](auto rot_tag, auto epa_tag, int ybn) TN_INL {
    constexpr int ROT = decltype(rot_tag)::value;
    constexpr bool HAS_EPA = decltype(epa_tag)::value != 0;
    static_for<72>([&](auto e_tag) TN_INL {
      constexpr int E = decltype(e_tag)::value;
      constexpr int S = E / 3, DX = E % 3, DY = S / 8, T = S % 8;
      constexpr int PROW = (ROT + 1 + DY) % 3;
      if constexpr (DX == 0) {     // one wait for the step's three weight fragments
        u32x4 (&w)[3] = w3f[S & 1];
        asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]));
      }
      if constexpr (S < 23) mfma32_win<S == 0, PROW, T>(bacc[DX], w3f[S & 1][DX]);
      else if constexpr (DX == 2) mfma32_win_last3<PROW, T>(bacc[0], bacc[1], bacc[2], w3f[S & 1][0], w3f[S & 1][1], w3f[S & 1][2]);
      if constexpr (S + 2 < 24) w3_item(ic<S + 2>{}, ic<DX>{});
      if constexpr (E < 40) {
        if constexpr (HAS_EPA) epa_item(rot_tag, e_tag);
      } else if constexpr (E < 42) {
        wa_group(ic<E - 40>{});
      } else if constexpr (E - 42 < PLB) {
        pl_item(ic<E - 42>{}, ybn);
      }
      TN_WIN_FENCE();
      TN_SB();
    });
  };
  auto load_w3_first = [&]() TN_INL { static_for<6>([&](auto i_tag) TN_INL { w3_item(ic<decltype(i_tag)::value / 3>{}, ic<decltype(i_tag)::value % 3>{}); }); };

  int nstamp = 0;
  auto stamp = [&]() TN_INL {
    if (a.ts && wid == 0 && nstamp < 125) {
      if (lane == 0) a.ts[(size_t)blockIdx.x * 128 + nstamp] = __builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  // one steady-state row: bottleneck row yb (1x1 phase, into window row ROT through the 3x3 phase's fillers), output row yb - 1
  auto row_event = [&](auto rot_tag, int yb, bool prev_valid) TN_INL {
    stamp();
    if (!(TN_DS_EXP & 4)) phase_a(yb, yb - 2, prev_valid);
    load_w3_first();
    TN_SB();
    stamp();
    if (!(TN_DS_EXP & 2)) phase_b(rot_tag, ic<1>{}, yb + 1);
  };

  // ================= the wave's program =================
  {   // accumulators of the 3x3 start defined (the first 1x1 phases run an epilogue B whose store is dropped)
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    bacc[0] = z; bacc[1] = z; bacc[2] = z;
  }
  stamp();
  const int yfirst = r_lo > 0 ? r_lo - 1 : 0;
  static_for<NSU>([&](auto u_tag) TN_INL {
    constexpr bool HALF = ODD && decltype(u_tag)::value == NSU - 1;
    static_for<(HALF ? 2 : 4)>([&](auto i_tag) TN_INL { ring[decltype(u_tag)::value][decltype(i_tag)::value] =
        *(const u32x4 *)(rowbase(yfirst) + (HALF ? colh : colb) + 128 * decltype(u_tag)::value + 16 * decltype(i_tag)::value); });
  });
  // bottleneck row r_lo - 1 -> window row 0 (zeros above the image)
  if (r_lo > 0) {
    prologue_exposed(r_lo - 1);
    phase_a(r_lo - 1, 0, false);
    epilogue_a_exposed(ic<0>{});
  } else {
    win_zero<0>();
  }
  // bottleneck row r_lo -> window row 1
  prologue_exposed(r_lo);
  phase_a(r_lo, 0, false);
  epilogue_a_exposed(ic<1>{});
  prologue_exposed(r_lo + 1);
  stamp();
  // rows r_lo + 1 .. r_hi - 1: the steady state, window rotation 2, 0, 1, ...
  int yb = r_lo + 1;
  for (; yb + 2 < r_hi; yb += 3) {
    row_event(ic<2>{}, yb, yb > r_lo + 1);
    row_event(ic<0>{}, yb + 1, true);
    row_event(ic<1>{}, yb + 2, true);
  }
  constexpr int NREM = (ROWS - 1) % 3;          // steady-state rows left over
  if constexpr (NREM >= 1) { row_event(ic<2>{}, yb, yb > r_lo + 1); ++yb; }
  if constexpr (NREM >= 2) { row_event(ic<0>{}, yb, true); ++yb; }
  // bottleneck row r_hi (zeros below the image) -> window row (ROWS + 1) % 3, output row r_hi - 1
  constexpr int ROTL = (ROWS + 1) % 3;
  if (r_hi < H) {
    row_event(ic<ROTL>{}, r_hi, true);
  } else {
    epilogue_b_exposed(r_hi - 2);
    win_zero<ROTL>();
    load_w3_first();
    TN_SB();
    phase_b(ic<ROTL>{}, ic<0>{}, r_hi);
  }
  stamp();
  epilogue_b_exposed(r_hi - 1);
  stamp();
  if (a.ts && wid == 0 && lane == 0) a.ts[(size_t)blockIdx.x * 128 + 125] = __builtin_amdgcn_s_memrealtime();
}

template <int W, int KS>
int launch_strip(const DenseStripArgs &a, hipStream_t s) {
  using G = DSGeom<W, KS>;
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_strip_kernel<W, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)));
  hipLaunchKernelGGL((dense_strip_kernel<W, KS>), dim3(a.B * G::WGS), dim3(256), G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

template <int W>
int launch_strip_w(const DenseStripArgs &a, hipStream_t s) {
  switch (a.K / 32) {
    case 2: return launch_strip<W, 2>(a, s);
    case 3: return launch_strip<W, 3>(a, s);
    case 4: return launch_strip<W, 4>(a, s);
    case 5: return launch_strip<W, 5>(a, s);
    case 6: return launch_strip<W, 6>(a, s);
    case 7: return launch_strip<W, 7>(a, s);
    case 8: return launch_strip<W, 8>(a, s);
    case 9: return launch_strip<W, 9>(a, s);
    case 10:
      if constexpr (W != 128) return launch_strip<W, 10>(a, s);     // (W = 128: K <= 288, dense_strip_supported)
      break;
  }
  TN_REQUIRE(false, "dense_strip: K out of range");
}

}  // namespace

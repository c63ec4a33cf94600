// Fused DenseNet dense layer, "strip-streaming" form (round 3; 56x56 and 28x28 blocks):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// (reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_dense_layer: BatchNorm-Activation-
// Conv1x1-BatchNorm-Activation-Conv3x3-Concat).  dense_layer_big.hip owns ROUT image rows per 8-wave workgroup and runs its
// phases (K loop, epilogue, 3x3, store) one after the other behind barriers, with the bottleneck tile in LDS aliasing the
// K-loop ring.  This kernel is built the other way round:
//
// * a workgroup = one frame, 4 waves = one wave per SIMD with the whole 512-entry register file; NO barrier after the
//   prologue.  Each wave walks DOWN its own 14-pixel-wide column strip (16 slots with the two halo columns: exactly one
//   v_mfma_f32_16x16x32_f16 N-fragment per image row) in groups of four rows, so the four SIMDs drift apart and one
//   wave's VALU / LDS / load phases meet the others' MFMA phases.
// * the 128-channel bottleneck never touches LDS either: with the weights as the A operand, the accumulator layout of
//   the 1x1 GEMM (lane = pixel l & 15, rows 4 (l >> 4) + r) IS the B-operand layout of the 3x3's MFMAs once the 3x3
//   weights are packed with the matching permutation of their input channels (chained MFMAs: BN2 + ReLU + fp16 pack are
//   lane-local).  A wave keeps a sliding window of six bottleneck rows in registers (96 VGPRs).
// * the 3x3 convolution applies the three kernel columns to the SAME input fragment (one fragment feeds 6 MFMAs instead
//   of 2) into three accumulator sets which are combined at the end by two DPP row shifts: out[x] = acc[dx=0][x] +
//   acc[dx=-1][x-1] + acc[dx=+1][x+1].  The shifts stay inside the 16-lane row = inside the strip (outputs of the two
//   halo slots are never stored), so there is no cross-fragment carry.
// * the layer's 1x1 weights (K x 128, as A fragments) and all nine taps of the 3x3 weights (72 KB) are resident in LDS
//   for the whole launch; activations go HBM -> registers directly in fragment shape (32 B per lane per 64-channel
//   super-step; the four lanes of a pixel cover one 128-B line) through a register ring three super-steps deep that runs
//   one row-group ahead of the MFMAs.
//
// No halo recompute in y when a wave owns the frame's full height (56x56); 16/14 in x.
#include <type_traits>

#include "common.h"

#ifndef TN_DS_EXP
#define TN_DS_EXP 0   // timing experiments only (results wrong): bit 0 no activation loads inside the group loop, bit 1 no 3x3 phase, bit 2 no 1x1 phase
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kW3Bytes = 3 * 4 * 3 * 2 * 1024;   // [dy][t][dx][of] fragments of 1 KiB

template <int W, int KS>
struct DSGeom {
  static constexpr int NS = W / 14;              // strips per frame
  static constexpr int NV = 4 / NS;              // vertical parts (waves stacked in y)
  static constexpr int ROWS = W / NV;            // output rows per wave
  static constexpr int NG = ROWS / 4;            // full row groups
  static constexpr bool TAIL = (ROWS % 4) == 2;  // + one half group
  static constexpr int NSU = (KS + 1) / 2;       // 64-channel super-steps (the last one is half when KS is odd)
  static constexpr int W1OFF = kW3Bytes;
  static constexpr int T1OFF = W1OFF + KS * 8192;          // s1[K] | t1[K]
  static constexpr int T2OFF = T1OFF + KS * 32 * 8;        // s2[128] | t2[128]
  static constexpr int ZOFF = T2OFF + 1024;                // 1 KiB of zeros: BN2 "tables" of padding pixels
  static constexpr int LDS_BYTES = ZOFF + 1024;
  static_assert(W % 14 == 0 && (NS == 1 || NS == 2 || NS == 4), "strip geometry");
  static_assert(ROWS % 2 == 0, "rows per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "weights do not fit LDS");
};

__device__ __forceinline__ f32x4 mfma16(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// MFMA with the B operand (and the accumulator) in the accumulator half of the register file.  The builtin takes A / B from
// VGPRs only (hipcc copies an AGPR value back first), so the 3x3's MFMAs - whose B operands are the 96-register bottleneck
// window - are issued through inline asm; hipcc neither pads hazards around an asm statement nor knows it is an MFMA
// (cdna_hip_programming.md 5.7): the writers of the window end with s_nop 1, and the accumulators pass through
// mfma_results_ready() before anything but an MFMA of the same chain touches them.
// The bottleneck window lives in LITERAL accumulator registers a[160:255] (six rows x four 32-channel k-steps x one 4-register
// MFMA B operand): hipcc's MFMA builtin takes A / B from VGPRs only, and a window held in compiler-allocated AGPR values gets
// its live ranges split and copied through VGPRs (measured: ~200 extra v_accvgpr moves per row group, some of them directly
// in front of the asm MFMA that reads the register two cycles later - a hazard hipcc cannot see).  Every statement that
// writes the window names all 96 registers as clobbered, which keeps compiler values out of them and makes the kernel
// descriptor allocate them (cdna_hip_programming.md 5.7 item 4; scripts/audit_strip_isa.py checks the ISA for strays).
#define TN_WIN_BASE 160
#define TN_WIN_CLOBBER                                                                                                              \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175",   \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191",   \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207",   \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223",   \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239",   \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
constexpr int win_reg(int row, int t) { return TN_WIN_BASE + 16 * row + 4 * t; }

// four packed VGPRs -> window tuple (ROW, T); the trailing s_nop 1 covers v_accvgpr_write -> MFMA operand read
template <int ROW, int T>
__device__ __forceinline__ void win_write(const unsigned v0, const unsigned v1, const unsigned v2, const unsigned v3) {
  constexpr int B = win_reg(ROW, T);
  asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3\n\ts_nop 1"
               :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(B), "n"(B + 1), "n"(B + 2), "n"(B + 3) : TN_WIN_CLOBBER);
}
#define TN_MV(d, s_) "v_accvgpr_mov_b32 a" #d ", a" #s_ "\n\t"
// window rows 4, 5 -> rows 0, 1 (the next group's first two rows)
__device__ __forceinline__ void win_shift() {
  asm volatile(
      TN_MV(160, 224) TN_MV(161, 225) TN_MV(162, 226) TN_MV(163, 227) TN_MV(164, 228) TN_MV(165, 229) TN_MV(166, 230) TN_MV(167, 231)
      TN_MV(168, 232) TN_MV(169, 233) TN_MV(170, 234) TN_MV(171, 235) TN_MV(172, 236) TN_MV(173, 237) TN_MV(174, 238) TN_MV(175, 239)
      TN_MV(176, 240) TN_MV(177, 241) TN_MV(178, 242) TN_MV(179, 243) TN_MV(180, 244) TN_MV(181, 245) TN_MV(182, 246) TN_MV(183, 247)
      TN_MV(184, 248) TN_MV(185, 249) TN_MV(186, 250) TN_MV(187, 251) TN_MV(188, 252) TN_MV(189, 253) TN_MV(190, 254) TN_MV(191, 255)
      "s_nop 1" ::: TN_WIN_CLOBBER);
}
// window rows 2, 3 -> rows 4, 5 (start-up: the first two rows were computed as a half group)
template <int DST, int SRC>
__device__ __forceinline__ void win_copy2() {
  static_assert(DST == 4 && SRC == 2, "only the start-up copy exists");
  asm volatile(
      TN_MV(224, 192) TN_MV(225, 193) TN_MV(226, 194) TN_MV(227, 195) TN_MV(228, 196) TN_MV(229, 197) TN_MV(230, 198) TN_MV(231, 199)
      TN_MV(232, 200) TN_MV(233, 201) TN_MV(234, 202) TN_MV(235, 203) TN_MV(236, 204) TN_MV(237, 205) TN_MV(238, 206) TN_MV(239, 207)
      TN_MV(240, 208) TN_MV(241, 209) TN_MV(242, 210) TN_MV(243, 211) TN_MV(244, 212) TN_MV(245, 213) TN_MV(246, 214) TN_MV(247, 215)
      TN_MV(248, 216) TN_MV(249, 217) TN_MV(250, 218) TN_MV(251, 219) TN_MV(252, 220) TN_MV(253, 221) TN_MV(254, 222) TN_MV(255, 223)
      "s_nop 1" ::: TN_WIN_CLOBBER);
}
#undef TN_MV
// one 3x3 weight fragment against window tuples (ROW, T) and (ROW + 1, T): the two MFMAs of a slot
template <bool FIRST, int ROW, int T>
__device__ __forceinline__ void mfma16_pair(f32x4 &d0, f32x4 &d1, const u32x4 a) {
  constexpr int B0 = win_reg(ROW, T), B1 = win_reg(ROW + 1, T);
  if constexpr (FIRST)
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, a[%c3:%c4], 0\n\tv_mfma_f32_16x16x32_f16 %1, %2, a[%c5:%c6], 0"
                 : "=&a"(d0), "=&a"(d1) : "v"(a), "n"(B0), "n"(B0 + 3), "n"(B1), "n"(B1 + 3));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, a[%c3:%c4], %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, a[%c5:%c6], %1"
                 : "+a"(d0), "+a"(d1) : "v"(a), "n"(B0), "n"(B0 + 3), "n"(B1), "n"(B1 + 3));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int V>
using ic = std::integral_constant<int, V>;

// The schedule is pinned by hand: hipcc's scheduler, left alone, issues every LDS read right in front of its consumer and
// piles the element-wise epilogues up in MFMA-free stretches.  The body below is written as a sequence of SLOTS - one MFMA
// (1x1 phase) or one two-MFMA block (3x3 phase) followed by its share of everything else - with a scheduling barrier
// behind each, so that program order IS issue order.
#define TN_SB() __builtin_amdgcn_sched_barrier(0)

template <int W, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_strip_kernel(DenseStripArgs a) {
  using G = DSGeom<W, KS>;
  constexpr int H = W, NSU = G::NSU, K = KS * 32;
  constexpr bool ODD = (KS & 1) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;

  // ---- prologue: the layer's weights and tables -> LDS (once per launch) ----
  {
    const uint4 *g3 = (const uint4 *)a.w3s;
    uint4 *l3 = (uint4 *)smem;
#pragma unroll 6
    for (int i = tid; i < kW3Bytes / 16; i += 256) l3[i] = g3[i];
    const uint4 *g1 = (const uint4 *)a.w1s;
    uint4 *l1 = (uint4 *)(smem + G::W1OFF);
#pragma unroll 4
    for (int i = tid; i < KS * 512; i += 256) l1[i] = g1[i];
    float *t1 = (float *)(smem + G::T1OFF);
    for (int i = tid; i < K; i += 256) {
      t1[i] = a.s1[i];
      t1[K + i] = a.t1[i];
    }
    float *t2 = (float *)(smem + G::T2OFF);
    if (tid < 128) {
      t2[tid] = a.s2[tid];
      t2[128 + tid] = a.t2[tid];
    }
    ((float *)(smem + G::ZOFF))[tid] = 0.f;
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g4 = lane >> 4;
  const int strip = wid % G::NS, part = wid / G::NS;
  const int r_lo = part * G::ROWS;
  const int x = 14 * strip - 1 + n;
  const bool xvalid = x >= 0 && x < W;
  const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
  const int ldc = a.ldc;
  const unsigned rowpitch = (unsigned)W * ldc * 2;
  unsigned char *fb = (unsigned char *)(a.buf + (size_t)blockIdx.x * H * W * ldc);
  const unsigned colb = (unsigned)xc * ldc * 2 + 32 * g4;     // full super-steps: 32 B per lane
  const unsigned colh = (unsigned)xc * ldc * 2 + 16 * g4;     // the trailing half super-step: 16 B per lane
  const bool store_ok = n >= 1 && n <= 14;
  const unsigned outb = store_ok ? (unsigned)xc * ldc * 2 + K * 2 + 16 * g4 : 0x80000000u;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(fb, 0, (int)((unsigned)H * rowpitch), 0x00020000);

  const unsigned char *w1l = smem + G::W1OFF + lane * 16;
  const unsigned char *w3l = smem + lane * 16;
  const float *tab1 = (const float *)(smem + G::T1OFF);
  const unsigned tab2_lane = xvalid ? (unsigned)(G::T2OFF + 16 * g4) : (unsigned)G::ZOFF;

  // ================= state that lives across slots =================
  u32x4 ring[3][4][2];   // activation ring [slot][row][k-step of the super-step]: 16 B per lane = 8 channels of the row's pixel
  // (the bottleneck window - row Y - 1 + i of the group of output rows Y .. Y + 3 in window row i - lives in a[160:255])
  f32x4 acc[4][8];       // 1x1 accumulators [row][16-channel fragment]
  u32x4 wa[8];           // 1x1 weight fragments of the current k-step (reloaded for the next one behind their last use)
  u32x4 xb[2][4];        // BN1 + ReLU'd pixel fragments [k-step parity][row]
  float cs[2][8], ct[2][8];   // BN1 constants [k-step parity]
  float bnt0 = 0.f, bnt1 = 0.f;
  u32x4 w3f[6];          // 3x3 weight fragments [dx * 2 + of] of the current (dy, t) step
  f32x4 bacc[2][2][3][2];   // 3x3 accumulators [row pair][row][dx][of]
  float4 e_sv[2], e_tv[2];  // epilogue A: BN2 constants [fragment parity]
  float e_f[4];
  unsigned e_pk[4];
  float o_v[4];
  u32x4 o_pk;

  auto rowbase = [&](int y) __attribute__((always_inline)) {
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return fb + (unsigned)yc * rowpitch;
  };
  // one 16-byte activation load: item i of the 8 (4 for the trailing half super-step) of ring slot SLOT <- super-step U, rows y0 ..
  auto ld_item = [&](auto slot_tag, auto u_tag, auto i_tag, int y0) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value, U = decltype(u_tag)::value, I = decltype(i_tag)::value;
    constexpr bool HALF = ODD && U == NSU - 1;
    if ((TN_DS_EXP & 1) && y0 > r_lo) return;
    if constexpr (HALF) {
      if constexpr (I < 4) ring[SLOT][I][0] = *(const u32x4 *)(rowbase(y0 + I) + colh + 128 * U);
    } else {
      ring[SLOT][I >> 1][I & 1] = *(const u32x4 *)(rowbase(y0 + (I >> 1)) + colb + 128 * U + 16 * (I & 1));
    }
  };
  auto consts_item = [&](auto q_tag, auto p_tag) __attribute__((always_inline)) {     // one ds_read_b128 of k-step Q's BN1 constants (P: s lo, s hi, t lo, t hi)
    constexpr int Q = decltype(q_tag)::value, P = decltype(p_tag)::value;
    constexpr int U = Q >> 1, I = Q & 1;
    constexpr bool half = ODD && Q == KS - 1;
    const int c0 = (half ? 64 * U + 8 * g4 : 64 * U + 16 * g4 + 8 * I) + (P & 1) * 4 + (P >> 1) * K;
    const float4 v = *(const float4 *)(tab1 + c0);
    float *d = (P >> 1) ? ct[Q & 1] : cs[Q & 1];
    d[(P & 1) * 4 + 0] = v.x; d[(P & 1) * 4 + 1] = v.y; d[(P & 1) * 4 + 2] = v.z; d[(P & 1) * 4 + 3] = v.w;
  };
  auto wa_item = [&](auto q_tag, auto mf_tag) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value, MF = decltype(mf_tag)::value;
    wa[MF] = *(const u32x4 *)(w1l + (Q * 8 + MF) * 1024);
  };
  // BN1 + ReLU micro-item I of k-step Q (two VALU instructions): row I >> 3, dword (I >> 1) & 3, first / second half
  auto bn_item = [&](auto q_tag, auto i_tag) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value, I = decltype(i_tag)::value;
    constexpr int U = Q >> 1, SLOT = U % 3, R = I >> 3, J = (I >> 1) & 3;
    float &t0 = bnt0, &t1 = bnt1;     // (asm operands alone do not capture in a generic lambda)
    if constexpr ((I & 1) == 0) {
      const unsigned in = ring[SLOT][R][Q & 1][J];
      const float s0 = cs[Q & 1][2 * J], s1 = cs[Q & 1][2 * J + 1], h0 = ct[Q & 1][2 * J], h1 = ct[Q & 1][2 * J + 1];
      asm("v_fma_mix_f32 %0, %2, %3, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %5, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
          : "=&v"(t0), "=&v"(t1) : "v"(in), "v"(s0), "v"(h0), "v"(s1), "v"(h1));
    } else {
      unsigned o;
      asm("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=v"(o) : "v"(t0), "v"(t1));
      xb[Q & 1][R][J] = o;
    }
  };
  // everything the first k-step of a group needs, as 16 + 8 NR items: constants of k-steps 0 and 1, the weight fragments and
  // the BN'd pixel fragments of k-step 0 (run as fillers of the PREVIOUS group's 3x3 phase, or exposed at the start)
  auto pro_item = [&](auto nr_tag, auto i_tag) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_tag)::value, I = decltype(i_tag)::value;
    if constexpr (I < 4) consts_item(ic<0>{}, ic<I>{});
    else if constexpr (I < 8) { if constexpr (KS > 1) consts_item(ic<1>{}, ic<I - 4>{}); }
    else if constexpr (I < 16) wa_item(ic<0>{}, ic<I - 8>{});
    else if constexpr (I < 16 + 8 * NR) bn_item(ic<0>{}, ic<I - 16>{});
  };

  // ---- epilogue A items: acc rows R0, R0 + 1 -> win[2 + R0], win[3 + R0]; 36 items per row ----
  unsigned e_tb = 0;
  auto epa_consts = [&](auto mf_tag, auto buf_tag) __attribute__((always_inline)) {
    constexpr int MF = decltype(mf_tag)::value, B = decltype(buf_tag)::value;
    e_sv[B] = *(const float4 *)(smem + e_tb + 64 * MF);
    e_tv[B] = *(const float4 *)(smem + e_tb + 512 + 64 * MF);
  };
  auto epa_set_row = [&](int y) __attribute__((always_inline)) { e_tb = (y >= 0 && y < H) ? tab2_lane : (unsigned)G::ZOFF; };   // rows above / below the image are zero
  auto epa_item = [&](auto r0_tag, auto i_tag, int ynew) __attribute__((always_inline)) {
    constexpr int R0 = decltype(r0_tag)::value, I = decltype(i_tag)::value;
    constexpr int R = R0 + I / 36, KI = I % 36, T = KI / 9, J = KI % 9;
    float (&ef)[4] = e_f;             // (asm operands alone do not capture in a generic lambda)
    unsigned (&epk)[4] = e_pk;
    f32x4 (&accr)[4][8] = acc;
    if constexpr (J == 8) {
      win_write<2 + R, T>(epk[0], epk[1], epk[2], epk[3]);
    } else {
      constexpr int E = J >> 2, P = J & 3, MF = 2 * T + E;
      if constexpr (P == 0) {            // constants of the NEXT fragment (of the next row behind the last one)
        if constexpr (MF < 7) epa_consts(ic<MF + 1>{}, ic<(MF + 1) & 1>{});
        else if constexpr (I / 36 == 0) { epa_set_row(ynew + R + 1); epa_consts(ic<0>{}, ic<0>{}); }
      } else if constexpr (P == 1 || P == 2) {
        constexpr int C = (P - 1) * 2;
        const float4 sv = e_sv[MF & 1], tv = e_tv[MF & 1];
        const float s0 = C ? sv.z : sv.x, s1 = C ? sv.w : sv.y, t0 = C ? tv.z : tv.x, t1 = C ? tv.w : tv.y;
        const float a0 = accr[R][MF][C], a1 = accr[R][MF][C + 1];
        asm("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %6, %7" : "=&v"(ef[C]), "=&v"(ef[C + 1]) : "v"(a0), "v"(s0), "v"(t0), "v"(a1), "v"(s1), "v"(t1));
      } else {
        asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\tv_pk_max_f16 %0, %0, 0\n\tv_pk_max_f16 %1, %1, 0"
            : "=&v"(epk[2 * E]), "=&v"(epk[2 * E + 1]) : "v"(ef[0]), "v"(ef[1]), "v"(ef[2]), "v"(ef[3]));
      }
    }
  };
  // ---- epilogue B items: output rows of row pair RP (12 items per row): out[x] = acc[dx=0][x] + acc[dx=-1][x-1] + acc[dx=+1][x+1] ----
  auto epb_item = [&](auto rp_tag, auto i_tag, int y0) __attribute__((always_inline)) {
    constexpr int RP = decltype(rp_tag)::value, I = decltype(i_tag)::value;
    constexpr int R = I / 12, KI = I % 12, OF = KI / 6, J = KI % 6;
    if constexpr (J == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) o_v[c] = bacc[RP][R][1][OF][c];
    } else if constexpr (J <= 4) {
      constexpr int DX = J <= 2 ? 0 : 2, C = ((J - 1) & 1) * 2;
      if constexpr (DX == 0) {
        o_v[C] += dpp_f32<0x111>(bacc[RP][R][0][OF][C]);           // row_shr:1: lane x reads lane x - 1
        o_v[C + 1] += dpp_f32<0x111>(bacc[RP][R][0][OF][C + 1]);
      } else {
        o_v[C] += dpp_f32<0x101>(bacc[RP][R][2][OF][C]);           // row_shl:1: lane x reads lane x + 1
        o_v[C + 1] += dpp_f32<0x101>(bacc[RP][R][2][OF][C + 1]);
      }
    } else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 p0 = {(f16)o_v[0], (f16)o_v[1]}, p1 = {(f16)o_v[2], (f16)o_v[3]};
      o_pk[OF * 2] = __builtin_bit_cast(unsigned, p0);
      o_pk[OF * 2 + 1] = __builtin_bit_cast(unsigned, p1);
      // (the two halo lanes carry an offset past the descriptor's range: the hardware drops their store, no branch)
      if constexpr (OF == 1) __builtin_amdgcn_raw_buffer_store_b128(o_pk, orsrc, outb + (unsigned)(y0 + 2 * RP + R) * rowpitch, 0, 0);
    }
  };
  auto w3_item = [&](auto s_tag, auto f_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(s_tag)::value, F = decltype(f_tag)::value;
    w3f[F] = *(const u32x4 *)(w3l + (S * 6 + F) * 1024);
  };

  // ================= 1x1 phase: NR new bottleneck rows ynew ..; reloads the ring for this group / the group at ynext =================
  auto phase_a = [&](auto nr_tag, int ynew, int ynext) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_tag)::value;
    static_for<KS>([&](auto q_tag) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_tag)::value;
      // the ring slot whose last k-step is Q has been read out (its BN items ran during step Q - 1): it is reloaded during this step
      constexpr int UL = Q >> 1;
      constexpr bool RELOAD = (Q & 1) == 1 || (ODD && Q == KS - 1);
      static_for<8 * NR>([&](auto i_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value;
        constexpr int MF = I / NR, R = I % NR;
        if constexpr (Q == 0) acc[R][MF] = mfma16(wa[MF], xb[0][R], (f32x4){0.f, 0.f, 0.f, 0.f});
        else acc[R][MF] = mfma16(wa[MF], xb[Q & 1][R], acc[R][MF]);
        if constexpr (Q + 1 < KS) {
          bn_item(ic<Q + 1>{}, i_tag);
          if constexpr (R == NR - 1) wa_item(ic<Q + 1>{}, ic<MF>{});
        }
        if constexpr (Q + 2 < KS && (I % (2 * NR)) == 1) consts_item(ic<Q + 2>{}, ic<I / (2 * NR)>{});
        if constexpr (RELOAD && (I % NR) == (NR > 2 ? 2 : 0) && I / NR < 8) {
          if constexpr (UL + 3 < NSU) ld_item(ic<UL % 3>{}, ic<UL + 3>{}, ic<I / NR>{}, ynew);
          else ld_item(ic<UL % 3>{}, ic<UL % 3>{}, ic<I / NR>{}, ynext);
        }
        TN_SB();
      });
    });
  };
  // epilogue A of rows R0, R0 + 1, exposed (no MFMAs beside it)
  auto epilogue_a = [&](auto r0_tag, int ynew) __attribute__((always_inline)) {
    constexpr int R0 = decltype(r0_tag)::value;
    epa_set_row(ynew + R0);
    epa_consts(ic<0>{}, ic<0>{});
    static_for<72>([&](auto i_tag) __attribute__((always_inline)) { epa_item(r0_tag, i_tag, ynew + R0 - decltype(r0_tag)::value); });
  };

  // ================= 3x3 phase: NR output rows y0 .. from win[0 .. NR + 1] =================
  // row-pair-major: the two MFMAs of a slot apply one weight fragment to the two rows of the pair.  Fillers of the first pair:
  // epilogue A of the 1x1 rows 2, 3 (their window rows are first needed by the second pair); of the second pair (of the only
  // pair of a half group): epilogue B of the first pair, then the next group's 1x1 prologue.
  auto phase_b = [&](auto nr_tag, auto nrnext_tag, int y0, int ynew) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_tag)::value, NRNEXT = decltype(nrnext_tag)::value;
    constexpr int NP = NR / 2;
    static_for<NP>([&](auto rp_tag) __attribute__((always_inline)) {
      constexpr int RP = decltype(rp_tag)::value;
      f32x4 (&bq)[2][2][3][2] = bacc;
      if constexpr (RP == 1) {      // the first pair's accumulators are read by this pair's fillers
        asm volatile("s_nop 7" : "+a"(bq[0][0][0][0]), "+a"(bq[0][0][0][1]), "+a"(bq[0][0][1][0]), "+a"(bq[0][0][1][1]), "+a"(bq[0][0][2][0]), "+a"(bq[0][0][2][1]),
                                 "+a"(bq[0][1][0][0]), "+a"(bq[0][1][0][1]), "+a"(bq[0][1][1][0]), "+a"(bq[0][1][1][1]), "+a"(bq[0][1][2][0]), "+a"(bq[0][1][2][1]));
      }
      static_for<72>([&](auto s_tag) __attribute__((always_inline)) {
        constexpr int SL = decltype(s_tag)::value;
        constexpr int S = SL / 6, F = SL % 6, DY = S / 4, T = S % 4;
        mfma16_pair<S == 0, 2 * RP + DY, T>(bacc[RP][0][F >> 1][F & 1], bacc[RP][1][F >> 1][F & 1], w3f[F]);
        // this fragment's register takes the fragment of the next step (of the next pair's first step)
        if constexpr (S < 11) w3_item(ic<S + 1>{}, ic<F>{});
        else if constexpr (RP + 1 < NP) w3_item(ic<0>{}, ic<F>{});
        if constexpr (NP == 2 && RP == 0) {
          epa_item(ic<2>{}, s_tag, ynew);
        } else {
          constexpr int PRO0 = NP == 2 ? 24 : 0;        // (a half group: epilogue B runs exposed behind the phase)
          if constexpr (NP == 2 && SL < 24) epb_item(ic<0>{}, s_tag, y0);
          else if constexpr (NRNEXT > 0 && SL - PRO0 < 16 + 8 * NRNEXT) pro_item(nrnext_tag, ic<SL - PRO0>{});
        }
        TN_SB();
      });
    });
    // the last pair's accumulators: wait, then epilogue B exposed
    constexpr int LP = NP - 1;
    asm volatile("s_nop 15" : "+a"(bacc[LP][0][0][0]), "+a"(bacc[LP][0][0][1]), "+a"(bacc[LP][0][1][0]), "+a"(bacc[LP][0][1][1]), "+a"(bacc[LP][0][2][0]), "+a"(bacc[LP][0][2][1]),
                              "+a"(bacc[LP][1][0][0]), "+a"(bacc[LP][1][0][1]), "+a"(bacc[LP][1][1][0]), "+a"(bacc[LP][1][1][1]), "+a"(bacc[LP][1][2][0]), "+a"(bacc[LP][1][2][1]));
    static_for<24>([&](auto i_tag) __attribute__((always_inline)) { epb_item(ic<LP>{}, i_tag, y0); });
  };
  auto load_w3_first = [&]() __attribute__((always_inline)) { static_for<6>([&](auto f_tag) __attribute__((always_inline)) { w3_item(ic<0>{}, f_tag); }); };
  // one group of NR output rows y .. (new bottleneck rows y + 1 ..), the next group has NRNEXT rows (0: none)
  auto group = [&](auto nr_tag, auto nrnext_tag, int y) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_tag)::value;
    win_shift();
    TN_SB();
    if (!(TN_DS_EXP & 4)) phase_a(nr_tag, y + 1, y + 1 + NR);
    load_w3_first();
    if (!(TN_DS_EXP & 4)) epilogue_a(ic<0>{}, y + 1);
    TN_SB();
    if (!(TN_DS_EXP & 2)) phase_b(nr_tag, nrnext_tag, y, y + 1);
    TN_SB();
  };

  // ================= the wave's program: rows r_lo - 1, r_lo first, then groups of four output rows (+ a half group) =================
  static_for<3>([&](auto u_tag) __attribute__((always_inline)) {
    constexpr int U = decltype(u_tag)::value;
    if constexpr (U < NSU) static_for<8>([&](auto i_tag) __attribute__((always_inline)) { ld_item(u_tag, u_tag, i_tag, r_lo - 1); });
  });
  static_for<16 + 16>([&](auto i_tag) __attribute__((always_inline)) { pro_item(ic<2>{}, i_tag); });
  TN_SB();
  phase_a(ic<2>{}, r_lo - 1, r_lo + 1);
  epilogue_a(ic<0>{}, r_lo - 1);
  win_copy2<4, 2>();
  constexpr int NFIRST = G::NG > 0 ? 4 : 2;
  static_for<16 + 8 * NFIRST>([&](auto i_tag) __attribute__((always_inline)) { pro_item(ic<NFIRST>{}, i_tag); });
  TN_SB();
  int y = r_lo;
  for (int g = 0; g + 1 < G::NG; ++g, y += 4) group(ic<4>{}, ic<4>{}, y);
  if constexpr (G::NG > 0) {
    group(ic<4>{}, ic<(G::TAIL ? 2 : 0)>{}, y);
    y += 4;
  }
  if constexpr (G::TAIL) group(ic<2>{}, ic<0>{}, y);
}

template <int W, int KS>
int launch_strip(const DenseStripArgs &a, hipStream_t s) {
  using G = DSGeom<W, KS>;
  static bool attr_set = false;
  if (!attr_set) {
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_strip_kernel<W, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL((dense_strip_kernel<W, KS>), dim3(a.B), dim3(256), G::LDS_BYTES, s, a);
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
    case 10: return launch_strip<W, 10>(a, s);
  }
  TN_REQUIRE(false, "dense_strip: K out of range");
}

}  // namespace

bool dense_strip_supported(int H, int W, int K) {
  return H == W && (W == 56 || W == 28) && K % 32 == 0 && K >= 64 && K <= 320;
}

int launch_dense_strip(const DenseStripArgs &a, hipStream_t s) {
  TN_REQUIRE(dense_strip_supported(a.H, a.W, a.K), "dense_strip: unsupported geometry");
  TN_REQUIRE(a.ldc % 64 == 0 && a.K + 32 <= a.ldc, "dense_strip: bad channel geometry");
  if (a.W == 56) return launch_strip_w<56>(a, s);
  return launch_strip_w<28>(a, s);
}

// ---- host-side packing (api.hip, dbg.hip) ----
// 1x1 weights [128][K] -> A fragments [K/32 k-steps][8 m-frags][64 lanes][8]: lane l: bottleneck channel 16 mf + (l & 15),
// input channels of 64-channel super-step u, k-step i: 64 u + 16 (l >> 4) + 8 i + j (a lane's two k-steps are 32 contiguous
// bytes of the pixel); the trailing 32-channel step of an odd K/32: 64 u + 8 (l >> 4) + j.
std::vector<f16> pack_w1_strip(const float *w, int K) {
  const int ks = K / 32;
  std::vector<f16> p((size_t)ks * 8 * 64 * 8);
  for (int q = 0; q < ks; ++q) {
    const int u = q >> 1, i = q & 1;
    const bool half = (ks & 1) && q == ks - 1;
    for (int mf = 0; mf < 8; ++mf)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int c = half ? 64 * u + 8 * (l >> 4) + j : 64 * u + 16 * (l >> 4) + 8 * i + j;
          p[(((size_t)q * 8 + mf) * 64 + l) * 8 + j] = (f16)w[(size_t)(16 * mf + (l & 15)) * K + c];
        }
  }
  return p;
}

// 3x3 weights (32,128,3,3) -> A fragments [3 dy][4 t][3 dx][2 of][64 lanes][8]: lane l, m = l & 15: output channel
// 8 (m >> 2) + 4 of + (m & 3) (so that a lane of the result holds 8 consecutive output channels of its pixel), bottleneck
// channel 32 t + 16 (j >> 2) + 4 (l >> 4) + (j & 3): the order in which the 1x1's accumulators hold them.
std::vector<f16> pack_w3_strip(const float *w) {
  std::vector<f16> p((size_t)kW3Bytes / 2);
  for (int dy = 0; dy < 3; ++dy)
    for (int t = 0; t < 4; ++t)
      for (int dx = 0; dx < 3; ++dx)
        for (int of = 0; of < 2; ++of)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int m = l & 15, o = 8 * (m >> 2) + 4 * of + (m & 3);
              const int c = 32 * t + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
              p[((((((size_t)dy * 4 + t) * 3 + dx) * 2 + of) * 64) + l) * 8 + j] = (f16)w[(((size_t)o * 128 + c) * 3 + dy) * 3 + dx];
            }
  return p;
}

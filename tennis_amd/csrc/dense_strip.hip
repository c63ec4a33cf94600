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
// One weight fragment against NR window rows.  The leading s_nop 1 covers a compiler-generated v_accvgpr_write (live-range
// split or tuple copy of a window register) directly in front of the statement: hipcc cannot know that the statement reads
// the register as an MFMA operand two cycles later.
template <bool FIRST, int NR>
__device__ __forceinline__ void mfma16_rows(f32x4 *d0, f32x4 *d1, f32x4 *d2, f32x4 *d3, const u32x4 a, const u32x4 b0, const u32x4 b1,
                                            const u32x4 b2, const u32x4 b3) {
  if constexpr (NR == 4) {
    if constexpr (FIRST)
      asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %4, %5, 0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %6, 0\n\t"
          "v_mfma_f32_16x16x32_f16 %2, %4, %7, 0\n\tv_mfma_f32_16x16x32_f16 %3, %4, %8, 0"
          : "=&a"(*d0), "=&a"(*d1), "=&a"(*d2), "=&a"(*d3) : "v"(a), "a"(b0), "a"(b1), "a"(b2), "a"(b3));
    else
      asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %6, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %8, %3"
          : "+a"(*d0), "+a"(*d1), "+a"(*d2), "+a"(*d3) : "v"(a), "a"(b0), "a"(b1), "a"(b2), "a"(b3));
  } else {
    if constexpr (FIRST)
      asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, 0"
          : "=&a"(*d0), "=&a"(*d1) : "v"(a), "a"(b0), "a"(b1));
    else
      asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %1"
          : "+a"(*d0), "+a"(*d1) : "v"(a), "a"(b0), "a"(b1));
  }
}
// four packed VGPRs -> one AGPR tuple (an MFMA B operand); the trailing s_nop 1 covers v_accvgpr_write -> MFMA operand read
__device__ __forceinline__ u32x4 to_agpr(const unsigned v0, const unsigned v1, const unsigned v2, const unsigned v3) {
  unsigned a0, a1, a2, a3;
  asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7\n\ts_nop 1"
               : "=a"(a0), "=a"(a1), "=a"(a2), "=a"(a3)
               : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
  return (u32x4){a0, a1, a2, a3};
}

// An asm MFMA's result may be read by something other than the next MFMA of its chain only 12+ wait states after issue
// (8-pass XDL op; hipcc pads nothing for an asm producer).  Every accumulator of the 3x3 goes through one of these
// statements (each carries its own wait: the MFMAs of the second half may be scheduled behind the first statement) before
// the DPP epilogue reads it.
template <int N>
__device__ __forceinline__ void mfma_results_ready(f32x4 (&b)[N][3][2]) {
  static_assert(N == 2 || N == 4, "rows");
  asm volatile("s_nop 15" : "+a"(b[0][0][0]), "+a"(b[0][0][1]), "+a"(b[0][1][0]), "+a"(b[0][1][1]), "+a"(b[0][2][0]), "+a"(b[0][2][1]),
                            "+a"(b[1][0][0]), "+a"(b[1][0][1]), "+a"(b[1][1][0]), "+a"(b[1][1][1]), "+a"(b[1][2][0]), "+a"(b[1][2][1]));
  if constexpr (N == 4)
    asm volatile("s_nop 15" : "+a"(b[2][0][0]), "+a"(b[2][0][1]), "+a"(b[2][1][0]), "+a"(b[2][1][1]), "+a"(b[2][2][0]), "+a"(b[2][2][1]),
                      "+a"(b[3][0][0]), "+a"(b[3][0][1]), "+a"(b[3][1][0]), "+a"(b[3][1][1]), "+a"(b[3][2][0]), "+a"(b[3][2][1]));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

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

  // ---- activation ring: [slot][row][i]: 16 B per lane = channels 64 u + 16 g4 + 8 i + (0..7) of the row's pixel ----
  u32x4 ring[3][4][2];
  auto rowbase = [&](int y) {
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return fb + (unsigned)yc * rowpitch;
  };
  auto issue = [&](auto slot_tag, auto u_tag, int y0) {
    constexpr int SLOT = decltype(slot_tag)::value, U = decltype(u_tag)::value;
    constexpr bool HALF = ODD && U == NSU - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned char *p = rowbase(y0 + r);
      if constexpr (HALF) {
        ring[SLOT][r][0] = *(const u32x4 *)(p + colh + 128 * U);
      } else {
        ring[SLOT][r][0] = *(const u32x4 *)(p + colb + 128 * U);
        ring[SLOT][r][1] = *(const u32x4 *)(p + colb + 128 * U + 16);
      }
    }
  };

  // window of bottleneck rows (B operands of the 3x3): win[i] = row Y - 1 + i of the current group of output rows Y .. Y+3
  u32x4 win[6][4];

  // ---- phase A: NR new bottleneck rows ynew .. ynew + NR - 1 -> win[2 .. 2 + NR - 1]; prefetches the group at ynext ----
  auto phase_a = [&](auto nr_tag, int ynew, int ynext) {
    constexpr int NR = decltype(nr_tag)::value;
    f32x4 acc[NR][8];
    u32x4 wa[8];
    u32x4 xb[2][NR];
    float cs[8], ct[8];
    auto load_consts = [&](int q) {        // BN1 constants of k-step q for this lane's 8 channels
      const int u = q >> 1, i = q & 1;
      const bool half = ODD && q == KS - 1;
      const int c0 = half ? 64 * u + 8 * g4 : 64 * u + 16 * g4 + 8 * i;
      const float4 s0 = *(const float4 *)(tab1 + c0), s1 = *(const float4 *)(tab1 + c0 + 4);
      const float4 t0 = *(const float4 *)(tab1 + K + c0), t1 = *(const float4 *)(tab1 + K + c0 + 4);
      cs[0] = s0.x; cs[1] = s0.y; cs[2] = s0.z; cs[3] = s0.w; cs[4] = s1.x; cs[5] = s1.y; cs[6] = s1.z; cs[7] = s1.w;
      ct[0] = t0.x; ct[1] = t0.y; ct[2] = t0.z; ct[3] = t0.w; ct[4] = t1.x; ct[5] = t1.y; ct[6] = t1.z; ct[7] = t1.w;
    };
    auto bn_row = [&](auto q_tag, int r, u32x4 &dst) {
      constexpr int Q = decltype(q_tag)::value;
      constexpr int U = Q >> 1, I = Q & 1, SLOT = U % 3;
      const u32x4 raw = ring[SLOT][r][I];
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[j] = bn_relu2_mix(raw[j], cs[2 * j], cs[2 * j + 1], ct[2 * j], ct[2 * j + 1]);
    };
    auto load_wa = [&](int q, int mf) { wa[mf] = *(const u32x4 *)(w1l + (q * 8 + mf) * 1024); };

    // operands of k-step 0
    load_consts(0);
#pragma unroll
    for (int mf = 0; mf < 8; ++mf) load_wa(0, mf);
#pragma unroll
    for (int r = 0; r < NR; ++r) bn_row(std::integral_constant<int, 0>{}, r, xb[0][r]);

    auto kstep = [&](auto q_tag) {
      constexpr int Q = decltype(q_tag)::value;
      constexpr int CUR = Q & 1, NXT = CUR ^ 1;
      constexpr bool LAST = Q == KS - 1;
      if constexpr (!LAST) load_consts(Q + 1);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if constexpr (Q == 0) acc[r][mf] = mfma16(wa[mf], xb[CUR][r], (f32x4){0.f, 0.f, 0.f, 0.f});
          else acc[r][mf] = mfma16(wa[mf], xb[CUR][r], acc[r][mf]);
        }
        if constexpr (!LAST) {
          load_wa(Q + 1, mf);
          // BN + ReLU of the next k-step's pixel fragments in the shadow of these MFMAs
          if (mf >= 1 && mf - 1 < NR) bn_row(std::integral_constant<int, (LAST ? Q : Q + 1)>{}, mf - 1, xb[NXT][mf - 1]);
        }
      }
      // the super-step this k-step closes is consumed: its ring slot takes the next one (this group's, or the next group's)
      constexpr int U = Q >> 1;
      constexpr bool CLOSES = (Q & 1) == 1 || (ODD && LAST);
      if constexpr (CLOSES) {
        // (the BN of k-step Q + 1 above reads the NEXT super-step's slot, never this one)
        if constexpr (U + 3 < NSU) issue(std::integral_constant<int, U % 3>{}, std::integral_constant<int, U + 3>{}, ynew);
        else issue(std::integral_constant<int, U % 3>{}, std::integral_constant<int, U % 3>{}, ynext);
      }
    };
    [&]<int... Q>(std::integer_sequence<int, Q...>) { (kstep(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, KS>{});

    // ---- epilogue A: BN2 + ReLU (fp32), one rounding to fp16, lane-local pack into the 3x3's B-operand layout ----
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int y = ynew + r;
      const unsigned tb = (y >= 0 && y < H) ? tab2_lane : (unsigned)G::ZOFF;     // rows above / below the image are zero
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 pk[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int mf = 2 * t + e;
          const float4 sv = *(const float4 *)(smem + tb + 64 * mf), tv = *(const float4 *)(smem + tb + 512 + 64 * mf);
          pk[e] = __builtin_bit_cast(u32x2, bn_relu4_from_f32(acc[r][mf], sv, tv));
        }
        win[2 + r][t] = to_agpr(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
      }
    }
  };

  // ---- phase B: NR output rows y0 .. y0 + NR - 1 from win[0 .. NR + 1] ----
  auto phase_b = [&](auto nr_tag, int y0) {
    constexpr int NR = decltype(nr_tag)::value;
    f32x4 bacc[NR][3][2];
    u32x4 w3f[6];
    auto load_w3 = [&](int dy, int t, int f) { w3f[f] = *(const u32x4 *)(w3l + ((dy * 4 + t) * 6 + f) * 1024); };
#pragma unroll
    for (int f = 0; f < 6; ++f) load_w3(0, 0, f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int nxt = dy * 4 + t + 1;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
          constexpr int R2 = NR > 2 ? 2 : 0, R3 = NR > 2 ? 3 : 0;      // (two-row groups: operands 2, 3 unused)
          if (dy == 0 && t == 0)
            mfma16_rows<true, NR>(&bacc[0][f >> 1][f & 1], &bacc[1][f >> 1][f & 1], &bacc[R2][f >> 1][f & 1], &bacc[R3][f >> 1][f & 1], w3f[f],
                                  win[dy][t], win[1 + dy][t], win[R2 + dy][t], win[R3 + dy][t]);
          else
            mfma16_rows<false, NR>(&bacc[0][f >> 1][f & 1], &bacc[1][f >> 1][f & 1], &bacc[R2][f >> 1][f & 1], &bacc[R3][f >> 1][f & 1], w3f[f],
                                   win[dy][t], win[1 + dy][t], win[R2 + dy][t], win[R3 + dy][t]);
          if (nxt < 12) load_w3(nxt >> 2, nxt & 3, f);
        }
      }
    mfma_results_ready(bacc);
    // ---- out[x] = acc[dx = 0][x] + acc[dx = -1][x - 1] + acc[dx = +1][x + 1]; fp16; 16 B per lane (channels 8 g4 .. 8 g4 + 7) ----
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32x4 o;
#pragma unroll
      for (int of = 0; of < 2; ++of) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          v[c] = bacc[r][1][of][c] + dpp_f32<0x111>(bacc[r][0][of][c]) + dpp_f32<0x101>(bacc[r][2][of][c]);   // row_shr:1, row_shl:1
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 p0 = {(f16)v[0], (f16)v[1]}, p1 = {(f16)v[2], (f16)v[3]};
        o[of * 2] = __builtin_bit_cast(unsigned, p0);
        o[of * 2 + 1] = __builtin_bit_cast(unsigned, p1);
      }
      // (the two halo lanes carry an offset past the descriptor's range: the hardware drops their store, no branch)
      __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, outb + (unsigned)(y0 + r) * rowpitch, 0, 0);
    }
  };
  auto shift_window = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      win[0][t] = win[4][t];
      win[1][t] = win[5][t];
    }
  };

  // ---- the wave's program: rows r_lo - 1, r_lo first, then groups of four output rows (+ a half group) ----
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  issue(I0{}, I0{}, r_lo - 1);
  if constexpr (NSU > 1) issue(I1{}, I1{}, r_lo - 1);
  if constexpr (NSU > 2) issue(I2{}, I2{}, r_lo - 1);
  phase_a(std::integral_constant<int, 2>{}, r_lo - 1, r_lo + 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    win[4][t] = win[2][t];
    win[5][t] = win[3][t];
  }
  int y = r_lo;
  for (int g = 0; g < G::NG; ++g, y += 4) {
    shift_window();
    phase_a(std::integral_constant<int, 4>{}, y + 1, y + 5);
    phase_b(std::integral_constant<int, 4>{}, y);
  }
  if constexpr (G::TAIL) {
    shift_window();
    phase_a(std::integral_constant<int, 2>{}, y + 1, y + 1);
    phase_b(std::integral_constant<int, 2>{}, y);
  }
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

// The 7x7 dense block (DenseNet-121 block 4: 16 layers, 512 -> 1024 channels) as ONE launch with the frame's concat buffer
// resident in LDS (round 3).
//
//   for l in 0 .. nl-1:   y[.., K_l : K_l+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( y[.., 0:K_l] )) ) )) ),   K_l = K0 + 32 l
//
// (reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_dense_block / _make_dense_layer.)
// dense_layer_big.hip walks this block with one 8-wave workgroup per frame too, but streams activations AND weights through
// a barrier-fenced LDS ring: 1 150 cycles per 64-channel stage for 128 cycles of MFMA work - latency, not
// bandwidth.  A 7x7 frame is 49 pixels: its whole 1024-channel concat buffer is 100 KB, and the only thing that has to
// stream is the weights (4.3 MB per frame from L2, the same for every frame).  So:
//
// * a workgroup = one frame, 4 waves = one wave per SIMD with the whole register file.  The frame's activations sit in LDS
//   for the whole launch as [8-channel chunk][pixel] 16-byte cells (fragment reads and the chunk-wise appends are both
//   conflict-free); each layer appends its 32 new channels there and stores them to the block buffer in HBM for the head.
// * 1x1 GEMM (128 x K x 64 pixel slots) on v_mfma_f32_32x32x16_f16 with the WEIGHTS as the A operand, loaded global -> VGPR
//   directly in fragment shape (1 KiB per wave-instruction) through a register ring of D k-steps; the packed stream is laid
//   out per wave in consumption order, so a load is "next KiB".  The four waves split K (each k-step's pixel fragment is
//   read from LDS and BatchNorm'd exactly once in the workgroup, and every weight byte is loaded exactly once); the four
//   128 x 64 partial sums are reduced through a 32 KB LDS buffer in three rounds, after which wave w owns bottleneck
//   channels 32w .. 32w+31.  Accumulator set m of wave w is M-tile (w + m) & 3 - the rotation is in the packed stream - so
//   that "my tile" is a compile-time register index.
// * the ring is indexed statically: the k loop runs in groups of D steps, and its last D + r steps (r = steps mod D) are
//   one of D straight-line variants which also refill the ring with the NEXT layer's first D steps - the weight stream never
//   stops at a layer boundary.
// * BN2's scale is folded into the 1x1 weights (weights.as_fp16_model); shift + ReLU + fp16 happen on the reduced tile,
//   which goes to a pixel-slot tile in LDS (272-byte slots: consecutive pixels rotate through the banks; slot 49 is the zero
//   padding).  The 3x3 convolution splits its K = 9 x 128 over the waves by bottleneck channel: wave w convolves exactly
//   the 32 channels it produced - no barrier between the two GEMMs - and the four 32 x 64 partial outputs are reduced through
//   the same LDS buffer.
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

#ifndef TN_B7_EXP
#define TN_B7_EXP 0   // timing experiments only (results wrong): bit 0 no ring refill, bit 1 no BatchNorm of the pixel fragments, bit 2 no LDS fragment reads
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define TN_INL __attribute__((always_inline))
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  [&]<int... I>(std::integer_sequence<int, I...>) TN_INL { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int V>
using ic = std::integral_constant<int, V>;
#define TN_SB() __builtin_amdgcn_sched_barrier(0)

constexpr int kPix = 49;
constexpr int kChunkRow = kPix * 16;                 // one 8-channel chunk of all pixels: 784 B
constexpr int kActBytes = 128 * kChunkRow;           // 100 352
constexpr int kSlotPitch = 272;                      // bottleneck tile: 128 channels of a pixel + 16 B of rotation
constexpr int kZeroSlot = 49, kDumpSlot = 50;
constexpr int kTileOff = kActBytes, kTileBytes = 51 * kSlotPitch;
constexpr int kRedOff = kTileOff + kTileBytes, kRedBytes = 4 * 8192;
constexpr int kTabOff = kRedOff + kRedBytes;
constexpr int kTabFloats = 512 + 512 + 128;          // lo1[1024] | hi1[1024] as fp16 (BN1 + ReLU as a clamp: calib_host.hip) | t2[128] fp32 of one layer
constexpr int kLdsBytes = kTabOff + kTabFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS");
constexpr int kStepUnits = 256;                      // a k-step of the 1x1 stream: 4 fragments x 64 lanes, in 16-byte units
constexpr int kW3Units = 18 * 64;                    // a wave's 3x3 fragments of one layer

__device__ __forceinline__ f32x16 mfma32(const f16x8 a, const f16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) dense_block7_kernel(DenseBlock7Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  f16 *xb = a.buf + (size_t)blockIdx.x * kPix * a.ldc;

  auto lds_h8 = [&](unsigned ad) TN_INL -> f16x8 { return *(const f16x8 *)(smem + ad); };
  auto lds_f4 = [&](unsigned ad) TN_INL -> f32x4 { return *(const f32x4 *)(smem + ad); };

  // ---- the weight streams of this wave (16-byte units, this lane's cell of each fragment) ----
  // (wave-uniform bases + the lane's cell: the loads take the base from SGPRs and need no per-lane pointer arithmetic)
  const f16x8 *wa = (const f16x8 *)a.wa + a.a_off[w];
  const f16x8 *wb = (const f16x8 *)a.wb + a.b_off[w];
  f16x8 ring[D][4];
  static_for<D>([&](auto i_tag) TN_INL {
    static_for<4>([&](auto m_tag) TN_INL { ring[i_tag.value][m_tag.value] = wa[i_tag.value * kStepUnits + m_tag.value * 64 + lane]; });
  });
  wa += D * kStepUnits;

  // ---- prologue: the block's input channels -> LDS, the zero slot, layer 0's tables ----
  {
    const int c8 = a.K0 >> 3, total = c8 * kPix;
    for (int idx = tid; idx < total; idx += 256) {
      const int p = idx / c8, c = idx - p * c8;
      *(f16x8 *)(smem + c * kChunkRow + p * 16) = *(const f16x8 *)(xb + (size_t)p * a.ldc + c * 8);
    }
    if (tid < kSlotPitch / 16) *(u32x4 *)(smem + kTileOff + kZeroSlot * kSlotPitch + tid * 16) = (u32x4){0, 0, 0, 0};
    const f32x4 *tb = (const f32x4 *)a.tab;
    f32x4 *tl = (f32x4 *)(smem + kTabOff);
    tl[tid] = tb[tid];
    if (tid < 32) tl[256 + tid] = tb[256 + tid];
  }
  __syncthreads();

  // ---- lane geometry (the same for every layer) ----
  const unsigned aaddr0 = (unsigned)(h * kPix + n) * 16;                 // pixel fragment of tile 0, chunk h; tile 1: + 512
  const unsigned caddr0 = kTabOff + h * 16;                              // lo1 of channels 8h .. 8h+7 (packed halves); hi1: + 2048
  unsigned waddr[2], offb[9][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = 32 * t + n, y = p / 7, x = p - 7 * y;
    const unsigned cell = (unsigned)(4 * w + h) * 16;                    // chunk 2q + h of k-step q = 2w + s; s: + 32
    waddr[t] = kTileOff + (p < kPix ? p : kDumpSlot) * kSlotPitch + cell;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
      const bool ok = p < kPix && yy >= 0 && yy < 7 && xx >= 0 && xx < 7;
      offb[tap][t] = kTileOff + (ok ? yy * 7 + xx : kZeroSlot) * kSlotPitch + cell;
    }
  }

  f32x16 acc[4][2];

  // pixel fragments: raw values + BatchNorm constants of the NEXT k-step, and the finished fragments of this / the next one
  // (the LDS reads run TWO steps ahead of the MFMAs, BatchNorm one step ahead: a read issued in step P is consumed by the
  // element-wise work in the shadow of step P+1's MFMAs)
  u32x4 raw[2][2], x[2][2];
  u32x4 cs[2][2];
  auto read_step = [&](unsigned aad, unsigned cad, auto i_tag) TN_INL {     // step I (relative to aad / cad) -> buffers I & 1
    constexpr int I = decltype(i_tag)::value, Bf = I & 1;
    raw[Bf][0] = __builtin_bit_cast(u32x4, lds_h8(aad + I * 2 * kChunkRow));
    raw[Bf][1] = __builtin_bit_cast(u32x4, lds_h8(aad + I * 2 * kChunkRow + 512));
    cs[Bf][0] = __builtin_bit_cast(u32x4, lds_h8(cad + I * 32));
    cs[Bf][1] = __builtin_bit_cast(u32x4, lds_h8(cad + I * 32 + 2048));
  };
  // BatchNorm + ReLU of two dwords (four channels) of one tile as a clamp against the channels' thresholds (round 5: no
  // arithmetic, no rounding - the scale is in the 1x1 weights, the shift in BN2's; calib_host.hip::bn_relu_clamp_fold): two
  // independent chains in ONE statement (between two statements hipcc pads the dependency with an s_nop the hardware does not
  // need).  Four packed-half instructions where the fp32 form had eight.
  auto bn_pair = [&](auto u_tag, auto par_tag) TN_INL {
    constexpr int U = decltype(u_tag)::value, T = U >> 1, D0 = 2 * (U & 1), D1 = D0 + 1, PAR = decltype(par_tag)::value;
    const unsigned in0 = raw[PAR][T][D0], in1 = raw[PAR][T][D1];
    const unsigned l0 = cs[PAR][0][D0], l1 = cs[PAR][0][D1], h0 = cs[PAR][1][D0], h1 = cs[PAR][1][D1];
    unsigned o0, o1;
    asm("v_pk_max_f16 %0, %2, %4\n\tv_pk_max_f16 %1, %3, %5\n\tv_pk_min_f16 %0, %0, %6\n\tv_pk_min_f16 %1, %1, %7"
        : "=&v"(o0), "=&v"(o1) : "v"(in0), "v"(in1), "v"(l0), "v"(l1), "v"(h0), "v"(h1));
    x[PAR][T][D0] = o0;
    x[PAR][T][D1] = o1;
  };
  // One k-step: 8 MFMAs (4 M-tiles x 2 pixel tiles) on ring slot S with the fragments of parity P & 1; in their shadow the
  // LDS reads and BatchNorm of the next step and the refill of the slot: KIND 0 none, 1 the next step of the stream,
  // 2 step S of the next layer (the stream pointer then stands at that layer's first step).
  // P = position relative to aad / cad (parity and LDS immediates), slot P % D; LAST = the last position of a straight-line
  // stretch that ends the layer (0: more steps follow); position 0 is the layer's first step and starts the accumulators.
  auto step = [&](auto p_tag, auto kind_tag, auto last_tag, unsigned aad, unsigned cad) TN_INL {
    constexpr int P = decltype(p_tag)::value, S = P % D, KIND = decltype(kind_tag)::value, PAR = P & 1;
    constexpr int LAST = decltype(last_tag)::value;
    constexpr bool NEXT = LAST == 0 || P + 1 <= LAST, NEXT2 = LAST == 0 || P + 2 <= LAST;
    static_for<8>([&](auto j_tag) TN_INL {
      constexpr int J = decltype(j_tag)::value, M = J >> 1, T = J & 1;
      if constexpr (J == 0 && NEXT2 && !(TN_B7_EXP & 4)) read_step(aad, cad, ic<P + 2>{});
      if constexpr (P == 0) {
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        acc[M][T] = mfma32(ring[S][M], __builtin_bit_cast(f16x8, x[PAR][T]), z);
      } else {
        acc[M][T] = mfma32(ring[S][M], __builtin_bit_cast(f16x8, x[PAR][T]), acc[M][T]);
      }
      if constexpr (NEXT && !(TN_B7_EXP & 2) && (J & 1) == 0) bn_pair(ic<(J >> 1)>{}, ic<PAR ^ 1>{});
      if constexpr (T == 1 && KIND == 1 && !(TN_B7_EXP & 1)) ring[S][M] = wa[M * 64 + lane];
      if constexpr (T == 1 && KIND == 2 && !(TN_B7_EXP & 1)) ring[S][M] = wa[S * kStepUnits + M * 64 + lane];
      TN_SB();
    });
    if constexpr (KIND == 1) wa += kStepUnits;
  };

  int nstamp = 0;
  auto stamp = [&]() TN_INL {
    if (a.ts) {
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
      if (w == 0 && lane == 0 && nstamp < 126) a.ts[(size_t)blockIdx.x * 128 + nstamp] = tnow;
      ++nstamp;
    }
  };
  if (a.ts && w == 0 && lane == 0) a.ts[(size_t)blockIdx.x * 128 + 127] = __builtin_amdgcn_s_memrealtime();
  stamp();
  int K = a.K0;
  for (int l = 0; l < a.nl; ++l, K += 32) {
    const int G = K >> 4, gbase = G >> 2, grem = G & 3;
    const int nA = gbase + (w < grem ? 1 : 0);
    const int g0 = w * gbase + (w < grem ? w : grem);

    // The ring's first D steps were requested a whole reduce + 3x3 ago, in an order that depends on the previous layer's
    // remainder variant.  An explicit vmcnt(0) here costs nothing and gives hipcc's wait-count pass ONE state to start the
    // layer from: merged over the six variants it would otherwise assume the worst of each and drain the queue (every load
    // of the k loop included) in front of the first ring read - 2 700 cycles per layer, measured.
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt / lgkmcnt untouched

    // ======================= 1x1: partial bottleneck over this wave's k-steps =======================
    unsigned aad = aaddr0 + (unsigned)g0 * (2 * kChunkRow), cad = caddr0 + (unsigned)g0 * 32;
    read_step(aad, cad, ic<0>{});
    read_step(aad, cad, ic<1>{});
    static_for<4>([&](auto u_tag) TN_INL { bn_pair(u_tag, ic<0>{}); });
    TN_SB();
    step(ic<0>{}, ic<1>{}, ic<0>{}, aad, cad);          // step 0 starts the accumulators (C = 0); its slot takes step D
#ifdef TN_B7_STAMPS_A
    stamp();
#endif
    int b = 0;                                          // steps b+1 .. are still to come; the ring holds b+1 .. b+D
    while (b + 2 * D + 1 <= nA) {
      static_for<D>([&](auto i_tag) TN_INL { step(ic<decltype(i_tag)::value + 1>{}, ic<1>{}, ic<0>{}, aad, cad); });
      aad += D * 2 * kChunkRow;
      cad += D * 32;
      b += D;
    }
#ifdef TN_B7_STAMPS_A
    stamp();
#endif
    const int r = nA - 1 - b - D;        // 0 .. D-1 steps of this layer still to be requested
    static_for<D>([&](auto r_tag) TN_INL {
      constexpr int R = decltype(r_tag)::value;
      if (r == R) {
        static_for<D + R>([&](auto i_tag) TN_INL {
          constexpr int P = decltype(i_tag)::value + 1;
          step(ic<P>{}, ic<(P <= R ? 1 : 2)>{}, ic<D + R>{}, aad, cad);
        });
      }
    });
    wa += D * kStepUnits;
    // this layer's 3x3 fragments: requested here, they land during the reduction (requested before the 1x1 they would sit in
    // 72 registers through the whole k loop, and hipcc then moves some of them between register files right behind their
    // loads: a full L2 latency per layer, measured)
    f16x8 w3f[18];
    static_for<18>([&](auto u_tag) TN_INL { w3f[u_tag.value] = wb[u_tag.value * 64 + lane]; });
    wb += kW3Units;
    stamp();

    // ======================= reduce the four partial tiles: wave w keeps M-tile w = accumulator set 0 =======================
    const float *tbn = a.tab + (size_t)(l + 1) * kTabFloats;
    f32x4 tv0, tv2;
    static_for<3>([&](auto rd_tag) TN_INL {
      constexpr int RD = decltype(rd_tag)::value + 1;
      {
        const unsigned base = kRedOff + (unsigned)((w + RD) & 3) * 8192 + lane * 16;
        static_for<8>([&](auto i_tag) TN_INL {
          constexpr int T = decltype(i_tag)::value >> 2, Q = decltype(i_tag)::value & 3;
          const f32x16 v = acc[RD][T];
          *(f32x4 *)(smem + base + T * 4096 + Q * 1024) = (f32x4){v[4 * Q], v[4 * Q + 1], v[4 * Q + 2], v[4 * Q + 3]};
        });
      }
      __syncthreads();
      if constexpr (RD == 1) {
        if (l + 1 < a.nl) {        // the next layer's tables: requested now, stored behind the 3x3
          tv0 = ((const f32x4 *)tbn)[tid];
          if (tid < 32) tv2 = ((const f32x4 *)tbn)[256 + tid];
        }
      }
      {
        const unsigned base = kRedOff + (unsigned)w * 8192 + lane * 16;
        static_for<8>([&](auto i_tag) TN_INL {
          constexpr int T = decltype(i_tag)::value >> 2, Q = decltype(i_tag)::value & 3;
          const f32x4 v = lds_f4(base + T * 4096 + Q * 1024);
          acc[0][T][4 * Q] += v[0];
          acc[0][T][4 * Q + 1] += v[1];
          acc[0][T][4 * Q + 2] += v[2];
          acc[0][T][4 * Q + 3] += v[3];
        });
      }
      if constexpr (RD < 3) __syncthreads();
    });

    stamp();
    // ======================= BN2 shift + ReLU + fp16 -> this wave's 32 channels of the pixel-slot tile =======================
    {
      const unsigned t2ad = kTabOff + 4096 + (unsigned)(32 * w + 4 * h) * 4;
      f32x4 sh[4];
      static_for<4>([&](auto g_tag) TN_INL { sh[g_tag.value] = lds_f4(t2ad + g_tag.value * 32); });
      static_for<2>([&](auto t_tag) TN_INL {
        constexpr int T = decltype(t_tag)::value;
        f16x8 lo, hi;
        static_for<8>([&](auto j_tag) TN_INL {
          constexpr int J = decltype(j_tag)::value;
          lo[J] = (f16)fmaxf(acc[0][T][J] + sh[J >> 2][J & 3], 0.f);
          hi[J] = (f16)fmaxf(acc[0][T][8 + J] + sh[2 + (J >> 2)][J & 3], 0.f);
        });
        *(f16x8 *)(smem + waddr[T]) = lo;
        *(f16x8 *)(smem + waddr[T] + 32) = hi;
      });
    }

    stamp();
    // ======================= 3x3 over this wave's 32 bottleneck channels (all nine taps) =======================
    f32x16 q[2];
    f16x8 bq[2][2];
    bq[0][0] = lds_h8(offb[0][0]);
    bq[0][1] = lds_h8(offb[0][1]);
    static_for<36>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value, U = I >> 1, T = I & 1, TAP = U >> 1, SS = U & 1;
      if constexpr (U + 1 < 18) bq[(U + 1) & 1][T] = lds_h8(offb[(U + 1) >> 1][T] + ((U + 1) & 1) * 32);
      if constexpr (U == 0) {
        f32x16 z;
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = 0.f;
        q[T] = mfma32(w3f[U], bq[U & 1][T], z);
      } else {
        q[T] = mfma32(w3f[U], bq[U & 1][T], q[T]);
      }
      (void)TAP; (void)SS;
      TN_SB();
    });

    stamp();
    // ======================= reduce the four partial outputs; append the layer's 32 channels =======================
    {
      const unsigned base = kRedOff + (unsigned)w * 8192 + lane * 16;
      static_for<8>([&](auto i_tag) TN_INL {
        constexpr int T = decltype(i_tag)::value >> 2, Q = decltype(i_tag)::value & 3;
        *(f32x4 *)(smem + base + T * 4096 + Q * 1024) = (f32x4){q[T][4 * Q], q[T][4 * Q + 1], q[T][4 * Q + 2], q[T][4 * Q + 3]};
      });
    }
    __syncthreads();
    {
      const int tf = w & 1, sf = w >> 1;
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
      static_for<4>([&](auto x_tag) TN_INL {
        const unsigned base = kRedOff + decltype(x_tag)::value * 8192 + (unsigned)tf * 4096 + (unsigned)(2 * sf) * 1024 + lane * 16;
        o0 += lds_f4(base);
        o1 += lds_f4(base + 1024);
      });
      f16x8 ov;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ov[j] = (f16)o0[j];
        ov[4 + j] = (f16)o1[j];
      }
      const int p = 32 * tf + n;
      if (p < kPix) {
        *(f16x8 *)(smem + (unsigned)((K >> 3) + 2 * sf + h) * kChunkRow + p * 16) = ov;
        *(f16x8 *)(xb + (size_t)p * a.ldc + K + 16 * sf + 8 * h) = ov;
        // the head (BN + ReLU + 7x7 average -> the features) reads the block's output UN-rounded from a side buffer: a feature
        // of magnitude 8 otherwise carries the fp16 rounding of its 49 stored values directly (round 5; the later layers of the
        // block go on consuming the fp16 copy: that is their MFMA operand)
        if (a.side) {
          float *sp = a.side + ((size_t)blockIdx.x * kPix + p) * a.ldc + K + 16 * sf + 8 * h;
          *(f32x4 *)sp = o0;
          *(f32x4 *)(sp + 4) = o1;
        }
      }
      if (l + 1 < a.nl) {
        f32x4 *tl = (f32x4 *)(smem + kTabOff);
        tl[tid] = tv0;
        if (tid < 32) tl[256 + tid] = tv2;
      }
    }
    __syncthreads();
    stamp();
  }
  if (a.ts && w == 0 && lane == 0) a.ts[(size_t)blockIdx.x * 128 + 126] = __builtin_amdgcn_s_memrealtime();
}

constexpr int kRingDepth = 6;

}  // namespace

bool dense_block7_supported(int H, int W, int K0, int nl) {
  return H == 7 && W == 7 && K0 % 32 == 0 && K0 >= 64 * (kRingDepth + 1) && nl >= 1 && K0 + 32 * nl <= 1024;
}

// Host side: the per-wave weight streams and the per-layer tables.
//  wa: for wave w, layer l, k-step g of that wave's range, accumulator set m: one A fragment [64 lanes][8] of M-tile
//      (w + m) & 3: lane (r = lane & 31, h = lane >> 5) holds w1[32 M + r][16 g + 8 h + i], i = 0..7.  The stream ends with
//      kRingDepth zero steps (the last layer's "next layer" refill).
//  wb: for wave w, layer l, tap, s: one A fragment of the 3x3: row r = output channel 16 (r >> 4) + 8 ((r >> 2) & 1) + (r & 3) +
//      4 ((r >> 3) & 1); k = 8 h + i = bottleneck channel 32 w + (i & 3) + 8 (2 s + (i >> 2)) + 4 h.
//  tab: per layer lo1[1024] | hi1[1024] as fp16 (zero-padded: a padding channel clamps to 0) | t2[128] fp32.
Block7Image pack_block7(const std::vector<Block7Layer> &layers, int K0) {
  Block7Image img;
  const int nl = (int)layers.size();
  std::vector<f16> wave_a[4], wave_b[4];
  for (int l = 0; l < nl; ++l) {
    const Block7Layer &L = layers[l];
    const int K = K0 + 32 * l, G = K / 16, gbase = G / 4, grem = G % 4;
    for (int w = 0; w < 4; ++w) {
      const int nA = gbase + (w < grem ? 1 : 0), g0 = w * gbase + (w < grem ? w : grem);
      for (int sidx = 0; sidx < nA; ++sidx)
        for (int m = 0; m < 4; ++m) {
          const int M = (w + m) & 3, g = g0 + sidx;
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i)
              wave_a[w].push_back((f16)L.w1f[(size_t)(32 * M + (lane & 31)) * K + 16 * g + 8 * (lane >> 5) + i]);
        }
      for (int tap = 0; tap < 9; ++tap)
        for (int s = 0; s < 2; ++s)
          for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 31, h = lane >> 5;
            const int co = 16 * (r >> 4) + 8 * ((r >> 2) & 1) + (r & 3) + 4 * ((r >> 3) & 1);
            for (int i = 0; i < 8; ++i) {
              const int cb = 32 * w + (i & 3) + 8 * (2 * s + (i >> 2)) + 4 * h;
              wave_b[w].push_back((f16)L.w3[((size_t)co * 128 + cb) * 9 + tap]);
            }
          }
    }
    img.tab.resize((size_t)(l + 1) * kTabFloats, 0.f);
    float *tb = img.tab.data() + (size_t)l * kTabFloats;
    f16 *th = (f16 *)tb;
    for (int k = 0; k < K; ++k) {
      th[k] = (f16)L.s1[k];           // (fp16 numbers: bn_relu_clamp_fold)
      th[1024 + k] = (f16)L.t1[k];
    }
    memcpy(tb + 1024, L.t2, sizeof(float) * 128);
  }
  for (int w = 0; w < 4; ++w) {
    wave_a[w].resize(wave_a[w].size() + (size_t)kRingDepth * kStepUnits * 8, (f16)0.f);
    img.a_off[w] = (unsigned)(img.wa.size() / 8);
    img.wa.insert(img.wa.end(), wave_a[w].begin(), wave_a[w].end());
    img.b_off[w] = (unsigned)(img.wb.size() / 8);
    img.wb.insert(img.wb.end(), wave_b[w].begin(), wave_b[w].end());
  }
  return img;
}

int launch_dense_block7(const DenseBlock7Args &a, hipStream_t s) {
  TN_REQUIRE(dense_block7_supported(7, 7, a.K0, a.nl) && a.ldc % 8 == 0 && a.K0 + 32 * a.nl <= a.ldc && a.B > 0,
             "dense_block7: bad geometry");
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_block7_kernel<kRingDepth>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes)));
  hipLaunchKernelGGL((dense_block7_kernel<kRingDepth>), dim3(a.B), dim3(256), kLdsBytes, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// The 28x28 dense block (12 layers at 224x224 input, K = 128 .. 480) as ONE launch of pixel-owning waves (round 4).
//
//   for l in 0 .. nl-1:   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) ),  K = K0 + 32 l
//
// (reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_dense_block / _make_dense_layer.)
//
// Until round 4 this block ran as seven strip-kernel launches (K <= 320: weights resident in LDS, dense_strip_impl.h) and five
// launches of round 2's barrier-phased tile kernel (K > 320: 23 % of the MFMA peak).  A 28x28 frame gives a
// SIMD 196 pixels, the 1x1 weights of its later layers alone (K x 128 fp16 = 120 KB at K = 480) do not fit LDS next to the 3x3's,
// and 256 frames are one workgroup per CU: a per-layer launch cannot hide its weight prologue behind another workgroup.  This
// kernel is dense_block14.hip's recipe with the frame walked in four PASSES of eight image rows:
//
// * a workgroup = one frame, 4 waves = ONE wave per SIMD with the whole register file.  In pass p wave w owns image rows
//   8 p + 2 w (fragment X) and 8 p + 2 w + 1 (fragment Y): 30 of the 32 slots of a v_mfma_f32_32x32x16_f16's N are a row with
//   its two padding columns; every 1 KiB weight fragment read from LDS feeds TWO MFMAs.  Rows 28 .. 31 of the last pass do not
//   exist (waves 2 / 3 idle through it: 28 of 32 rows x 30 of 32 slots = 82 % of the issued MFMA columns are pixels).
// * ALL weights stream, once per pass: the layer's 1x1 fragments with their BN1 constants and its 3x3 fragments are a linear
//   sequence of 16.5 KiB units in consumption order (pack_block28: the four passes of a layer repeat the layer's units), copied
//   by LDS-DMA into a ring of five slots four units ahead, one s_barrier per unit (dense_block14.hip).
// * activations go HBM / L2 -> registers in fragment shape through the register ring of two 64-channel super-steps, which runs
//   THROUGH pass and layer boundaries (a private k-step-major copy of the frame, as in dense_block14.hip: 896 contiguous bytes
//   per fragment row).  Nothing is forwarded in registers here: a wave's 3x3 output rows are one row above its 1x1 rows.
// * the bottleneck goes accumulator -> ReLU -> fp16 -> a ROLLING LDS tile of ten rows (the eight of the pass + the last two of
//   the pass before: 75 KB, 200 KB for the whole frame would not fit): after the 1x1 of rows 8 p .. 8 p + 7 the 3x3 produces
//   output rows 8 p - 1 .. 8 p + 6 (wave w: rows 8 p + 2 w - 1 and 8 p + 2 w), the pass after brings the row below.  Kernel rows
//   are applied in the order +1, 0, -1: the first third of the 3x3 only reads the wave's own bottleneck rows, the neighbours'
//   are published by the barrier of the third 3x3 unit.  The tile slots of rows 28 / 29 (written as zeros: masked by the shift
//   k-step) are the slots of rows -2 / -1 of the next layer's first pass.
// * BN2's shift k-step runs FIRST: its eight MFMAs open the pass' first 1x1 unit (the fragments ride in the four spare fragment
//   slots of the previous pass' last 3x3 unit and wait in registers), so a pass' accumulators start from the shift and its last
//   1x1 unit runs straight into the 3x3.
// * vmcnt is hand-counted for every steady-state load as in dense_block14.hip (tests/test_cpu_block28.py replays the issue order).
#include <array>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

constexpr int kUnitFrag = 16384;                  // 16 A fragments of 1 KiB
constexpr int kUnitBytes = kUnitFrag + 512;       // + BN1 constants of the unit's k-steps (dense_block14.hip's layout)
constexpr int kNR = 5;                            // ring slots
constexpr int kRowSlots = 30;                     // a tile row: columns -1 .. 28
constexpr int kTileRowB = kRowSlots * 256;        // 128 bottleneck channels, fp16, per slot
constexpr int kTileRows = 10;                     // rolling: row R lives in tile row (R + 2) % 10
constexpr int kTileBytes = kTileRows * kTileRowB;
constexpr int kDumpBytes = 512;                   // where the two lanes without a slot (n = 30, 31) read and write
constexpr int kRingOff = kTileBytes + kDumpBytes;
constexpr int kLdsBytes = kRingOff + kNR * kUnitBytes;
constexpr int kPix = 784;
constexpr int kPlaneB = kPix * 32;                // one k-step (16 channels) of a frame in the private k-step-major copy
constexpr int kPlanes = 36;                       // 576 channels: K <= 512 + the zero-weighted pad of a last super-step
constexpr int kFrameScrB = kPlanes * kPlaneB + 4096;   // (+ rows 28 .. 31 of the last plane: read, never used)
constexpr int kPassB = 8 * 28 * 32;               // a pass' rows inside a plane
static_assert(kLdsBytes <= 160 * 1024, "LDS");

// s_waitcnt vmcnt(N) constants (asm loads only; tests/test_cpu_block28.py derives every one of them from the issue order)
constexpr int kVmRing = 24;        // a ring register pair is waited for two super-step intervals (2 x 13 loads) after its refills
constexpr int kVmDmaSU0 = 12, kVmDmaSU = 20, kVmDmaB0 = 24, kVmDmaB = 10;

#define TN_INL __attribute__((always_inline))
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  [&]<int... I>(std::integer_sequence<int, I...>) TN_INL { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int V>
using ic = std::integral_constant<int, V>;
#define TN_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x16 mfma32(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// LDS-DMA: global (wave-uniform base in SGPRs + per-lane 32-bit offset) -> LDS (M0 + lane * size); the instruction offset applies
// to the global AND the LDS address (dense_block14.hip)
template <int OFF>
__device__ __forceinline__ void dma16x2(const void *gbase, unsigned voff16, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%c4\n\t"
               "global_load_lds_dwordx4 %2, %3 offset:%c5\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff16), "s"(gbase), "n"(OFF), "n"(OFF + 1024));
}
__device__ __forceinline__ void dma4(const void *gbase, unsigned voff4, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff4), "s"(gbase));
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N) : "memory"); }

// The activation ring lives in LITERAL registers v[192:255] (dense_block14.hip: a value hipcc knows about may be copied or spilled
// while its load is in flight); scripts/audit_block14_isa.py checks the ISA for strays.
#define TN_RING_BASE 192
#define TN_RING_CLOBBER                                                                                                             \
  "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",   \
  "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",   \
  "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",   \
  "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define TN_RING_FENCE() asm volatile("" ::: TN_RING_CLOBBER)
constexpr int ring_reg(int rs, int kq, int f) { return TN_RING_BASE + ((rs * 4 + kq) * 2 + f) * 4; }   // [super-step parity][k-step][fragment] x 4 dwords

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_block28_kernel(DenseBlock28Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int ldc = a.ldc;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);
  unsigned char *fb = (unsigned char *)(a.buf + (size_t)blockIdx.x * kPix * ldc);
  if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memtime();

  // ---- per-lane geometry: slot n holds column n - 1; fragment F of pass p holds row 8 p + 2 wid + F ----
  const int col = n < 1 ? 0 : (n > 28 ? 27 : n - 1);
  const bool colvalid = n >= 1 && n <= 28;
  const bool haslot = n < kRowSlots;
  unsigned voff[2];          // scr: byte offset of the lane's 16 B inside a k-step plane, pass 0 (a pass adds kPassB)
  unsigned spix[2];          // pixel index of the lane's 3x3 OUTPUT of fragment F in pass 0: row 2 wid - 1 + F
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    voff[f] = (unsigned)((2 * wid + f) * 28 + col) * 32 + 16 * h;
    spix[f] = (unsigned)((2 * wid - 1 + f) * 28 + col);
  }
  // (wave-uniform by construction; says so to hipcc, whose "s" operand otherwise comes out as a VGPR pair when a select feeds it)
  auto uniform_ptr = [&](const unsigned char *q) TN_INL -> const unsigned char * {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const unsigned char *)(((unsigned long long)hi << 32) | lo);
  };
  unsigned char *scr = (unsigned char *)uniform_ptr((const unsigned char *)a.scratch + (size_t)blockIdx.x * kFrameScrB);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(scr, 0, (int)(kPlanes * kPlaneB), 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(fb, 0, (int)((unsigned)kPix * ldc * 2), 0x00020000);
  // tile: cell of (tile row, slot, tuple T, half h) at row * kTileRowB + slot * 256 + ((2 T + h) ^ (slot & 15)) * 16 = base ^ 32 T;
  // lanes 30 / 31 have no slot: their reads and writes go to the dump area behind the tile
  const unsigned lt0 = haslot ? (unsigned)n * 256 + (((unsigned)((n & 15) ^ h)) << 4) : (unsigned)kTileBytes + (unsigned)(n - kRowSlots) * 256 + ((unsigned)h << 4);
  const unsigned lane16 = lane * 16, lane4 = lane * 4;

  // ================= the weight stream (dense_block14.hip) =================
  const unsigned ring0 = lds0 + kRingOff, ring_end = ring0 + kNR * kUnitBytes;
  const unsigned char *dsrc = a.stream + wid * 4096;            // this wave's part of the unit to copy next (unit g + 4)
  unsigned ddst = ring0 + wid * 4096;                           // ... and where it goes
  const int cdelta = kUnitFrag - wid * 4096 + (wid & 1) * 256;  // from there to this wave's piece of the constants
  unsigned nxt = kRingOff;                                      // byte offset in smem of the slot of unit g + 1
  unsigned vb_cur = 0, vb_next = 0, vc_cur = 0, vc_next = 0;    // LDS byte offsets: fragments (+ lane * 16) / constants (+ h * 64) of unit g, g + 1
  auto dma_pair = [&](auto pr_tag) TN_INL { dma16x2<decltype(pr_tag)::value * 2048>(dsrc, lane16, ddst); };
  auto dma_consts = [&]() TN_INL { dma4(dsrc + cdelta, lane4, ddst + cdelta); };
  auto advance_dma = [&]() TN_INL {
    dsrc += kUnitBytes;
    ddst = ddst + kUnitBytes >= ring_end ? ddst + kUnitBytes - kNR * kUnitBytes : ddst + kUnitBytes;
  };
  // start of interval g: unit g + 1 becomes visible, the slot of unit g - 1 becomes free
  auto begin_interval = [&](auto vm_tag) TN_INL {
    wait_vm<decltype(vm_tag)::value>();          // this wave's pieces of unit g + 1 (issued in interval g - 3) have landed
    asm volatile("s_barrier" ::: "memory");
    vb_cur = vb_next; vc_cur = vc_next;
    vb_next = nxt + lane16;
    vc_next = nxt + kUnitFrag + 64 * h;
    nxt = nxt + kUnitBytes >= (unsigned)kLdsBytes ? (unsigned)kRingOff : nxt + kUnitBytes;
  };
  auto end_interval = [&]() TN_INL { advance_dma(); };

  // ================= state that lives across slots =================
  u32x4 xb[2][2];        // BN1 + ReLU'd pixel fragments [k-step parity][fragment]: produced one k-step ahead of the MFMAs
  u32x4 wa[4];           // 1x1 weight fragments [32-channel block]: reloaded for the next k-step behind the block's second MFMA
  u32x4 wsh[4];          // ... of the shift k-step [block]: read from the last 3x3 unit of the pass before, used by the pass' first slots
  u32x4 cb[2];           // BN1 constants of dword J of the k-step in production [J parity]: .x = (a0, a1), .y = (b0, b1) packed halves
  f32x16 acc[4][2];      // 1x1 accumulators [block][fragment]
  u32x4 w3f[2][3];       // 3x3 weight fragments [step parity][dx]
  u32x4 bop[2][2];       // 3x3 pixel fragments [step parity][fragment]
  f32x16 bacc[3][2];     // 3x3 accumulators [dx][fragment]
  unsigned rbase[4];     // tile byte offsets (lt0 included) of bottleneck rows 8 p + 2 wid - 2 + k, k = 0 .. 3, of the pass
  u32x4 xb_shift;        // the pixel fragment of the shift k-step of the pass whose accumulators are being started

  // ---- ring: asm loads into literal registers + counted waits ----
  const unsigned char *rb_a = scr, *rb_b = scr;   // plane base (pass offset included) of the two refill targets of an interval
  auto ring_load = [&](auto rs_tag, auto k_tag, auto f_tag, const unsigned char *base) TN_INL {
    constexpr int KQ = decltype(k_tag)::value, F = decltype(f_tag)::value, R = ring_reg(decltype(rs_tag)::value, KQ, F);
    const unsigned vo = voff[F];
    const unsigned char *pb = base + KQ * kPlaneB;
    asm volatile("global_load_dwordx4 v[%c0:%c1], %2, %3" ::"n"(R), "n"(R + 3), "v"(vo), "s"(pb) : TN_RING_CLOBBER);
  };
  auto ring_wait = [&]() TN_INL { asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(kVmRing) : TN_RING_CLOBBER); };
  // BN1 + ReLU of one dword (two channels) of a pixel fragment: relu(a x + b) with fp16 constants, fused multiply-add, packed max
  auto bn_ring = [&](auto reg_tag, auto j_tag) TN_INL -> unsigned {      // input: ring register REG
    constexpr int J = decltype(j_tag)::value, REG = decltype(reg_tag)::value;
    const u32x4 c = cb[J & 1];
    unsigned o;
    asm volatile("v_pk_max_f16 %0, v%c3, %1\n\tv_pk_min_f16 %0, %0, %2" : "=&v"(o) : "v"(c.x), "v"(c.y), "n"(REG) : TN_RING_CLOBBER);
    return o;
  };
  auto consts_read = [&](const unsigned vc, auto kq_tag, auto j_tag) TN_INL {
    constexpr int KQ = decltype(kq_tag)::value, J = decltype(j_tag)::value;
    cb[J & 1] = *(const u32x4 *)(smem + vc + 128 * KQ + 16 * J);
  };
  auto wa_read = [&](auto mb_tag, const unsigned vb, auto kq_tag) TN_INL {
    constexpr int MB = decltype(mb_tag)::value, KQ = decltype(kq_tag)::value;
    wa[MB] = *(const u32x4 *)(smem + vb + (KQ * 4 + MB) * 1024);
  };

  // ---- epilogue A: block MB's accumulators -> ReLU -> fp16 -> the tile rows of the pass' own bottleneck rows (rbase[2 + F]);
  // 20 items per block: (fragment, tuple) x (4 converts + the write) ----
  unsigned e_pk[4];
  auto epa_item = [&](auto mb_tag, auto e_tag) TN_INL {
    constexpr int MB = decltype(mb_tag)::value, E = decltype(e_tag)::value;
    constexpr int F = E / 10, T2 = (E / 5) & 1, I = E % 5, T = 2 * MB + T2, R0 = 8 * T2;
    unsigned (&epk)[4] = e_pk;
    f32x16 (&accr)[4][2] = acc;
    if constexpr (I < 4) {
      const float a0 = accr[MB][F][R0 + 2 * I], a1 = accr[MB][F][R0 + 2 * I + 1];
      asm("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=v"(epk[I]) : "v"(a0), "v"(a1));
    } else {
      *(u32x4 *)(smem + (rbase[2 + F] ^ (32u * T))) = u32x4{epk[0], epk[1], epk[2], epk[3]};
    }
  };
  auto epa_items = [&](auto mb_tag, auto k0_tag, auto n_tag) TN_INL {
    static_for<decltype(n_tag)::value>([&](auto k_tag) TN_INL { epa_item(mb_tag, ic<decltype(k0_tag)::value + decltype(k_tag)::value>{}); });
  };
  auto w3_read = [&](auto s_tag, auto dx_tag, const unsigned vb) TN_INL {      // fragment (step S, dx) = fragment 3 (S & 3) + dx of its unit
    constexpr int S = decltype(s_tag)::value, DX = decltype(dx_tag)::value;
    w3f[S & 1][DX] = *(const u32x4 *)(smem + vb + ((S & 3) * 3 + DX) * 1024);
  };
  // pixel fragment of step S = (kernel row index, tuple): kernel rows in the order dy = +1, 0, -1; output row of fragment F is
  // bottleneck row (8 p + 2 wid - 2) + 1 + F, so it reads rbase[1 + F + dy]
  auto bop_read = [&](auto s_tag, auto f_tag) TN_INL {
    constexpr int S = decltype(s_tag)::value, F = decltype(f_tag)::value, DYI = S / 8, T = S % 8;
    constexpr int DY = 1 - DYI;
    bop[S & 1][F] = *(const u32x4 *)(smem + (rbase[1 + F + DY] ^ (32u * T)));
  };

  // ================= a super-step interval (dense_block14.hip::su_interval without the forwarded channels) =================
  // consumes unit g = super-step u of the pass (ring slot RS): 4 k-steps x 4 blocks x 2 fragments = 32 MFMA slots; slot (q, mb, f),
  // e = 2 mb + f, carries the reload of the block's weight register for k-step q + 1, ONE item of the BN1 pipeline (k-step q + 1;
  // during q = 3 k-step 0 of the NEXT unit), the constants two dwords on, the ring refills, one DMA statement behind k-steps 0, 1, 3.
  //   KIND 0: first super-step of a pass, 1: inner, 2: last - the next unit is the first 3x3 unit: its k-step 3 carries no BN items
  //   but epilogue A of block 0 (final after the block's k-step-3 MFMAs) and the 3x3's first operands
  auto su_interval = [&](auto rs_tag, auto kind_tag, auto vm_tag) TN_INL {
    constexpr int RS = decltype(rs_tag)::value, KIND = decltype(kind_tag)::value;
    constexpr bool LAST = KIND == 2;
    begin_interval(vm_tag);
    if constexpr (KIND == 0) {      // the shift k-step: the pass' accumulators start from W_shift x (1, 1, mask, 0 ...)
      f32x16 z;
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = 0.f;
      static_for<8>([&](auto i_tag) TN_INL {
        constexpr int I = decltype(i_tag)::value, MB = I >> 1, F = I & 1;
        acc[MB][F] = mfma32(wsh[MB], xb_shift, z);
        TN_RING_FENCE();
        TN_SB();
      });
    }
    static_for<32>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value, Q = I >> 3, MB = (I >> 1) & 3, F = I & 1, E = I & 7;
      acc[MB][F] = mfma32(wa[MB], xb[Q & 1][F], acc[MB][F]);
      // the block's weight fragment of k-step q + 1
      if constexpr (F == 1 && !(LAST && Q == 3)) {
        if constexpr (Q < 3) wa_read(ic<MB>{}, vb_cur, ic<Q + 1>{});
        else wa_read(ic<MB>{}, vb_next, ic<0>{});
      }
      // BN1 item e of k-step q + 1
      if constexpr (!(LAST && Q == 3)) {
        constexpr int J = E >> 1, BF = E & 1;
        if constexpr (Q < 3) {
          if constexpr (E == 0) ring_wait();
          xb[(Q + 1) & 1][BF][J] = bn_ring(ic<ring_reg(RS, Q + 1, BF) + J>{}, ic<J>{});
          if constexpr (J == 3) ring_load(ic<RS>{}, ic<Q + 1>{}, ic<BF>{}, rb_a);
        } else {
          if constexpr (E == 0) ring_wait();
          xb[0][BF][J] = bn_ring(ic<ring_reg(RS ^ 1, 0, BF) + J>{}, ic<J>{});
          if constexpr (J == 3) ring_load(ic<RS ^ 1>{}, ic<0>{}, ic<BF>{}, rb_b);
        }
        if constexpr (BF == 1 && !(LAST && Q == 2 && J >= 2)) {
          if constexpr (J < 2) {
            if constexpr (Q < 3) consts_read(vc_cur, ic<Q + 1>{}, ic<J + 2>{});
            else consts_read(vc_next, ic<0>{}, ic<J + 2>{});
          } else {
            if constexpr (Q < 2) consts_read(vc_cur, ic<Q + 2>{}, ic<J - 2>{});
            else consts_read(vc_next, ic<Q - 2>{}, ic<J - 2>{});
          }
        }
      } else {
        // last k-step of the pass' 1x1: the 3x3's first weight fragments (steps 0 / 1 of unit g + 1), epilogue A of block 0, and
        // the pixel fragments of step 0 (tuple 0, own rows: written by this wave a few slots earlier)
        if constexpr (I - 24 < 6) w3_read(ic<(I - 24) / 3>{}, ic<(I - 24) % 3>{}, vb_next);
        if constexpr (I >= 26 && I < 31) epa_items(ic<0>{}, ic<(I - 26) * 4>{}, ic<4>{});
        if constexpr (I == 31) { bop_read(ic<0>{}, ic<0>{}); bop_read(ic<0>{}, ic<1>{}); }
        TN_RING_FENCE();
      }
      if constexpr (E == 7) {
        if constexpr (Q == 0) dma_pair(ic<0>{});
        else if constexpr (Q == 1) dma_pair(ic<1>{});
        else if constexpr (Q == 3) dma_consts();
      }
      TN_SB();
    });
    end_interval();
  };

  // ================= the 3x3 intervals: J = 0 .. 5, steps 4 J .. 4 J + 3, slot (step, dx, f) =================
  // J = 0 / 1 carry epilogue A of blocks 1 - 3 (each ahead of the first read of its tuples); the barrier of J = 2 publishes the tile
  // (kernel row 0, steps 8 .., is the first that needs a neighbour's row: step 8 reads fragment X's operand behind that barrier and
  // runs fragment Y first); J = 5 carries the head of the NEXT pass' BN1 pipeline (pre_item) and reads the next pass' shift
  // fragments (fragments 12 .. 15 of the unit) into registers.
  auto pre_item = [&](auto pn_tag, auto i_tag) TN_INL {      // the next pass' k-step 0 (ring slot PN): constants, 8 BN items, weights
    constexpr int PN = decltype(pn_tag)::value, I = decltype(i_tag)::value;
    if constexpr (I < 2) {
      consts_read(vc_next, ic<0>{}, ic<I>{});
    } else if constexpr (I < 10) {
      constexpr int E = I - 2, J = E >> 1, BF = E & 1;
      if constexpr (E == 0) ring_wait();
      xb[0][BF][J] = bn_ring(ic<ring_reg(PN, 0, BF) + J>{}, ic<J>{});
      if constexpr (J == 3) ring_load(ic<PN>{}, ic<0>{}, ic<BF>{}, rb_a);
      if constexpr (BF == 1) {
        if constexpr (J < 2) consts_read(vc_next, ic<0>{}, ic<J + 2>{});
        else consts_read(vc_next, ic<1>{}, ic<J - 2>{});
      }
    } else {
      wa_read(ic<I - 10>{}, vb_next, ic<0>{});
    }
  };
  constexpr int kPreItems = 14;
  auto wsh_read = [&](auto mb_tag, const unsigned vb) TN_INL {      // fragments 12 .. 15 of the unit at vb
    constexpr int MB = decltype(mb_tag)::value;
    wsh[MB] = *(const u32x4 *)(smem + vb + (12 + MB) * 1024);
  };
  auto b_interval = [&](auto j_tag, auto pn_tag, auto vm_tag) TN_INL {
    constexpr int J = decltype(j_tag)::value;
    // the tile writes of every wave (the last ones sit in J = 1) have to be complete, not only issued, when the wave arrives
    if constexpr (J == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    begin_interval(vm_tag);
    static_for<24>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value, S = 4 * J + I / 6, P = I % 6;
      constexpr bool YFIRST = S == 8;                       // slot order (dx, f), or fragment Y's three MFMAs first
      constexpr int DX = YFIRST ? P % 3 : P >> 1, F = YFIRST ? (P < 3 ? 1 : 0) : (P & 1);
      constexpr bool SECOND_USE = YFIRST ? P >= 3 : F == 1;  // of the weight register w3f[S & 1][DX]
      if constexpr (P == 0) {
        u32x4 (&w)[3] = w3f[S & 1];
        asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]));
      }
      if constexpr (S == 0) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        bacc[DX][F] = mfma32(w3f[0][DX], bop[0][F], z);
      } else {
        bacc[DX][F] = mfma32(w3f[S & 1][DX], bop[S & 1][F], bacc[DX][F]);
      }
      if constexpr (SECOND_USE && S + 2 < 24) {      // weight fragments of step S + 2: this unit or the next
        if constexpr (((S + 2) >> 2) == J) w3_read(ic<S + 2>{}, ic<DX>{}, vb_cur);
        else w3_read(ic<S + 2>{}, ic<DX>{}, vb_next);
      }
      // pixel fragments of step S + 1, three slots ahead; step 8's fragment X (the first operand with a neighbour's row) behind
      // the barrier that publishes the tile, at the top of the step
      if constexpr (S + 1 < 24 && (P == 2 || P == 3) && !(S + 1 == 8 && P == 2)) bop_read(ic<S + 1>{}, ic<P - 2>{});
      if constexpr (S == 8 && P == 0) bop_read(ic<8>{}, ic<0>{});
      // epilogue A of blocks 1, 2, 3: 20 items each, ahead of the first read of their tuples
      if constexpr (J == 0 && I < 8) epa_items(ic<1>{}, ic<I * 5 / 2>{}, ic<(I + 1) * 5 / 2 - I * 5 / 2>{});
      if constexpr (J == 0 && I >= 8 && I < 20) epa_items(ic<2>{}, ic<(I - 8) * 5 / 3>{}, ic<(I - 7) * 5 / 3 - (I - 8) * 5 / 3>{});
      if constexpr (J == 0 && I >= 20) epa_items(ic<3>{}, ic<(I - 20) * 5 / 3>{}, ic<(I - 19) * 5 / 3 - (I - 20) * 5 / 3>{});
      if constexpr (J == 1 && I < 8) epa_items(ic<3>{}, ic<(I + 4) * 5 / 3>{}, ic<(I + 5) * 5 / 3 - (I + 4) * 5 / 3>{});
      if constexpr (J == 5 && I < kPreItems) pre_item(pn_tag, ic<I>{});     // (ahead of the interval's DMA statements: kVmRing counts on it)
      if constexpr (J == 5 && I >= 16 && I < 20) wsh_read(ic<I - 16>{}, vb_cur);      // the next pass' shift fragments
      if constexpr (I == 15) dma_pair(ic<0>{});
      if constexpr (I == 19) dma_pair(ic<1>{});
      if constexpr (I == 23) dma_consts();
      TN_RING_FENCE();
      TN_SB();
    });
    end_interval();
  };

  // ---- epilogue B: out[x] = acc[dx = 1][x] + acc[dx = 0][x - 1] + acc[dx = 2][x + 1] (whole-wave DPP shifts: a row is the 32 lanes
  // of one half; what crosses into slot 0 / 31 belongs to padding columns, never stored), fp16: the lane's 16 output channels
  // 16 h .. 16 h + 15 -> 32 B to the concat buffer and 2 x 16 B to the k-step-major copy ----
  auto epilogue_b = [&](int K, int p) TN_INL {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      unsigned o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float c0 = bacc[1][f][2 * q], c1 = bacc[1][f][2 * q + 1];
        c0 += dpp_f32<0x138>(bacc[0][f][2 * q]);          // wave_shr:1: lane x reads lane x - 1
        c1 += dpp_f32<0x138>(bacc[0][f][2 * q + 1]);
        c0 += dpp_f32<0x130>(bacc[2][f][2 * q]);          // wave_shl:1: lane x reads lane x + 1
        c1 += dpp_f32<0x130>(bacc[2][f][2 * q + 1]);
        const h2_t pk = {(f16)c0, (f16)c1};
        o[q] = __builtin_bit_cast(unsigned, pk);
        TN_RING_FENCE();      // (the ring holds the next pass' first super-steps: no temporaries in its registers)
      }
      const int orow = 8 * p + 2 * wid - 1 + f;
      const bool ok = colvalid && orow >= 0 && orow < 28;
      const unsigned px = spix[f] + (unsigned)(p * 8 * 28);
      const unsigned so = ok ? px * (unsigned)ldc * 2 + 32 * h : 0x80000000u;
      const unsigned ss = ok ? px * 32 + (unsigned)h * kPlaneB : 0x80000000u;
      const u32x4 v0 = u32x4{o[0], o[1], o[2], o[3]}, v1 = u32x4{o[4], o[5], o[6], o[7]};
      __builtin_amdgcn_raw_buffer_store_b128(v0, orsrc, so, 2 * K, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v1, orsrc, so + 16, 2 * K, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v0, srsrc, ss, (K >> 4) * kPlaneB, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v1, srsrc, ss + 16, (K >> 4) * kPlaneB, 0);
    }
  };

  // the tile rows and the shift fragment of pass p (the accumulators of pass p are started one pass early: set_shift(p) runs before
  // the shift MFMAs, set_rows(p) at the top of pass p)
  auto set_rows = [&](int p) TN_INL {
    const int b0 = (8 * p + 2 * wid) % kTileRows;       // tile row of bottleneck row 8 p + 2 wid - 2
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = b0 + k >= kTileRows ? b0 + k - kTileRows : b0 + k;
      rbase[k] = haslot ? (unsigned)r * kTileRowB + lt0 : lt0;
    }
  };
  auto set_shift = [&](int p) TN_INL {
    // (1, 1, mask, 0 ...) in the lanes that hold k = 0 .. 7; mask = -60000 where the slot is padding: a padding column, or a row
    // that does not exist (rows 28 .. 31 of the last pass: both fragments of a wave together)
    const bool valid = colvalid && 8 * p + 2 * wid < 28;
    xb_shift = u32x4{h == 0 ? 0x3c003c00u : 0u, (h == 0 && !valid) ? 0x0000fb53u : 0u, 0u, 0u};
  };

  // ================= prologue =================
  {
    // tile rows 0 / 1 (bottleneck rows -2 / -1 of the first pass) and the dump area: zeros
    for (int i = tid; i < (2 * kTileRowB) / 16; i += 256) *(u32x4 *)(smem + i * 16) = u32x4{0, 0, 0, 0};
    if (tid < kDumpBytes / 16) *(u32x4 *)(smem + kTileBytes + tid * 16) = u32x4{0, 0, 0, 0};
    for (int u = 0; u < 4; ++u) {      // units 0 .. 3 (unit 0: the first pass' shift fragments)
      dma_pair(ic<0>{});
      dma_pair(ic<1>{});
      dma_consts();
      advance_dma();
    }
    // channels 0 .. K0 - 1 of the frame: NHWC -> the k-step-major copy
    // (seven loads in flight per round trip: 784 x K0 / 8 sixteen-byte pieces = 7 x 7 x 256 at K0 = 128)
    const int nq = a.K0 / 16, npiece = kPix * nq * 2;
    auto src_of = [&](int i) TN_INL { return fb + (size_t)((i >> 1) / nq) * ldc * 2 + 32 * ((i >> 1) % nq) + 16 * (i & 1); };
    auto dst_of = [&](int i) TN_INL { return scr + (size_t)((i >> 1) % nq) * kPlaneB + ((i >> 1) / nq) * 32 + 16 * (i & 1); };
    int i0 = tid;
    for (; i0 + 6 * 256 < npiece; i0 += 7 * 256) {
      u32x4 v[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) v[k] = *(const u32x4 *)src_of(i0 + k * 256);
#pragma unroll
      for (int k = 0; k < 7; ++k) *(u32x4 *)dst_of(i0 + k * 256) = v[k];
    }
    for (int i = i0; i < npiece; i += 256) *(u32x4 *)dst_of(i) = *(const u32x4 *)src_of(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the first pass' ring: super-steps 0 (slot 0) and 1 (slot 1)
    static_for<16>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value;
      ring_load(ic<(I >> 3)>{}, ic<((I >> 1) & 3)>{}, ic<(I & 1)>{}, scr + 4 * kPlaneB * (I >> 3));
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: TN_RING_CLOBBER);
    // "interval 0": unit 0 = the first pass' shift fragments; unit 1 is the next unit
    vb_cur = kRingOff + lane16; vc_cur = kRingOff + kUnitFrag + 64 * h;
    vb_next = kRingOff + kUnitBytes + lane16; vc_next = kRingOff + kUnitBytes + kUnitFrag + 64 * h;
    nxt = kRingOff + 2 * kUnitBytes;
    set_shift(0);
    static_for<4>([&](auto mb_tag) TN_INL { wsh_read(mb_tag, vb_cur); });
    {   // the head of the pipeline; its refill: super-step 2 of the first pass, or super-step 0 of the second one (K0 = 128)
      const int nsu0 = (a.K0 + 63) >> 6;
      rb_a = uniform_ptr(nsu0 > 2 ? scr + 4 * kPlaneB * 2 : scr + kPassB);
    }
    static_for<kPreItems>([&](auto i_tag) TN_INL { pre_item(ic<0>{}, i_tag); });
    dma_pair(ic<0>{});
    dma_pair(ic<1>{});
    dma_consts();
    end_interval();
    asm volatile("s_waitcnt vmcnt(0)" ::: TN_RING_CLOBBER);
    TN_SB();
  }

  // ================= the block =================
  // stage = (layer, pass); super-step u of the stage `ahead` stages on lives at scr + 4 planes * u + its pass' row offset
  int par = 0;                                   // ring slot of the stage's super-step 0
  for (int l = 0; l < a.nl; ++l) {
    const int K = a.K0 + 32 * l;
    const int nsu = (K + 63) >> 6;               // super-steps (channels rounded up; the pad has zero weights and constants)
    for (int p = 0; p < 4; ++p) {
      // refill target of "super-step u of this stage" for u up to nsu + 2, wrapping into the (at most two) stages behind it - a
      // pixel's channels do not move between layers; past the last stage of the block the loads are never consumed: any valid
      // address.  Straight-line selects: a loop here puts control flow between the slots, and hipcc splits (and spills) the
      // accumulators' live ranges at every block boundary (first version: 132 spilled registers)
      const int n1 = p < 3 ? nsu : (K + 32 + 63) >> 6;      // super-steps of the next stage
      auto target = [&](int u) TN_INL -> const unsigned char * {
        const bool w1 = u >= nsu;
        const int u1 = w1 ? u - nsu : u;
        const bool w2 = w1 && u1 >= n1;
        const int u2 = w2 ? u1 - n1 : u1;
        const int pp = (p + (w1 ? 1 : 0) + (w2 ? 1 : 0)) & 3;
        return uniform_ptr(scr + 4 * kPlaneB * u2 + kPassB * pp);
      };
      auto refill_bases = [&](int u) TN_INL { rb_a = target(u + 2); rb_b = target(u + 3); };
      set_rows(p);
      auto front = [&](auto p_tag) TN_INL {
        constexpr int P = decltype(p_tag)::value;
        refill_bases(0);
        su_interval(ic<P>{}, ic<0>{}, ic<kVmDmaSU0>{});
        int u = 1;
        for (; u + 1 < nsu - 1; u += 2) {
          refill_bases(u);
          su_interval(ic<P ^ 1>{}, ic<1>{}, ic<kVmDmaSU>{});
          refill_bases(u + 1);
          su_interval(ic<P>{}, ic<1>{}, ic<kVmDmaSU>{});
        }
        if (u < nsu - 1) {
          refill_bases(u);
          su_interval(ic<P ^ 1>{}, ic<1>{}, ic<kVmDmaSU>{});
          refill_bases(u + 1);
          su_interval(ic<P>{}, ic<2>{}, ic<kVmDmaSU>{});
        } else {
          refill_bases(u);
          su_interval(ic<P ^ 1>{}, ic<2>{}, ic<kVmDmaSU>{});
        }
      };
      if (par) front(ic<1>{});
      else front(ic<0>{});
      b_interval(ic<0>{}, ic<0>{}, ic<kVmDmaB0>{});
      b_interval(ic<1>{}, ic<0>{}, ic<kVmDmaB>{});
      b_interval(ic<2>{}, ic<0>{}, ic<kVmDmaB>{});
      b_interval(ic<3>{}, ic<0>{}, ic<kVmDmaB>{});
      b_interval(ic<4>{}, ic<0>{}, ic<kVmDmaB>{});
      par = (par + nsu) & 1;
      // the head of the next stage's pipeline refills ITS super-step 2 (k-step 0); its accumulators start from its shift k-step
      {
        const int pn = (p + 1) & 3;
        const bool wrap2 = n1 <= 2;                 // the next stage has two super-steps: its "super-step 2" is super-step 0 of the stage after
        rb_a = uniform_ptr(scr + (wrap2 ? 0 : 4 * kPlaneB * 2) + kPassB * (wrap2 ? (pn + 1) & 3 : pn));
        set_shift(pn);
      }
      if (par) b_interval(ic<5>{}, ic<1>{}, ic<kVmDmaB>{});
      else b_interval(ic<5>{}, ic<0>{}, ic<kVmDmaB>{});
      epilogue_b(K, p);
    }
    if (a.ts && tid == 0 && l < 62) a.ts[(size_t)blockIdx.x * 64 + l] = __builtin_amdgcn_s_memtime();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_memtime();
}

}  // namespace

bool dense_block28_supported(int H, int W, int K0, int nl) {
  return H == 28 && W == 28 && K0 % 32 == 0 && K0 >= 128 && nl >= 1 && K0 + 32 * (nl - 1) <= 512;
}

int dense_block28_units(int K0, int nl) {      // (the prologue unit in front, four units of padding behind: the last intervals' DMA reads them)
  int n = 1 + 4;
  for (int l = 0; l < nl; ++l) n += 4 * ((K0 + 32 * l + 63) / 64 + 6);
  return n;
}

size_t dense_block28_scratch_halfs() { return (size_t)kFrameScrB / 2; }

int launch_dense_block28(const DenseBlock28Args &a, hipStream_t s) {
  TN_REQUIRE(a.buf && a.stream && a.scratch, "dense_block28: null operand");
  TN_REQUIRE(dense_block28_supported(28, 28, a.K0, a.nl) && a.ldc % 64 == 0 && a.K0 + 32 * a.nl <= a.ldc && a.B > 0, "dense_block28: unsupported geometry");
  TN_REQUIRE(a.total_units == dense_block28_units(a.K0, a.nl), "dense_block28: stream does not match the block");
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_block28_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes)));
  hipLaunchKernelGGL(dense_block28_kernel, dim3(a.B), dim3(256), kLdsBytes, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// ---- host-side packing: the block's weight stream ----
// unit 0: fragments 12 .. 15 = the shift k-step of layer 0 (pass 0).  Then per layer, four times (once per pass):
//   ceil(K / 64) super-step units: fragment (q, mb): lane l, j: bottleneck channel 32 mb + (l & 31), input channel c = 64 u + 16 q
//     + 8 (l >> 5) + j (zero weight and zero constants for c >= K); constants (q, h), dword J: halves a1[c + 2 J], a1[c + 2 J + 1],
//     b1[c + 2 J], b1[c + 2 J + 1] (+ 8 B unused) for c = 64 u + 16 q + 8 h (s1 / t1 of Block14Layer are those fp16 numbers);
//   six 3x3 units J: fragments (step 4 J + s, dx) at s * 3 + dx, s = 0 .. 3: kernel rows in the order ky = 2, 1, 0 (dy = +1, 0, -1),
//     tuple t = step % 8, lane layout as pack_w3_strip (dense_strip.hip); unit J = 5 also carries, in fragments 12 .. 15, the shift
//     k-step (dense_strip.hip: fp16 hi + lo of BN2's shift, and 1 for the mask) of the NEXT pass: this layer's, or behind the
//     fourth pass the next layer's.
std::vector<unsigned char> pack_block28(const std::vector<Block14Layer> &layers, int K0) {
  const int nl = (int)layers.size();
  std::vector<unsigned char> out((size_t)dense_block28_units(K0, nl) * kUnitBytes, 0);
  size_t unit = 0;
  auto frag = [&](size_t u, int fi) { return (f16 *)(out.data() + u * kUnitBytes + (size_t)fi * 1024); };
  auto cons = [&](size_t u, int q, int h) { return (f16 *)(out.data() + u * kUnitBytes + kUnitFrag + (q * 2 + h) * 64); };
  auto put_const = [](f16 *d, int j, float a, float b) {
    d[8 * (j >> 1) + (j & 1)] = (f16)a;
    d[8 * (j >> 1) + 2 + (j & 1)] = (f16)b;
  };
  auto put_shift = [&](size_t u, const Block14Layer &L) {
    for (int mb = 0; mb < 4; ++mb) {
      f16 *d = frag(u, 12 + mb);
      for (int ln = 0; ln < 32; ++ln) {
        const float t = L.t2[32 * mb + ln];
        d[ln * 8 + 0] = (f16)t;
        d[ln * 8 + 1] = (f16)(t - (float)d[ln * 8 + 0]);
        d[ln * 8 + 2] = (f16)1.f;
      }
    }
  };
  put_shift(unit++, layers[0]);
  for (int l = 0; l < nl; ++l) {
    const Block14Layer &L = layers[l];
    const int K = K0 + 32 * l, nsu = (K + 63) / 64;
    for (int p = 0; p < 4; ++p) {
      for (int u = 0; u < nsu; ++u, ++unit)
        for (int q = 0; q < 4; ++q) {
          for (int mb = 0; mb < 4; ++mb) {
            f16 *d = frag(unit, q * 4 + mb);
            for (int ln = 0; ln < 64; ++ln)
              for (int j = 0; j < 8; ++j) {
                const int c = 64 * u + 16 * q + 8 * (ln >> 5) + j;
                d[ln * 8 + j] = c < K ? (f16)L.w1f[(size_t)(32 * mb + (ln & 31)) * K + c] : (f16)0.f;
              }
          }
          for (int h = 0; h < 2; ++h) {
            f16 *d = cons(unit, q, h);
            for (int j = 0; j < 8; ++j) {
              const int c = 64 * u + 16 * q + 8 * h + j;
              put_const(d, j, c < K ? L.s1[c] : 0.f, c < K ? L.t1[c] : 0.f);
            }
          }
        }
      for (int J = 0; J < 6; ++J, ++unit) {
        for (int s = 0; s < 4; ++s) {
          const int step = 4 * J + s, ky = 2 - step / 8, t = step % 8;
          for (int dx = 0; dx < 3; ++dx) {
            f16 *d = frag(unit, s * 3 + dx);
            for (int ln = 0; ln < 64; ++ln)
              for (int j = 0; j < 8; ++j) {
                const int m = ln & 31, o = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
                const int c = 16 * t + 8 * (j >> 2) + 4 * (ln >> 5) + (j & 3);
                d[ln * 8 + j] = (f16)L.w3[(((size_t)o * 128 + c) * 3 + ky) * 3 + dx];
              }
          }
        }
        if (J == 5) {
          if (p < 3) put_shift(unit, L);
          else if (l + 1 < nl) put_shift(unit, layers[l + 1]);
        }
      }
    }
  }
  return out;
}

// The 14x14 dense block (24 layers at 224x224 input) as ONE launch of pixel-owning waves (round 4).
//
//   for l in 0 .. nl-1:   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) ),  K = K0 + 32 l
//
// (reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_dense_block / _make_dense_layer.)
//
// dense_layer_big.hip ran this block as a chain of barrier-phased tiles: K loop, epilogue, 3x3 and store one after the other, the
// waves that issue the ring's DMA being the waves that feed the matrix pipe, 23 % of the MFMA peak.  The strip
// kernel's recipe (dense_strip_impl.h) does not carry over as it is: a 14x14 frame gives a SIMD 49 pixels, so there is no strip
// to walk down and no room for resident weights (K = 256 .. 992: 64 .. 248 KB of 1x1 weights per layer).  What does carry over:
//
// * a workgroup = one frame, 4 waves = ONE wave per SIMD with the whole register file.  Wave w owns image rows 4w .. 4w + 3 as
//   two N = 32 fragments of v_mfma_f32_32x32x16_f16: X = [row r0 | row r0 + 2], Y = [row r0 + 1 | row r0 + 3] (16 slots per
//   row: 14 pixels + the two padding columns).  Every 1 KiB weight fragment read from LDS feeds TWO MFMAs.
//   Wave 3's rows 14 / 15 do not exist: 196 of 256 slots are real (the same 77 % the tile kernel had: 49 pixels per SIMD round up
//   to 64 under any N = 16 granularity).
// * activations go HBM -> registers directly in fragment shape through a register ring of two 64-channel super-steps that runs
//   THROUGH layer boundaries: a wave only ever reads its own pixels, and those of the next layer's first 128 channels are
//   requested while this layer's last super-steps are consumed.  The newest 32 channels never come back from memory: the 3x3's
//   result registers of layer l are the pixel fragments of two k-steps of layer l + 1 (the 1x1 weights of those k-steps are
//   packed in the matching channel order); they are stored for the layers after that and for the transition.
// * ALL weights stream: the block's 1x1 fragments, BN1 constants and 3x3 fragments are one linear sequence of 16.5 KiB UNITS in
//   consumption order (pack_block14), copied by LDS-DMA into a ring of five slots, every wave issuing a quarter of each unit
//   four units ahead.  One s_barrier per unit (32 / 24 MFMAs per wave) publishes unit g + 1 and frees the slot of unit g - 1;
//   fragment reads run two k-steps ahead across unit boundaries.
// * the bottleneck goes accumulator -> ReLU -> fp16 -> a (18 rows x 16 slots x 256 B) LDS tile in the chained-MFMA channel order
//   of the strip kernel (BN2's scale folded into the 1x1 weights, its shift and the padding mask in one extra k-step), and the
//   3x3 reads its pixel fragments from there: row r0 - 1 / r0 + 4 come from the neighbour waves, zero rows above / below the
//   image are rows of the tile.  The three kernel columns go to three accumulator sets that are combined by two DPP row shifts.
// * every vector-memory LOAD of the steady state is inline asm with hand-counted s_waitcnt vmcnt(N): the ring loads (consumed two
//   super-steps later: 24 younger loads) and the DMA pieces (consumed three units later).  tests/test_cpu_block14.py replays
//   the issue order of a whole block and proves every count.  The ring registers are only ever touched by asm statements
//   (loads, waits, BN1): scripts/audit_block14_isa.py fails the build if hipcc copies one of them while its load is in flight.
#include <array>
#include <type_traits>
#include <utility>

#include "common.h"

#ifndef TN_B14_STAMPS
#define TN_B14_STAMPS 0   // tuning build: wave 0 also stamps the start of every layer's tail and 3x3 phase and the end of its 3x3 phase (ts[64 ..])
#endif
#ifndef TN_B14_EXP
#define TN_B14_EXP 0   // timing experiments only (results wrong): bit 0 no ring loads, bit 1 no 3x3 phase, bit 2 no BN items, bit 3 no DMA, bit 4 ring loads read 1 KiB contiguous per instruction
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

constexpr int kUnitFrag = 16384;                  // 16 A fragments of 1 KiB
constexpr int kUnitBytes = kUnitFrag + 512;       // + BN1 constants of the unit's k-steps: [k-step][lane >> 5][dword J: s[2J], s[2J+1], t[2J], t[2J+1]] fp32
constexpr int kNR = 5;                            // ring slots
constexpr int kTileRowB = 4096;                   // 16 slots x 256 B (128 bottleneck channels, fp16)
constexpr int kTileRows = 18;                     // row 0: zeros above the image, 1 .. 14 the image, 15: zeros below, 16 / 17: wave 3's rows that do not exist
constexpr int kTileBytes = kTileRows * kTileRowB;
constexpr int kLdsBytes = kTileBytes + kNR * kUnitBytes;
constexpr int kPlaneB = 196 * 32;                 // one k-step (16 channels) of a frame in the private k-step-major copy
constexpr int kFrameScrB = 64 * kPlaneB;          // 1024 channels
static_assert(kLdsBytes <= 160 * 1024, "LDS");

// s_waitcnt vmcnt(N) constants (asm loads only; tests/test_cpu_block14.py derives every one of them from the issue order)
constexpr int kVmRing = 24;        // a ring register pair is waited for two super-step intervals (2 x 13 loads) after its refills, at the slot of
                                   // the FIRST of its two loads' k-step: 26 - 2 loads lie behind the second one (25 was one too many: the replay test)
constexpr int kVmDmaSU0 = 12, kVmDmaSU = 20, kVmDmaTail = 24, kVmDmaB0 = 16, kVmDmaB = 10;

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

// LDS-DMA: global (wave-uniform base in SGPRs + per-lane 32-bit offset) -> LDS (M0 + lane * size).  The instruction offset
// applies to the global AND the LDS address (scripts/microbench/dmaoff.hip), so two 1-KiB pieces share one M0.
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

// The activation ring lives in LITERAL registers v[192:255]: its loads are in flight for two super-step intervals, and a value
// hipcc knows about may be copied or spilled at any time - with the bytes of a pending load not there yet (the first version of
// this kernel, ring in compiler-allocated registers tied through the waits: "scratch_store_dwordx4 v[28:31]" one instruction
// behind the load that fills v[28:31]).  Every slot names the 64 registers as clobbered, which keeps compiler values out of them
// (the technique of the strip kernel's accumulator window); scripts/audit_block14_isa.py checks the ISA for strays.
#define TN_RING_BASE 192
#define TN_RING_CLOBBER                                                                                                             \
  "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",   \
  "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",   \
  "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",   \
  "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define TN_RING_FENCE() asm volatile("" ::: TN_RING_CLOBBER)
constexpr int ring_reg(int rs, int kq, int f) { return TN_RING_BASE + ((rs * 4 + kq) * 2 + f) * 4; }   // [super-step parity][k-step][fragment] x 4 dwords

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_block14_kernel(DenseBlock14Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5, st = n >> 4, slot16 = n & 15;
  const int r0 = 4 * wid;
  const int ldc = a.ldc;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);
  unsigned char *fb = (unsigned char *)(a.buf + (size_t)blockIdx.x * 196 * ldc);
  if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memtime();

  // ---- per-lane geometry: fragment F (0 = X, 1 = Y) holds row r0 + F + 2 st, column slot16 - 1 ----
  // The block works on a PRIVATE copy of the frame in k-step-major layout, scr[k-step = channel / 16][pixel][16 channels]: a
  // fragment load (lane = (pixel slot, half of the k-step), 16 B) then reads two runs of 448 contiguous bytes (eight 128-byte
  // lines) instead of 32 B out of each of 32 lines of the NHWC buffer - the texture-address path, not HBM, bounded the first
  // version of this kernel (and bounds the strip kernel): 72-81 cycles per MFMA slot in the 1x1 phase, 52-56 with contiguous
  // loads (TN_B14_EXP bit 4).  Channels 0 .. K0 - 1 are copied there in the prologue, every layer appends its 32 channels to
  // both copies (the NHWC buffer is what the transition reads).
  unsigned voff[2];          // scr: byte offset of the lane's 16 B inside a k-step plane
  unsigned noff[2];          // NHWC: byte offset of the lane's pixel
  unsigned soff[2], sscr[2]; // store offsets of the lane's 32 output bytes: NHWC (without the layer's 2 K) / scr (k-step h of the pair); invalid slots: out of range
  bool valid[2];
  const int col = slot16 < 1 ? 0 : (slot16 > 14 ? 13 : slot16 - 1);
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int row = r0 + f + 2 * st;
    const int rowc = row > 13 ? 13 : row;
    valid[f] = row < 14 && slot16 >= 1 && slot16 <= 14;
    const unsigned px = (unsigned)(rowc * 14 + col);
    noff[f] = px * ldc * 2;
    voff[f] = px * 32 + 16 * h;
    soff[f] = valid[f] ? noff[f] + 32 * h : 0x80000000u;
    sscr[f] = valid[f] ? px * 32 + (unsigned)h * kPlaneB : 0x80000000u;
  }
  unsigned char *scr = (unsigned char *)a.scratch + (size_t)blockIdx.x * kFrameScrB;
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(scr, 0, (int)kFrameScrB, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(fb, 0, (int)(196u * ldc * 2), 0x00020000);
  // the pixel fragment of the shift k-step: (1, 1, mask, 0 ...) in the lanes that hold k = 0 .. 7; mask = -60000 where the slot is padding
  // (a padding column, or one of wave 3's rows 14 / 15: the same lanes in both fragments)
  const u32x4 xb_shift = u32x4{h == 0 ? 0x3c003c00u : 0u, (h == 0 && !valid[0]) ? 0x0000fb53u : 0u, 0u, 0u};
  // bottleneck tile: cell of (row R, slot, tuple T, half h) at R * 4096 + slot * 256 + ((2 T + h) ^ slot) * 16 = lt0 ^ 32 T for the
  // lane's strip of fragment X with dy = -1 (row R = r0 + 2 st): fragment F, kernel row dy add (F + dy + 1) * 4096
  const unsigned lt0 = (unsigned)(r0 + 2 * st) * kTileRowB + slot16 * 256 + (((unsigned)(slot16 ^ h)) << 4);
  const unsigned lane16 = lane * 16, lane4 = lane * 4;

  // ================= the weight stream =================
  // unit g lives in ring slot g % kNR; interval g (= the code that consumes unit g) issues the DMA of unit g + 4 (the stream ends
  // with four units of padding, so the last intervals need no special case).  Wave w copies bytes [4096 w, 4096 w + 4096) of a
  // unit's fragments (two statements of two 1-KiB pieces) and 256 B of its constants (waves 2 / 3 repeat the piece of waves 0 / 1:
  // every wave issues the same number of loads in every interval, which is what the vmcnt constants count).
  const unsigned ring0 = lds0 + kTileBytes, ring_end = ring0 + kNR * kUnitBytes;
  const unsigned char *dsrc = a.stream + wid * 4096;            // this wave's part of the unit to copy next (unit g + 4)
  unsigned ddst = ring0 + wid * 4096;                           // ... and where it goes
  const int cdelta = kUnitFrag - wid * 4096 + (wid & 1) * 256;  // from there to this wave's piece of the constants
  unsigned nxt = kTileBytes;                                    // byte offset in smem of the slot of unit g + 1
  unsigned vb_cur = 0, vb_next = 0, vc_cur = 0, vc_next = 0;    // LDS byte offsets: fragments (+ lane * 16) / constants (+ h * 64) of unit g, g + 1
  auto dma_pair = [&](auto pr_tag) TN_INL {
    if (TN_B14_EXP & 8) return;
    dma16x2<decltype(pr_tag)::value * 2048>(dsrc, lane16, ddst);
  };
  auto dma_consts = [&]() TN_INL {
    if (TN_B14_EXP & 8) return;
    dma4(dsrc + cdelta, lane4, ddst + cdelta);
  };
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
    nxt = nxt + kUnitBytes >= (unsigned)kLdsBytes ? (unsigned)kTileBytes : nxt + kUnitBytes;
  };
  auto end_interval = [&]() TN_INL { advance_dma(); };

  // ================= state that lives across slots =================
  // (the activation ring [super-step parity][k-step][fragment], 8 channels of the lane's pixel each: v[192:255], ring_reg())
  u32x4 xb[2][2];        // BN1 + ReLU'd pixel fragments [k-step parity][fragment]: produced one k-step ahead of the MFMAs
  u32x4 wa[4];           // 1x1 weight fragments [32-channel block]: reloaded for the next k-step behind the block's second MFMA
  u32x4 wsh[2];          // ... of the shift k-step [block parity]
  u32x4 cb[2];           // BN1 constants of dword J of the k-step in production [J parity]: .x = (a0, a1), .y = (b0, b1) as packed halves (bn_relu_fold_fp16); reloaded for dword J + 2 behind the dword's second item
  f32x16 acc[4][2];      // 1x1 accumulators [block][fragment]
  u32x4 fwd[2][2];       // the newest 32 channels, raw fp16 [fragment][k-step]: channels K - 32 + 16 h + 8 k .. + 7 of the lane's pixel
  u32x4 w3f[2][3];       // 3x3 weight fragments [step parity][dx]
  u32x4 bop[2][2];       // 3x3 pixel fragments [step parity][fragment]
  f32x16 bacc[3][2];     // 3x3 accumulators [dx][fragment]

  // ---- ring: asm loads into literal registers + counted waits ----
  const unsigned char *rb_a = scr, *rb_b = scr;   // scr + 4 planes * (super-step index) of the two refill targets of an interval
  auto ring_load = [&](auto rs_tag, auto k_tag, auto f_tag, const unsigned char *base) TN_INL {
    constexpr int KQ = decltype(k_tag)::value, F = decltype(f_tag)::value, R = ring_reg(decltype(rs_tag)::value, KQ, F);
    if (TN_B14_EXP & 1) return;
    const unsigned vo = (TN_B14_EXP & 16) ? lane16 + (unsigned)(wid * 8 + F * 4 + KQ) * 1024u : voff[F];
    const unsigned char *pb = base + KQ * kPlaneB;
    asm volatile("global_load_dwordx4 v[%c0:%c1], %2, %3" ::"n"(R), "n"(R + 3), "v"(vo), "s"(pb) : TN_RING_CLOBBER);
  };
  auto ring_wait = [&](auto rs_tag, auto k_tag) TN_INL { asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(kVmRing) : TN_RING_CLOBBER); };
  // BN1 + ReLU of one dword (two channels) of a pixel fragment: relu(a x + b) with fp16 constants, fused multiply-add (one rounding), packed max
  auto bn_ring = [&](auto reg_tag, auto j_tag) TN_INL -> unsigned {      // input: ring register REG
    constexpr int J = decltype(j_tag)::value, REG = decltype(reg_tag)::value;
    const u32x4 c = cb[J & 1];
    unsigned o;
    asm volatile("v_pk_max_f16 %0, v%c3, %1\n\tv_pk_min_f16 %0, %0, %2" : "=&v"(o) : "v"(c.x), "v"(c.y), "n"(REG) : TN_RING_CLOBBER);
    return o;
  };
  auto bn_dword = [&](const unsigned in, auto j_tag) TN_INL -> unsigned {
    constexpr int J = decltype(j_tag)::value;
    const u32x4 c = cb[J & 1];
    unsigned o;
    asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3" : "=&v"(o) : "v"(in), "v"(c.x), "v"(c.y));
    return o;
  };
  // constants of dword J of k-step KQ of the unit at vc: halves (a[2J], a[2J+1]), (b[2J], b[2J+1]), 8 B unused - 16 B per lane
  auto consts_read = [&](const unsigned vc, auto kq_tag, auto j_tag) TN_INL {
    constexpr int KQ = decltype(kq_tag)::value, J = decltype(j_tag)::value;
    cb[J & 1] = *(const u32x4 *)(smem + vc + 128 * KQ + 16 * J);
  };
  auto wa_read = [&](auto mb_tag, const unsigned vb, auto kq_tag) TN_INL {
    constexpr int MB = decltype(mb_tag)::value, KQ = decltype(kq_tag)::value;
    wa[MB] = *(const u32x4 *)(smem + vb + (KQ * 4 + MB) * 1024);
  };

  // ================= a super-step interval =================
  // consumes unit g = super-step u of the layer (ring slot RS): 4 k-steps x 4 blocks x 2 fragments = 32 MFMA slots.
  // Slot (q, mb, f), e = 2 mb + f, carries: the reload of the block's weight register for k-step q + 1 (behind f = 1); ONE item
  // of the BN1 pipeline, which runs one k-step ahead: item e = dword e >> 1 of fragment e & 1 of k-step q + 1 (k-steps 1 .. 3 of
  // this super-step; during q = 3 k-step 0 of the NEXT unit: the other ring slot, or the forwarded registers when the next unit
  // is the layer's tail); on even e the constants of the dword after next; behind the last item that reads a ring register its
  // refill (two super-steps ahead); behind the last slot of k-steps 0, 1, 3 one DMA statement.
  //   KIND 0: first super-step of a layer (accumulators start from C = 0), 1: inner, 2: last (the next unit is the tail)
  auto su_interval = [&](auto rs_tag, auto kind_tag, auto vm_tag) TN_INL {
    constexpr int RS = decltype(rs_tag)::value, KIND = decltype(kind_tag)::value;
    constexpr bool FIRST = KIND == 0, LAST = KIND == 2;
    begin_interval(vm_tag);
    static_for<32>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value, Q = I >> 3, MB = (I >> 1) & 3, F = I & 1, E = I & 7;
      if constexpr (FIRST && Q == 0) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        acc[MB][F] = mfma32(wa[MB], xb[0][F], z);
      } else {
        acc[MB][F] = mfma32(wa[MB], xb[Q & 1][F], acc[MB][F]);
      }
      // the block's weight fragment of k-step q + 1
      if constexpr (F == 1) {
        if constexpr (Q < 3) wa_read(ic<MB>{}, vb_cur, ic<Q + 1>{});
        else wa_read(ic<MB>{}, vb_next, ic<0>{});
      }
      // BN1 item e of k-step q + 1
      {
        constexpr int J = E >> 1, BF = E & 1;
        if constexpr (Q < 3) {
          if constexpr (E == 0) ring_wait(ic<RS>{}, ic<Q + 1>{});
          if (!(TN_B14_EXP & 4)) xb[(Q + 1) & 1][BF][J] = bn_ring(ic<ring_reg(RS, Q + 1, BF) + J>{}, ic<J>{});
          if constexpr (J == 3) ring_load(ic<RS>{}, ic<Q + 1>{}, ic<BF>{}, rb_a);
        } else if constexpr (!LAST) {
          if constexpr (E == 0) ring_wait(ic<RS ^ 1>{}, ic<0>{});
          if (!(TN_B14_EXP & 4)) xb[0][BF][J] = bn_ring(ic<ring_reg(RS ^ 1, 0, BF) + J>{}, ic<J>{});
          if constexpr (J == 3) ring_load(ic<RS ^ 1>{}, ic<0>{}, ic<BF>{}, rb_b);
        } else {
          if (!(TN_B14_EXP & 4)) xb[0][BF][J] = bn_dword(fwd[BF][0][J], ic<J>{});
        }
        // behind a dword's second item its constant register is free: the constants two dwords on (three slots ahead of their
        // first use) - of this k-step, or dwords 0 / 1 of the k-step produced next
        if constexpr (BF == 1) {
          if constexpr (J < 2) {
            if constexpr (Q < 3) consts_read(vc_cur, ic<Q + 1>{}, ic<J + 2>{});
            else consts_read(vc_next, ic<0>{}, ic<J + 2>{});
          } else {
            if constexpr (Q < 2) consts_read(vc_cur, ic<Q + 2>{}, ic<J - 2>{});
            else consts_read(vc_next, ic<Q - 2>{}, ic<J - 2>{});
          }
        }
      }
      if constexpr (LAST && Q == 3) TN_RING_FENCE();      // (no ring statement in these slots)
      if constexpr (E == 7) {
        if constexpr (Q == 0) dma_pair(ic<0>{});
        else if constexpr (Q == 1) dma_pair(ic<1>{});
        else if constexpr (Q == 3) dma_consts();
      }
      TN_SB();
    });
    end_interval();
  };

  // ================= the tail interval: forwarded k-steps A / B and the shift k-step =================
  // k-step A in block order (its 8 slots carry BN1 of k-step B); then per block the MFMAs of k-step B and of the shift k-step:
  // after them the block's accumulators are final, and epilogue A of block mb - 1 (convert + ReLU + the tile write of tuples
  // 2 mb, 2 mb + 1) runs under block mb's four MFMAs; block 3's epilogue runs under the first 3x3 slots.
  unsigned e_pk[4];
  auto epa_item = [&](auto mb_tag, auto e_tag) TN_INL {       // 20 items per block: (fragment, tuple) x (4 converts + the write)
    constexpr int MB = decltype(mb_tag)::value, E = decltype(e_tag)::value;
    constexpr int F = E / 10, T2 = (E / 5) & 1, I = E % 5, T = 2 * MB + T2, R0 = 8 * T2;
    unsigned (&epk)[4] = e_pk;
    f32x16 (&accr)[4][2] = acc;
    if constexpr (I < 4) {
      const float a0 = accr[MB][F][R0 + 2 * I], a1 = accr[MB][F][R0 + 2 * I + 1];
      asm("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=v"(epk[I]) : "v"(a0), "v"(a1));
    } else {
      *(u32x4 *)(smem + (lt0 ^ (32u * T)) + (F + 1) * kTileRowB) = u32x4{epk[0], epk[1], epk[2], epk[3]};
    }
  };
  auto w3_read = [&](auto s_tag, auto dx_tag, const unsigned vb) TN_INL {      // fragment (step S, dx) = fragment 3 (S & 3) + dx of its unit
    constexpr int S = decltype(s_tag)::value, DX = decltype(dx_tag)::value;
    w3f[S & 1][DX] = *(const u32x4 *)(smem + vb + ((S & 3) * 3 + DX) * 1024);
  };
  auto bop_read = [&](auto s_tag, auto f_tag) TN_INL {       // pixel fragment of step S = (dy index, tuple): kernel rows in the order 0, -1, +1
    constexpr int S = decltype(s_tag)::value, F = decltype(f_tag)::value, DYI = S / 8, T = S % 8;
    constexpr int DY = DYI == 0 ? 0 : (DYI == 1 ? -1 : 1);
    bop[S & 1][F] = *(const u32x4 *)(smem + (lt0 ^ (32u * T)) + (F + DY + 1) * kTileRowB);
  };
  // epilogue A item k (0 .. 19) of block MB, `n` of them from K0 on
  auto epa_items = [&](auto mb_tag, auto k0_tag, auto n_tag) TN_INL {
    static_for<decltype(n_tag)::value>([&](auto k_tag) TN_INL { epa_item(mb_tag, ic<decltype(k0_tag)::value + decltype(k_tag)::value>{}); });
  };
  // 24 slots: k-step A (its slots carry BN1 of k-step B), k-step B, the shift k-step - each in block order, so that an
  // accumulator is touched every 8th slot (back-to-back MFMAs on one accumulator wait for each other: the first version ran
  // k-step B and the shift k-step of a block right behind each other, 100 cycles per slot).  Block 0 is final after slot 17: its
  // epilogue A (convert + ReLU + the tile write of tuples 0, 1) runs under slots 18 - 22; the other blocks' under the first 3x3 slots.
  auto tail_interval = [&]() TN_INL {
    begin_interval(ic<kVmDmaTail>{});
    static_for<24>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value, KS = I >> 3, MB = (I >> 1) & 3, F = I & 1;
      if constexpr (KS == 0) {
        constexpr int J = (I & 7) >> 1, BF = I & 1;
        acc[MB][F] = mfma32(wa[MB], xb[0][F], acc[MB][F]);
        if constexpr (F == 1) wa_read(ic<MB>{}, vb_cur, ic<1>{});
        if (!(TN_B14_EXP & 4)) xb[1][BF][J] = bn_dword(fwd[BF][1][J], ic<J>{});
        if constexpr (BF == 1 && J < 2) consts_read(vc_cur, ic<1>{}, ic<J + 2>{});
      } else if constexpr (KS == 1) {
        acc[MB][F] = mfma32(wa[MB], xb[1][F], acc[MB][F]);
      } else {
        acc[MB][F] = mfma32(wsh[MB & 1], xb_shift, acc[MB][F]);
      }
      // the shift k-step's fragments (k-step 2 of the unit), two registers: blocks 0 / 1 early, 2 / 3 behind the last use of 0 / 1
      if constexpr (I == 5) wsh[0] = *(const u32x4 *)(smem + vb_cur + (2 * 4 + 0) * 1024);
      if constexpr (I == 9) wsh[1] = *(const u32x4 *)(smem + vb_cur + (2 * 4 + 1) * 1024);
      if constexpr (I == 17) wsh[0] = *(const u32x4 *)(smem + vb_cur + (2 * 4 + 2) * 1024);
      if constexpr (I == 19) wsh[1] = *(const u32x4 *)(smem + vb_cur + (2 * 4 + 3) * 1024);
      if constexpr (I >= 18 && I < 23) epa_items(ic<0>{}, ic<(I - 18) * 4>{}, ic<4>{});
      if constexpr (I == 7) dma_pair(ic<0>{});
      if constexpr (I == 11) dma_pair(ic<1>{});
      if constexpr (I == 15) dma_consts();
      // the 3x3's first operands: weight fragments of steps 0 / 1 (unit g + 1), pixel fragments of step 0 (tuple 0: block 0's epilogue is done)
      if constexpr (I >= 12 && I < 18) w3_read(ic<(I - 12) / 3>{}, ic<(I - 12) % 3>{}, vb_next);
      if constexpr (I == 23) { bop_read(ic<0>{}, ic<0>{}); bop_read(ic<0>{}, ic<1>{}); }
      TN_RING_FENCE();
      TN_SB();
    });
    end_interval();
  };

  // ================= the 3x3 intervals: J = 0 .. 5, steps 4 J .. 4 J + 3, slot (step, dx, f) =================
  // J = 0 / 1 carry epilogue A of blocks 1 - 3 (each ahead of the first read of its tuples); J = 5 carries the head of the NEXT
  // layer's BN1 pipeline (pre_item).  The barrier of J = 2 publishes the tile: kernel row -1 (steps 8 ..) is the first that needs
  // the neighbours' rows; step 8 reads fragment X's operand behind that barrier and runs fragment Y first.
  auto pre_item = [&](auto pn_tag, auto i_tag) TN_INL {      // the next layer's k-step 0 (ring slot PN): constants, 8 BN items, weights
    constexpr int PN = decltype(pn_tag)::value, I = decltype(i_tag)::value;
    // items 0, 1: constants of k-step 0, dwords 0 / 1; 2 - 9: BN items (dword, fragment), behind a dword's second item the constants
    // two dwords on (dwords 0 / 1 of k-step 1 at the end) and behind dword 3 the ring refills; 10 - 13: weight fragments of k-step 0
    if constexpr (I < 2) {
      consts_read(vc_next, ic<0>{}, ic<I>{});
    } else if constexpr (I < 10) {
      constexpr int E = I - 2, J = E >> 1, BF = E & 1;
      if constexpr (E == 0) ring_wait(ic<PN>{}, ic<0>{});
      if (!(TN_B14_EXP & 4)) xb[0][BF][J] = bn_ring(ic<ring_reg(PN, 0, BF) + J>{}, ic<J>{});
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
      if (!(TN_B14_EXP & 2)) {
        if constexpr (S == 0) {
          f32x16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          bacc[DX][F] = mfma32(w3f[0][DX], bop[0][F], z);
        } else {
          bacc[DX][F] = mfma32(w3f[S & 1][DX], bop[S & 1][F], bacc[DX][F]);
        }
      }
      if constexpr (SECOND_USE && S + 2 < 24) {      // weight fragments of step S + 2: this unit or the next
        if constexpr (((S + 2) >> 2) == J) w3_read(ic<S + 2>{}, ic<DX>{}, vb_cur);
        else w3_read(ic<S + 2>{}, ic<DX>{}, vb_next);
      }
      // pixel fragments of step S + 1 (its parity's last MFMAs were step S - 1's), three slots ahead; step 8's fragment X
      // (the first operand with a neighbour's row) behind the barrier that publishes the tile, at the top of the step
      if constexpr (S + 1 < 24 && (P == 2 || P == 3) && !(S + 1 == 8 && P == 2)) bop_read(ic<S + 1>{}, ic<P - 2>{});
      if constexpr (S == 8 && P == 0) bop_read(ic<8>{}, ic<0>{});
      // epilogue A of blocks 1, 2, 3: 20 items each, ahead of the first read of their tuples
      if constexpr (J == 0 && I < 8) epa_items(ic<1>{}, ic<I * 5 / 2>{}, ic<(I + 1) * 5 / 2 - I * 5 / 2>{});
      if constexpr (J == 0 && I >= 8 && I < 20) epa_items(ic<2>{}, ic<(I - 8) * 5 / 3>{}, ic<(I - 7) * 5 / 3 - (I - 8) * 5 / 3>{});
      if constexpr (J == 0 && I >= 20) epa_items(ic<3>{}, ic<(I - 20) * 5 / 3>{}, ic<(I - 19) * 5 / 3 - (I - 20) * 5 / 3>{});
      if constexpr (J == 1 && I < 8) epa_items(ic<3>{}, ic<(I + 4) * 5 / 3>{}, ic<(I + 5) * 5 / 3 - (I + 4) * 5 / 3>{});
      if constexpr (J == 5 && I < kPreItems) pre_item(pn_tag, ic<I>{});     // (ahead of the interval's DMA statements: kVmRing counts on it)
      if constexpr (I == 15) dma_pair(ic<0>{});
      if constexpr (I == 19) dma_pair(ic<1>{});
      if constexpr (I == 23) dma_consts();
      TN_RING_FENCE();
      TN_SB();
    });
    end_interval();
  };

  // ---- epilogue B: out[x] = acc[dx = 1][x] + acc[dx = 0][x - 1] + acc[dx = 2][x + 1] (two DPP row shifts inside the 16-lane
  // strip), fp16: the lane's 16 output channels 16 h .. 16 h + 15 = the forwarded registers of the next layer, and 32 B to HBM ----
  auto epilogue_b = [&](int K) TN_INL {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      unsigned o[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float c0 = bacc[1][f][2 * p], c1 = bacc[1][f][2 * p + 1];
        c0 += dpp_f32<0x111>(bacc[0][f][2 * p]);          // row_shr:1: lane x reads lane x - 1
        c1 += dpp_f32<0x111>(bacc[0][f][2 * p + 1]);
        c0 += dpp_f32<0x101>(bacc[2][f][2 * p]);          // row_shl:1: lane x reads lane x + 1
        c1 += dpp_f32<0x101>(bacc[2][f][2 * p + 1]);
        const h2_t pk = {(f16)c0, (f16)c1};
        o[p] = __builtin_bit_cast(unsigned, pk);
        TN_RING_FENCE();      // (the ring holds the next layer's first super-steps: no temporaries in its registers)
      }
      fwd[f][0] = u32x4{o[0], o[1], o[2], o[3]};
      fwd[f][1] = u32x4{o[4], o[5], o[6], o[7]};
      __builtin_amdgcn_raw_buffer_store_b128(fwd[f][0], orsrc, soff[f], 2 * K, 0);
      __builtin_amdgcn_raw_buffer_store_b128(fwd[f][1], orsrc, soff[f] + 16, 2 * K, 0);
      __builtin_amdgcn_raw_buffer_store_b128(fwd[f][0], srsrc, sscr[f], (K >> 4) * kPlaneB, 0);
      __builtin_amdgcn_raw_buffer_store_b128(fwd[f][1], srsrc, sscr[f] + 16, (K >> 4) * kPlaneB, 0);
    }
  };

  // ================= prologue =================
  {
    // tile rows 0, 15, 16, 17: zeros
    for (int i = tid; i < kTileRowB / 16; i += 256) {
      *(u32x4 *)(smem + i * 16) = u32x4{0, 0, 0, 0};
      *(u32x4 *)(smem + 15 * kTileRowB + i * 16) = u32x4{0, 0, 0, 0};
      *(u32x4 *)(smem + 16 * kTileRowB + i * 16) = u32x4{0, 0, 0, 0};
      *(u32x4 *)(smem + 17 * kTileRowB + i * 16) = u32x4{0, 0, 0, 0};
    }
    for (int u = 0; u < 4; ++u) {      // units 0 .. 3 (unit 4 belongs to interval 0)
      dma_pair(ic<0>{});
      dma_pair(ic<1>{});
      dma_consts();
      advance_dma();
    }
    // channels 0 .. K0 - 1 of this wave's pixels: NHWC -> the k-step-major copy (a wave only ever reads its own pixels' planes)
    // (eight k-steps' loads in flight per round trip: one load - wait - store per k-step was 16 dependent round trips, 20 us of a
    // 360 us launch)
    const int nq = a.K0 / 16;
    int q0 = 0;
    for (; q0 + 8 <= nq; q0 += 8) {
      u32x4 v[8][2];
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int f = 0; f < 2; ++f) v[q][f] = *(const u32x4 *)(fb + noff[f] + 32 * (q0 + q) + 16 * h);
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          if (valid[f]) *(u32x4 *)(scr + (size_t)(q0 + q) * kPlaneB + voff[f]) = v[q][f];
    }
    for (int q = q0; q < nq; ++q)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const u32x4 v = *(const u32x4 *)(fb + noff[f] + 32 * q + 16 * h);
        if (valid[f]) *(u32x4 *)(scr + (size_t)q * kPlaneB + voff[f]) = v;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the first layer's ring: super-steps 0 (slot 0) and 1 (slot 1); its "forwarded" channels K0 - 32 .. K0 - 1 from memory
    static_for<16>([&](auto i_tag) TN_INL {
      constexpr int I = decltype(i_tag)::value;
      ring_load(ic<(I >> 3)>{}, ic<((I >> 1) & 3)>{}, ic<(I & 1)>{}, scr + 4 * kPlaneB * (I >> 3));
    });
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const unsigned char *p = fb + noff[f] + 2 * (a.K0 - 32) + 32 * h;
      fwd[f][0] = *(const u32x4 *)p;
      fwd[f][1] = *(const u32x4 *)(p + 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    vb_next = kTileBytes + lane16;                 // the head of the pipeline reads unit 0 as "the next unit" of an interval that does not exist
    vc_next = kTileBytes + kUnitFrag + 64 * h;
    nxt = kTileBytes + kUnitBytes;                 // ... and interval 0 will find unit 1 there
    rb_a = scr + 4 * kPlaneB * 2;                // its refill: super-step 2, k-step 0
    static_for<kPreItems>([&](auto i_tag) TN_INL { pre_item(ic<0>{}, i_tag); });
    // (its two refills have no DMA statements behind them as they have in a 3x3 interval: kVmRing would count one load short
    // at their consumer in the layer's second super-step - tests/test_cpu_block14.py found it; they are simply waited for here)
    asm volatile("s_waitcnt vmcnt(0)" ::: TN_RING_CLOBBER);
    TN_SB();
  }

  // ================= the block =================
  int par = 0;                                   // ring slot of the layer's super-step 0
  for (int l = 0; l < a.nl; ++l) {
    const int K = a.K0 + 32 * l;
    const int nsu = (K - 32 + 63) >> 6;          // super-steps from memory (channels 0 .. K - 33, rounded up; the pad has zero weights)
    const int nsu_next = (K + 63) >> 6;
    // refill targets of super-step interval u: k-steps 1 .. 3 of super-step u + 2, k-step 0 of super-step u + 3 (past this
    // layer's last one: the next layer's, same addresses - a pixel's channels do not move)
    auto refill_bases = [&](int u) TN_INL {
      const int ua = u + 2 < nsu ? u + 2 : u + 2 - nsu, ub = u + 3 < nsu ? u + 3 : u + 3 - nsu;
      rb_a = scr + 4 * kPlaneB * ua;
      rb_b = scr + 4 * kPlaneB * ub;
    };
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
    if (TN_B14_STAMPS && a.ts && tid == 0) a.ts[(size_t)gridDim.x * 64 + (size_t)blockIdx.x * 96 + 3 * l] = __builtin_amdgcn_s_memtime();
    tail_interval();
    if (TN_B14_STAMPS && a.ts && tid == 0) a.ts[(size_t)gridDim.x * 64 + (size_t)blockIdx.x * 96 + 3 * l + 1] = __builtin_amdgcn_s_memtime();
    b_interval(ic<0>{}, ic<0>{}, ic<kVmDmaB0>{});
    b_interval(ic<1>{}, ic<0>{}, ic<kVmDmaB>{});
    b_interval(ic<2>{}, ic<0>{}, ic<kVmDmaB>{});
    b_interval(ic<3>{}, ic<0>{}, ic<kVmDmaB>{});
    b_interval(ic<4>{}, ic<0>{}, ic<kVmDmaB>{});
    par = (par + nsu) & 1;
    rb_a = scr + 4 * kPlaneB * (2 < nsu_next ? 2 : 0);    // the head of the next layer's pipeline refills its super-step 2, k-step 0
    if (par) b_interval(ic<5>{}, ic<1>{}, ic<kVmDmaB>{});
    else b_interval(ic<5>{}, ic<0>{}, ic<kVmDmaB>{});
    if (TN_B14_STAMPS && a.ts && tid == 0) a.ts[(size_t)gridDim.x * 64 + (size_t)blockIdx.x * 96 + 3 * l + 2] = __builtin_amdgcn_s_memtime();
    epilogue_b(K);
    if (a.ts && tid == 0 && l < 62) a.ts[(size_t)blockIdx.x * 64 + l] = __builtin_amdgcn_s_memtime();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_memtime();
}

}  // namespace

bool dense_block14_supported(int H, int W, int K0, int nl) {
  return H == 14 && W == 14 && K0 % 32 == 0 && K0 >= 256 && nl >= 1 && K0 + 32 * (nl - 1) <= 2048;
}

int dense_block14_units(int K0, int nl) {      // (with the four units of padding the last intervals' DMA reads)
  int n = 4;
  for (int l = 0; l < nl; ++l) n += (K0 + 32 * l - 32 + 63) / 64 + 1 + 6;
  return n;
}

size_t dense_block14_scratch_halfs() { return (size_t)kFrameScrB / 2; }

int launch_dense_block14(const DenseBlock14Args &a, hipStream_t s) {
  TN_REQUIRE(a.buf && a.stream && a.scratch, "dense_block14: null operand");
  TN_REQUIRE(dense_block14_supported(14, 14, a.K0, a.nl) && a.ldc % 64 == 0 && a.K0 + 32 * a.nl <= a.ldc, "dense_block14: unsupported geometry");
  TN_REQUIRE(a.total_units == dense_block14_units(a.K0, a.nl), "dense_block14: stream does not match the block");
  TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_block14_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  hipLaunchKernelGGL(dense_block14_kernel, dim3(a.B), dim3(256), kLdsBytes, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// ---- host-side packing: the block's weight stream ----
// Per layer (K input channels, G = K - 32 of them read from memory): ceil(G / 64) super-step units, the tail unit, six 3x3 units;
// every unit is kUnitBytes = 16 fragments [64 lanes][8 halfs] + 128 floats of BN1 constants.
//   super-step unit u: fragment (q, mb): lane l, j: bottleneck channel 32 mb + (l & 31), input channel c = 64 u + 16 q + 8 (l >> 5)
//     + j (zero weight and zero constants for c >= G: the pad half of the last super-step is the forwarded channels, which
//     the tail takes); constants (q, h), dword J: halves a1[c + 2 J], a1[c + 2 J + 1], b1[c + 2 J], b1[c + 2 J + 1] (+ 8 B unused)
//     for c = 64 u + 16 q + 8 h (s1 / t1 of Block14Layer are those fp16 numbers: bn_relu_fold_fp16)
//   tail unit: k-step 0 / 1: input channel G + 16 (l >> 5) + 8 ks + j (the order in which the previous layer's 3x3 leaves its 32
//     output channels in registers); k-step 2: the shift k-step of dense_strip.hip (fp16 hi + lo of BN2's shift, and 1 for the mask)
//   3x3 unit J: fragments (step 4 J + s, dx), s = 0 .. 3: kernel rows in the order ky = 1, 0, 2, tuple t = step % 8; lane layout as
//     pack_w3_strip (dense_strip.hip)
std::vector<unsigned char> pack_block14(const std::vector<Block14Layer> &layers, int K0) {
  const int nl = (int)layers.size();
  std::vector<unsigned char> out((size_t)dense_block14_units(K0, nl) * kUnitBytes, 0);
  size_t unit = 0;
  auto frag = [&](size_t u, int fi) { return (f16 *)(out.data() + u * kUnitBytes + (size_t)fi * 1024); };
  auto cons = [&](size_t u, int q, int h) { return (f16 *)(out.data() + u * kUnitBytes + kUnitFrag + (q * 2 + h) * 64); };
  // channel j (0 .. 7) of a (k-step, half) group: dword J = j >> 1 holds halves (a[2J], a[2J+1]) | (b[2J], b[2J+1]) | 8 B unused
  auto put_const = [](f16 *d, int j, float a, float b) {
    d[8 * (j >> 1) + (j & 1)] = (f16)a;
    d[8 * (j >> 1) + 2 + (j & 1)] = (f16)b;
  };
  for (int l = 0; l < nl; ++l) {
    const Block14Layer &L = layers[l];
    const int K = K0 + 32 * l, G = K - 32, nsu = (G + 63) / 64;
    for (int u = 0; u < nsu; ++u, ++unit)
      for (int q = 0; q < 4; ++q) {
        for (int mb = 0; mb < 4; ++mb) {
          f16 *d = frag(unit, q * 4 + mb);
          for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
              const int c = 64 * u + 16 * q + 8 * (ln >> 5) + j;
              d[ln * 8 + j] = c < G ? (f16)L.w1f[(size_t)(32 * mb + (ln & 31)) * K + c] : (f16)0.f;
            }
        }
        for (int h = 0; h < 2; ++h) {
          f16 *d = cons(unit, q, h);
          for (int j = 0; j < 8; ++j) {
            const int c = 64 * u + 16 * q + 8 * h + j;
            put_const(d, j, c < G ? L.s1[c] : 0.f, c < G ? L.t1[c] : 0.f);
          }
        }
      }
    {  // tail
      for (int ks = 0; ks < 2; ++ks) {
        for (int mb = 0; mb < 4; ++mb) {
          f16 *d = frag(unit, ks * 4 + mb);
          for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
              const int c = G + 16 * (ln >> 5) + 8 * ks + j;
              d[ln * 8 + j] = (f16)L.w1f[(size_t)(32 * mb + (ln & 31)) * K + c];
            }
        }
        for (int h = 0; h < 2; ++h) {
          f16 *d = cons(unit, ks, h);
          for (int j = 0; j < 8; ++j) {
            const int c = G + 16 * h + 8 * ks + j;
            put_const(d, j, L.s1[c], L.t1[c]);
          }
        }
      }
      for (int mb = 0; mb < 4; ++mb) {
        f16 *d = frag(unit, 2 * 4 + mb);
        for (int ln = 0; ln < 32; ++ln) {
          const float t = L.t2[32 * mb + ln];
          d[ln * 8 + 0] = (f16)t;
          d[ln * 8 + 1] = (f16)(t - (float)d[ln * 8 + 0]);
          d[ln * 8 + 2] = (f16)1.f;
        }
      }
      ++unit;
    }
    for (int J = 0; J < 6; ++J, ++unit)
      for (int s = 0; s < 4; ++s) {
        const int step = 4 * J + s, ky = step / 8 == 0 ? 1 : (step / 8 == 1 ? 0 : 2), t = step % 8;
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
  }
  return out;
}

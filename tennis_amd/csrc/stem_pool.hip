// K1 fused — stem Convolution 7x7/2 pad 3 (3 -> 64) + BatchNorm + ReLU + MaxPool 3x3/2 pad 1,
// the first four operators of gluoncv's DenseNet .features (reference call site
// models/vision/definitions.py:30).  The 112x112x64 conv map never goes
// to HBM: a tile is 4 x 14 POOLED pixels; the 9 x 32 conv pixels under it (one halo row / column
// recomputed) run on v_mfma_f32_16x16x32_f16, BN is applied in fp32, the three conv rows of a
// pooled row are max-ed in registers, the row maxima are parked in LDS and the horizontal 3-max
// goes from there straight into channels [0,64) of dense block 1's concat buffer.  ReLU is
// applied to the maximum (it commutes with max), so positions outside the conv map (MaxPool
// pads with -inf) are represented by 0.
//
// The kernel is bound by instruction issue, not by the matrix pipe or by LDS / HBM (phase
// stamps and knock-out builds): a 16-pass MFMA hides two single-issue VALU
// instructions, everything beyond that adds its four issue cycles.  Hence
//  * a wave owns one 16-column half of the conv tile for ALL nine conv rows and half of the
//    output channels: patch row p feeds the conv rows r with ky = p - 2r in [0,7), so one
//    16-byte operand read serves up to eight MFMAs (23 reads per tile instead of 63);
//  * accumulators live in VGPRs (three waves per SIMD requested, so no AGPR copies), BN +
//    v_fma_f32 by hand, the vertical max is four v_max_f32 per row half (fp32: the one rounding to fp16 comes after the
//    pool, dithered - dither_pack), border masking is a separate instantiation taken by border tiles only;
//  * the patch is staged as aligned 64-byte blocks (the k-slot layout starts with the zero
//    tap, which makes the operand reads 16-byte aligned at that offset);
//  * workgroups are persistent over a contiguous range of tiles: weights and BN constants are
//    loaded once, and the patch of the next tile is requested before the MFMAs of this one.
// Operand layout: one k-step per kernel row ky; the 32 k-slots are 8 x-taps x 4 channels
// (tap 0 and channel 3 carry zero weights: tap t' = kx + 1), so a lane's 8 values are 2 adjacent
// NHWC4 pixels = one aligned ds_read_b128 of the staged input patch.
#include <type_traits>

#include "common.h"

#ifdef TN_STEM_STAMPS
__device__ unsigned long long tn_stem_acc[4096 * 16];
#define ST_ADD(i, v) do { st_sum[i] += (unsigned long long)(v); } while (0)
#define ST_NOW() __builtin_amdgcn_s_memtime()
#else
#define ST_ADD(i, v) do { } while (0)
#define ST_NOW() 0ull
#endif

namespace {

constexpr int PR = 4, PC = 14;            // pooled tile
constexpr int CR = 2 * PR + 1, CC = 32;   // conv tile (rows, cols; 29 of the 32 columns are needed)
constexpr int IR = 2 * CR + 5;            // 23 input rows
constexpr int IPX = 80;                   // patch row: input pixels ix0 - 3 ... ix0 + 76 (slot = pixel - (ix0 - 3))
constexpr int IPITCH = IPX * 8;           // bytes per patch row (NHWC4 fp16)
constexpr int CPX = 272;                  // bytes per pixel of the row-max tile: 64 ch fp32 + 16 pad (bank spread)
constexpr int GPR = IPX / 8;              // 8-pixel groups per patch row (vector path)

// workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for the global loads of the next patch
// (in flight on purpose) and for the pooled-output stores to be acknowledged
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// fp32 max / three-way max without the canonicalising max(x, x) the IEEE builtins put in front of every operand
__device__ __forceinline__ float fmax_raw(float x, float y) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ float fmax3_raw(float x, float y, float z) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}
// The ONE rounding of the stem's output, fp32 -> fp16, is DITHERED by position (round 5): on a flat image region every pixel of
// a channel holds the same value, round-to-nearest makes the same error at every one of them, and no average downstream
// reduces it (scripts/round_study.py: 9e-4 on the features of a constant frame from this rounding alone, 2e-5 dithered; it also
// de-correlates the roundings of the layers behind it).  A 13-bit number keyed on (pooled row, pooled column, channel) - NOT on
// the frame: a frame's features do not depend on its neighbours in the batch - is added below the fp16 mantissa and the sum is
// truncated (v_cvt_pkrtz_f16_f32): stochastic rounding with a deterministic random number, unbiased for every value.  The
// values are post-ReLU (>= 0); it has to happen AFTER the max pool (a max over differently dithered values is biased upwards),
// which is why the row maxima travel through LDS as fp32.
__device__ __forceinline__ unsigned dither_pack(float a, float b, unsigned h, int i) {
  const unsigned ta = __builtin_amdgcn_ubfe(h, 2 * i, 13), tb = __builtin_amdgcn_ubfe(h, 2 * i + 2, 13);
  const float da = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) + ta);
  const float db = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) + tb);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, (h2)__builtin_amdgcn_cvt_pkrtz(da, db));
}

// staged channel triple of one input pixel (address clamped by the caller, always in bounds): x - 255 mean_c in the form
// common.h "the stem's operand" describes - the integer x - q_c for uint8 frames, v * 255 std_c for normalised input
template <int LAY>
__device__ __forceinline__ void load_raw(const StemArgs &a, long pix, long plane, float (&v)[3]) {
  if constexpr (LAY == TN_LAYOUT_NCHW_F32) {
    const float *x = (const float *)a.x + pix;      // pix = b*3*plane + iy*W + ix
    v[0] = x[0] * stem_unscale(0); v[1] = x[plane] * stem_unscale(1); v[2] = x[2 * plane] * stem_unscale(2);
  } else if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
    const f16 *x = (const f16 *)a.x + pix * 3;      // pix = (b*H + iy)*W + ix
    v[0] = (float)x[0] * stem_unscale(0); v[1] = (float)x[1] * stem_unscale(1); v[2] = (float)x[2] * stem_unscale(2);
  } else {
    const uint8_t *x = (const uint8_t *)a.x + pix * 3;
    // ToTensor (/255) then Normalize (mean,std) - reference evaluate.py:96-97 - with 1 / (255 std) in the weights
    v[0] = (float)x[0] - kStemQ[0];
    v[1] = (float)x[1] - kStemQ[1];
    v[2] = (float)x[2] - kStemQ[2];
  }
}
// what an out-of-frame tap is staged as: the normalised zero (packed halves: {c0 c1}, {c2 0})
template <int LAY>
__device__ __forceinline__ unsigned stem_pad_dword(int odd) {
  if constexpr (LAY != TN_LAYOUT_NHWC_U8) return 0u;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 a = {(f16)(float)stem_pad(0), (f16)(float)stem_pad(1)}, b = {(f16)(float)stem_pad(2), (f16)0.f};
  return odd ? __builtin_bit_cast(unsigned, b) : __builtin_bit_cast(unsigned, a);
}

struct Tile { int b, pr0, pc0; };

// VEC (frame width a multiple of 8): the patch is fetched as aligned groups of 8 pixels per thread with
// 16-byte (fp16 / f32) or 8-byte (u8) loads instead of one element per load.
// EXW (exact-weights mode, TN_ENC_EXACT_WEIGHTS): conv0's weights as hi + lo fp16 pairs, every MFMA issued twice (round 5: with the
// stem's weights plainly rounded the mode measured 2.3e-3 on near-white frames - all 147 taps see the same large operand)
template <int LAY, bool VEC, bool EXW = false>
// three waves per SIMD where the staging registers allow it (fp16 / u8 vector path), two otherwise
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((VEC && LAY != TN_LAYOUT_NCHW_F32 && !EXW) ? 3 : 2))) void stem_pool_kernel(
    StemArgs a, f16 *__restrict__ out, int ldy, int Hp, int Wp, int nstrip, int ntc, int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[IR * IPITCH];
  __shared__ __attribute__((aligned(16))) unsigned char vtile[PR * CC * CPX];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int pl = lane & 15, kc = lane >> 4;

  // this workgroup's contiguous range of tiles; tile f -> (frame, strip of 4 pooled rows, column tile)
  const int per = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  int f = (int)blockIdx.x * per + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
  const int fend = f + per + ((int)blockIdx.x < rem ? 1 : 0);
  auto tile_of = [&](int i) {
    const int b = i / (nstrip * ntc), r = i - b * (nstrip * ntc), st = r / ntc;
    return Tile{b, st * PR, (r - st * ntc) * PC};
  };

  // a wave owns one half of the output channels (n-fragments 2*nh, 2*nh+1) and one 16-column half of the conv
  // tile (ch): its 14 weight fragments and BN constants stay in registers for all of its tiles
  const int nh = wid & 1, ch = wid >> 1;
  f16x8 wa[7][2];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) wa[ky][nf] = ((const f16x8 *)a.wp_zf)[(ky * 4 + 2 * nh + nf) * 64 + lane];
  [[maybe_unused]] f16x8 wl[EXW ? 7 : 1][2];
  if constexpr (EXW) {
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) wl[ky][nf] = ((const f16x8 *)a.wp_zf_lo)[(ky * 4 + 2 * nh + nf) * 64 + lane];
  }
  float sc[2][4], sh[2][4], fl[2][4];          // fl: the ReLU's floor, -m_c of the centred output (StemArgs::floor), 0 without it
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[nf][r] = a.scale[(2 * nh + nf) * 16 + kc * 4 + r];
      sh[nf][r] = (LAY == TN_LAYOUT_NHWC_U8 ? a.shift_u8 : a.shift)[(2 * nh + nf) * 16 + kc * 4 + r];
      fl[nf][r] = a.floor ? a.floor[(2 * nh + nf) * 16 + kc * 4 + r] : 0.f;
    }

  // input patch staging, split in two halves so the loads of the next tile fly during the MFMAs of this one:
  // request (clamped addresses, all loads issued back to back) / commit (zero what lies outside the frame, write
  // NHWC4 fp16 to LDS)
  constexpr int NPX = (IR * IPX + 255) / 256;   // 8 pixels per thread (element-wise path)
  constexpr int NQ = LAY == TN_LAYOUT_NCHW_F32 ? 6 : 3;   // 16-B (8-B for u8) loads per group
  float raw[VEC ? 1 : NPX][3];
  uint4 vq[VEC ? NQ : 1];
  const long plane = (long)a.H * a.W;
  const int vg = t % GPR, vpr = t / GPR < IR ? t / GPR : IR - 1;   // vector path: this thread's patch row and group
  auto request = [&](const Tile &tl) {
    const int iy0 = 2 * (2 * tl.pr0 - 1) - 3, gx0 = 2 * (2 * tl.pc0 - 1) - 6;   // patch origin (row, first staged pixel)
    if constexpr (VEC) {
      // gx0 = 4 pc0 - 8 is a multiple of 8 (pc0 is a multiple of 14) and so is the frame width: a group lies
      // entirely inside or entirely outside the frame
      const int iy = iy0 + vpr, gx = gx0 + 8 * vg;
      const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = gx < 0 ? 0 : (gx > a.W - 8 ? a.W - 8 : gx);
      // unconditional (the 26 threads past the patch repeat its last row, and the last tile of a workgroup is requested
      // twice): loads under a branch reach commit() through PHI copies, which the compiler places - with their
      // s_waitcnt - right behind the loads, and the prefetch is gone
      if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
        const uint4 *src = (const uint4 *)((const f16 *)a.x + (((long)tl.b * a.H + cy) * a.W + cx) * 3);
        vq[0] = src[0]; vq[1] = src[1]; vq[2] = src[2];
      } else if constexpr (LAY == TN_LAYOUT_NHWC_U8) {
        const uint2 *src = (const uint2 *)((const uint8_t *)a.x + (((long)tl.b * a.H + cy) * a.W + cx) * 3);
        const uint2 q0 = src[0], q1 = src[1], q2 = src[2];
        vq[0] = make_uint4(q0.x, q0.y, q1.x, q1.y); vq[1] = make_uint4(q2.x, q2.y, 0, 0);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint4 *src = (const uint4 *)((const float *)a.x + ((long)tl.b * 3 + c) * plane + (long)cy * a.W + cx);
          vq[2 * c] = src[0]; vq[2 * c + 1] = src[1];
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (VEC ? 0 : NPX); ++i) {
      const int p = t + 256 * i;
      const int pr = p / IPX, pc = p - pr * IPX;
      const int iy = iy0 + pr, ix = gx0 + pc;
      const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      const long pix = LAY == TN_LAYOUT_NCHW_F32 ? (long)tl.b * 3 * plane + (long)cy * a.W + cx
                                                  : ((long)tl.b * a.H + cy) * a.W + cx;
      load_raw<LAY>(a, pix, plane, raw[i]);
    }
  };
  auto commit = [&](const Tile &tl) {
    const int iy0 = 2 * (2 * tl.pr0 - 1) - 3, gx0 = 2 * (2 * tl.pc0 - 1) - 6;
    if constexpr (VEC) {
      const int iy = iy0 + vpr, gx = gx0 + 8 * vg;
      const bool in = (unsigned)iy < (unsigned)a.H && gx >= 0 && gx <= a.W - 8;
      unsigned o[16];                              // 8 NHWC4 pixels: {c0 c1} {c2 0}
      if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
        // 24 packed halves d[0..12) -> pixel i starts at half 3i: even pixels are dword-aligned, odd ones straddle
        const unsigned d[12] = {vq[0].x, vq[0].y, vq[0].z, vq[0].w, vq[1].x, vq[1].y, vq[1].z, vq[1].w,
                                vq[2].x, vq[2].y, vq[2].z, vq[2].w};
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const int e = 3 * i / 2;                 // dword of half 3i
          o[2 * i] = d[e];
          o[2 * i + 1] = d[e + 1] & 0xffffu;
          o[2 * i + 2] = __builtin_amdgcn_alignbit(d[e + 2], d[e + 1], 16);
          o[2 * i + 3] = d[e + 2] >> 16;
        }
        // v -> v * 255 std_c in fp32, one rounding (the weights carry 1 / (255 std_c))
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const h2 q = __builtin_bit_cast(h2, o[i]);
          const h2 r = {(f16)((float)q[0] * stem_unscale((i & 1) ? 2 : 0)), (i & 1) ? (f16)0.f : (f16)((float)q[1] * stem_unscale(1))};
          o[i] = __builtin_bit_cast(unsigned, r);
        }
      } else if constexpr (LAY == TN_LAYOUT_NHWC_U8) {
        // x - q_c as packed halves without a conversion: byte b -> 0x6400 | b = the fp16 number 1024 + b (exact), minus
        // 1024 + q_c (exact: integers below 2048) - v_perm_b32, v_or_b32, v_pk_add_f16 per dword
        const unsigned w[7] = {vq[0].x, vq[0].y, vq[0].z, vq[0].w, vq[1].x, vq[1].y, 0u};
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 k01 = {(f16)(-1024.f - kStemQ[0]), (f16)(-1024.f - kStemQ[1])}, k2 = {(f16)(-1024.f - kStemQ[2]), (f16)-1024.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int b = 3 * i, d = b >> 2, bo = b & 3;          // bytes b, b + 1, b + 2 of the 24: in dwords d, d + 1
          const unsigned s01 = (unsigned)bo | (0x0cu << 8) | ((unsigned)(bo + 1) << 16) | (0x0cu << 24);
          const unsigned s2 = (unsigned)(bo + 2) | 0x0c0c0c00u;
          const unsigned p01 = __builtin_amdgcn_perm(w[d + 1], w[d], s01) | 0x64006400u;
          const unsigned p2 = __builtin_amdgcn_perm(w[d + 1], w[d], s2) | 0x64006400u;
          o[2 * i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, p01) + k01);
          o[2 * i + 1] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, p2) + k2);
        }
      } else {
        f16 v[8][3];
        {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float fl[8] = {__builtin_bit_cast(float, vq[2 * c].x), __builtin_bit_cast(float, vq[2 * c].y),
                                 __builtin_bit_cast(float, vq[2 * c].z), __builtin_bit_cast(float, vq[2 * c].w),
                                 __builtin_bit_cast(float, vq[2 * c + 1].x), __builtin_bit_cast(float, vq[2 * c + 1].y),
                                 __builtin_bit_cast(float, vq[2 * c + 1].z), __builtin_bit_cast(float, vq[2 * c + 1].w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i][c] = (f16)(fl[i] * stem_unscale(c));
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f16x4 px = {v[i][0], v[i][1], v[i][2], (f16)0.f};
          const uint2 u = __builtin_bit_cast(uint2, px);
          o[2 * i] = u.x; o[2 * i + 1] = u.y;
        }
      }
      // wave-uniform test first: interior tiles (most of them) take no selects
      const bool border = iy0 < 0 || iy0 + IR > a.H || gx0 < 0 || gx0 + IPX > a.W;
      if (border) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = in ? o[i] : stem_pad_dword<LAY>(i & 1);
      }
      if (t < IR * GPR) {
        uint4 *dst = (uint4 *)(patch + vpr * IPITCH + vg * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (VEC ? 0 : NPX); ++i) {
      const int p = t + 256 * i;
      const int pr = p / IPX, pc = p - pr * IPX;
      const int iy = iy0 + pr, ix = gx0 + pc;
      const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float pd[3] = {LAY == TN_LAYOUT_NHWC_U8 ? (float)stem_pad(0) : 0.f, LAY == TN_LAYOUT_NHWC_U8 ? (float)stem_pad(1) : 0.f,
                           LAY == TN_LAYOUT_NHWC_U8 ? (float)stem_pad(2) : 0.f};
      f16x4 v;
      v[0] = (f16)(in ? raw[i][0] : pd[0]);
      v[1] = (f16)(in ? raw[i][1] : pd[1]);
      v[2] = (f16)(in ? raw[i][2] : pd[2]);
      v[3] = (f16)0.f;
      if (p < IR * IPX) *(f16x4 *)(patch + pr * IPITCH + pc * 8) = v;
    }
  };

  // horizontal 3-max + store: item = (pooled row, pooled column, 8-channel group); 448 items, two rounds.  The item
  // geometry does not depend on the tile: LDS offset and output offset (relative to the tile's first pixel) are fixed
  int p_lds[2], p_pr[2], p_pc[2];
  unsigned p_hash[2];                     // the item's share of the dither key (the tile adds its origin)
  long p_out[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = t + 256 * k, pr = id / (PC * 8), r2 = id - pr * (PC * 8), pc = r2 >> 3, cg = r2 & 7;
    p_pr[k] = id < PR * PC * 8 ? pr : 1 << 20;            // out of range: never valid
    p_pc[k] = pc;
    p_lds[k] = (pr * CC + 2 * pc) * CPX + cg * 32;
    p_hash[k] = (unsigned)pr * 0x85EBCA77u + (unsigned)pc * 0x9E3779B1u + (unsigned)cg * 0xC2B2AE3Du;
    p_out[k] = ((long)pr * Wp + pc) * ldy + cg * 8;
  }

  [[maybe_unused]] unsigned long long st_sum[9] = {};
  [[maybe_unused]] const unsigned long long st_begin = ST_NOW();
  if (f >= fend) return;
  Tile cur = tile_of(f);
  request(cur);
  commit(cur);
  __syncthreads();
  ST_ADD(0, ST_NOW() - st_begin);
  for (; f < fend; ++f) {
    [[maybe_unused]] const unsigned long long st0 = ST_NOW();
    const bool more = f + 1 < fend;
    const Tile nxt = tile_of(more ? f + 1 : f);
    request(nxt);                         // (the last tile once more: see request)
    const int cy0 = 2 * cur.pr0 - 1, cx0 = 2 * cur.pc0 - 1;     // conv coordinates of the tile origin

    // conv + BN -> vertical max -> LDS
    auto conv_tile = [&](auto border_tag) {
      constexpr bool BORDER = decltype(border_tag)::value;
      f32x4 acc[CR][2];
#pragma unroll
      for (int r = 0; r < CR; ++r)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[r][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char *prow = patch + (ch * 16 + pl + kc + 1) * 16;
      const int c = ch * 16 + pl;
      const bool cvalid = (unsigned)(cx0 + c) < (unsigned)a.Wo;
      float m[2][4];                                // running maximum of the pooled row in progress, per channel fragment (fp32: the rounding comes after the pool)
      // conv row r (complete after patch row 2r + 6): BN in fp32, max into the pooled rows it belongs to
      // (r = 2 pr + {0,1,2}); an even row closes pooled row r/2 - 1 (ReLU on the maximum, then LDS) and opens row r/2
      // The hazard recogniser does not count wait states in front of inline asm that reads an MFMA result (seen in round 2
      // as garbage in the low halves of the last pooled row, depending on where the scheduler put the row's last MFMA): the
      // statement OPENS with the 12 wait states an 8-pass MFMA result needs and is ONE statement, so nothing can come
      // between the wait and the reads - correctness does not depend on MFMA placement (the tail calls keep the plain C++ form)
      auto finish_half = [&](int r, int nf, auto tail_tag) {
        float b[4];
        if constexpr (decltype(tail_tag)::value) {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = fmaf(acc[r][nf][j], sc[nf][j], sh[nf][j]);
        } else {
          asm("s_nop 11\n\t"
              "v_fma_f32 %0, %4, %8, %12\n\tv_fma_f32 %1, %5, %9, %13\n\tv_fma_f32 %2, %6, %10, %14\n\tv_fma_f32 %3, %7, %11, %15"
              : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
              : "v"(acc[r][nf][0]), "v"(acc[r][nf][1]), "v"(acc[r][nf][2]), "v"(acc[r][nf][3]),
                "v"(sc[nf][0]), "v"(sc[nf][1]), "v"(sc[nf][2]), "v"(sc[nf][3]), "v"(sh[nf][0]), "v"(sh[nf][1]), "v"(sh[nf][2]), "v"(sh[nf][3]));
        }
        if constexpr (BORDER) {
          const bool valid = cvalid && (unsigned)(cy0 + r) < (unsigned)a.Ho;
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = valid ? b[j] : fl[nf][j];
        }
        if (r == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) m[nf][j] = b[j];
          return;
        }
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fmax_raw(m[nf][j], b[j]);
        if (r & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) m[nf][j] = x[j];
          return;
        }
        *(float4 *)(vtile + ((r / 2 - 1) * CC + c) * CPX + nh * 128 + nf * 64 + kc * 16) =
            make_float4(fmax_raw(x[0], fl[nf][0]), fmax_raw(x[1], fl[nf][1]), fmax_raw(x[2], fl[nf][2]), fmax_raw(x[3], fl[nf][3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) m[nf][j] = b[j];
      };
      // operand ring of three: the read of patch row p + 2 is requested before the MFMAs of row p are issued (the
      // compiler would otherwise place every read right in front of its first consumer and wait for it)
      f16x8 xb[3];
      xb[0] = *(const f16x8 *)(prow);
      xb[1] = *(const f16x8 *)(prow + IPITCH);
#pragma unroll
      for (int p = 0; p < IR; ++p) {
        if (p + 2 < IR) xb[(p + 2) % 3] = *(const f16x8 *)(prow + (p + 2) * IPITCH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < CR; ++r) {
          const int ky = p - 2 * r;
          if (ky >= 0 && ky < 7) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
              acc[r][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ky][nf], xb[p % 3], acc[r][nf], 0, 0, 0);
              if constexpr (EXW) acc[r][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ky][nf], xb[p % 3], acc[r][nf], 0, 0, 0);
            }
          }
        }
        // the two channel fragments of a finished row go one and two patch rows later, between the MFMAs in flight
        if (p >= 7 && (p - 7) / 2 < CR) finish_half((p - 7) >> 1, (p - 7) & 1, std::false_type{});
      }
      finish_half(CR - 1, 0, std::true_type{});
      finish_half(CR - 1, 1, std::true_type{});
    };
    // positions outside the conv map occur in tiles on the frame border only (c <= 28 is what the pooling reads)
    const bool border = cy0 < 0 || cy0 + CR > a.Ho || cx0 < 0 || cx0 + 29 > a.Wo;
    if (border) conv_tile(std::true_type{}); else conv_tile(std::false_type{});
    [[maybe_unused]] const unsigned long long st1 = ST_NOW();
    lds_barrier();                        // row maxima complete; nobody reads the patch any more
    [[maybe_unused]] const unsigned long long st2 = ST_NOW();
    if (more) commit(nxt);
    [[maybe_unused]] const unsigned long long st3 = ST_NOW();
    f16 *obase = out + (((long)cur.b * Hp + cur.pr0) * Wp + cur.pc0) * ldy;
    const unsigned tile_hash = (unsigned)cur.pr0 * 0x85EBCA77u + (unsigned)cur.pc0 * 0x9E3779B1u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (cur.pr0 + p_pr[k] < Hp && cur.pc0 + p_pc[k] < Wp) {
        const float4 *q = (const float4 *)(vtile + p_lds[k]);
        const float4 a0 = q[0], a1 = q[1], b0 = q[CPX / 16], b1 = q[CPX / 16 + 1], c0 = q[2 * (CPX / 16)], c1 = q[2 * (CPX / 16) + 1];
        // the dither key of (pooled row, pooled column, channel group): a multiplicative hash, two bits further on per channel
        unsigned h = tile_hash + p_hash[k];
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        uint4 o;
        o.x = dither_pack(fmax3_raw(a0.x, b0.x, c0.x), fmax3_raw(a0.y, b0.y, c0.y), h, 0);
        o.y = dither_pack(fmax3_raw(a0.z, b0.z, c0.z), fmax3_raw(a0.w, b0.w, c0.w), h, 2);
        o.z = dither_pack(fmax3_raw(a1.x, b1.x, c1.x), fmax3_raw(a1.y, b1.y, c1.y), h, 4);
        o.w = dither_pack(fmax3_raw(a1.z, b1.z, c1.z), fmax3_raw(a1.w, b1.w, c1.w), h, 6);
        *(uint4 *)(obase + p_out[k]) = o;
      }
    }
    [[maybe_unused]] const unsigned long long st4 = ST_NOW();
    lds_barrier();                        // pooling done before the next tile overwrites the row maxima
    ST_ADD(1, st1 - st0); ST_ADD(2, st2 - st1); ST_ADD(3, st3 - st2); ST_ADD(4, st4 - st3); ST_ADD(5, ST_NOW() - st4);
    ST_ADD(7, 1);
    cur = nxt;
  }
#ifdef TN_STEM_STAMPS
  if (t == 0) {
    st_sum[6] = ST_NOW() - st_begin;
    for (int i = 0; i < 9; ++i) tn_stem_acc[(blockIdx.x & 4095) * 16 + i] = st_sum[i];
  }
#endif
}

}  // namespace

// conv output (Ho,Wo) is implied by a.Ho/a.Wo; pooled output (Hp,Wp) = ((Ho-1)/2+1, (Wo-1)/2+1)
int launch_stem_pool(const StemArgs &a, f16 *out, int ldy, int Hp, int Wp, hipStream_t s) {
  TN_REQUIRE(a.layout >= 0 && a.layout <= 2, "stem: unknown input layout");
  TN_REQUIRE(ldy % 8 == 0, "stem: output stride must be a multiple of 8");
  TN_REQUIRE(a.wp_zf != nullptr && a.shift_u8 != nullptr, "stem: packed weights / uint8 shift missing");
  static int slots = 0;
  if (!slots) {
    int dev = 0;
    hipDeviceProp_t prop;
    TN_HIP_CHECK(hipGetDevice(&dev));
    TN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    slots = 3 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);   // three workgroups per CU (registers)
    if (getenv("TN_STEM_WGS")) slots = atoi(getenv("TN_STEM_WGS"));               // tuning hook
  }
  const int nstrip = (Hp + PR - 1) / PR, ntc = (Wp + PC - 1) / PC;
  const long nt = (long)a.B * nstrip * ntc;
  TN_REQUIRE(nt > 0 && nt < (1l << 31), "stem: tile count out of range");
  const int ntiles = (int)nt;
  const dim3 grid(ntiles < slots ? ntiles : slots), block(256);   // persistent: a workgroup walks a contiguous range of tiles
  static const bool novec = getenv("TN_STEM_NOVEC") != nullptr;   // tuning hook
  const bool vec = (a.W % 8 == 0) && a.W >= 80 && !novec;
#define TN_STEM(L) do { if (a.wp_zf_lo) { if (vec) hipLaunchKernelGGL((stem_pool_kernel<L, true, true>), grid, block, 0, s, a, out, ldy, Hp, Wp, nstrip, ntc, ntiles); \
                                            else hipLaunchKernelGGL((stem_pool_kernel<L, false, true>), grid, block, 0, s, a, out, ldy, Hp, Wp, nstrip, ntc, ntiles); } \
                        else if (vec) hipLaunchKernelGGL((stem_pool_kernel<L, true>), grid, block, 0, s, a, out, ldy, Hp, Wp, nstrip, ntc, ntiles); \
                        else hipLaunchKernelGGL((stem_pool_kernel<L, false>), grid, block, 0, s, a, out, ldy, Hp, Wp, nstrip, ntc, ntiles); } while (0)
  if (a.layout == TN_LAYOUT_NCHW_F32) TN_STEM(TN_LAYOUT_NCHW_F32);
  else if (a.layout == TN_LAYOUT_NHWC_F16) TN_STEM(TN_LAYOUT_NHWC_F16);
  else TN_STEM(TN_LAYOUT_NHWC_U8);
#undef TN_STEM
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

#ifdef TN_STEM_STAMPS
extern "C" int tn_dbg_stem_stamps(unsigned long long *out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(tn_stem_acc), sizeof(unsigned long long) * 4096 * 16) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[4096 * 16]; if (hipMemcpyToSymbol(HIP_SYMBOL(tn_stem_acc), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

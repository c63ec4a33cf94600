// K1 fused — stem Convolution 7x7/2 pad 3 (3 -> 64) + BatchNorm + ReLU + MaxPool 3x3/2 pad 1,
// the first four operators of gluoncv's DenseNet .features (reference call site
// models/vision/definitions.py:30).  The 112x112x64 conv map never goes
// to HBM: a workgroup owns a 4 x 14 tile of POOLED pixels, computes the 9 x 32 conv pixels
// under it (one halo row/column recomputed) on v_mfma_f32_16x16x32_f16, applies BN+ReLU in
// fp32, parks the fp16 result in LDS and max-pools it from there straight into channels
// [0,64) of dense block 1's concat buffer.  Post-ReLU values are >= 0, so out-of-range conv
// positions (MaxPool pads with -inf) are represented by 0.
// Operand layout as in stem.hip: one k-step per kernel row ky; the 32 k-slots are 8 x-taps x
// 4 channels (tap 7 and channel 3 carry zero weights), so a lane's 8 values are 2 adjacent
// NHWC4 pixels = one aligned ds_read_b128 of the staged input patch.
#include <type_traits>

#include "common.h"

#ifndef TN_STEM_EXP
#define TN_STEM_EXP 0   // timing experiments: bit 0 skip the conv fragments, bit 1 skip the pooling / store loop
#endif

namespace {

constexpr int PR = 4, PC = 14;            // pooled tile
constexpr int CR = 2 * PR + 1, CC = 32;   // conv tile (rows, cols; 29 of the 32 columns are needed)
constexpr int IR = 2 * CR + 5;            // 23 input rows
constexpr int IPITCH = 576;               // bytes per input patch row: 72 px * 8 B
constexpr int CPX = 136;                  // bytes per conv pixel in LDS: 64 ch fp16 + 8 pad (bank spread)
constexpr int NFRAG = CR * 2;             // 16-pixel fragments of the conv tile

// raw channel triple of one input pixel (address clamped by the caller, always in bounds)
template <int LAY>
__device__ __forceinline__ void load_raw(const StemArgs &a, long pix, long plane, float (&v)[3]) {
  if constexpr (LAY == TN_LAYOUT_NCHW_F32) {
    const float *x = (const float *)a.x + pix;      // pix = b*3*plane + iy*W + ix
    v[0] = x[0]; v[1] = x[plane]; v[2] = x[2 * plane];
  } else if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
    const f16 *x = (const f16 *)a.x + pix * 3;      // pix = (b*H + iy)*W + ix
    v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2];
  } else {
    const uint8_t *x = (const uint8_t *)a.x + pix * 3;
    // ToTensor (/255) then Normalize (mean,std) — reference evaluate.py:96-97
    v[0] = ((float)x[0] / 255.0f - 0.485f) / 0.229f;
    v[1] = ((float)x[1] / 255.0f - 0.456f) / 0.224f;
    v[2] = ((float)x[2] / 255.0f - 0.406f) / 0.225f;
  }
}

// VEC (frame width a multiple of 8): the patch is fetched as aligned groups of 8 pixels per thread with
// 16-byte (fp16 / f32) or 8-byte (u8) loads instead of one element per load.
template <int LAY, bool VEC>
__global__ __launch_bounds__(256) void stem_pool_kernel(StemArgs a, f16 *__restrict__ out, int ldy, int Hp, int Wp) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[IR * IPITCH];
  __shared__ __attribute__((aligned(16))) unsigned char ctile[CR * CC * CPX];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int b = blockIdx.z;
  const int pr0 = blockIdx.y * PR;
  const int cy0 = 2 * pr0 - 1;                           // conv row of the tile origin
  const int iy0 = 2 * cy0 - 3;                           // input row of the patch origin
  const int pl = lane & 15, kc = lane >> 4;
  const int ntc = (Wp + PC - 1) / PC;                    // column tiles walked by this workgroup

  // a wave owns one half of the output channels (n-fragments 2*nh, 2*nh+1): its 14 weight fragments and BN
  // constants stay in registers for the whole row strip (all four n-fragments would cost 112 VGPRs and a
  // third of the occupancy); the other half of each pixel fragment belongs to the partner wave
  const int nh = wid & 1;
  f16x8 wa[7][2];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) wa[ky][nf] = ((const f16x8 *)a.wp)[(ky * 4 + 2 * nh + nf) * 64 + lane];
  float sc[2][4], sh[2][4];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[nf][r] = a.scale[(2 * nh + nf) * 16 + kc * 4 + r];
      sh[nf][r] = a.shift[(2 * nh + nf) * 16 + kc * 4 + r];
    }

  // input patch staging, split in two halves so the loads of tile ct+1 fly during the MFMAs of
  // tile ct: request (clamped addresses, all loads issued back to back) / commit (zero what lies
  // outside the frame, write NHWC4 fp16 to LDS)
  constexpr int NPX = (IR * 72 + 255) / 256;    // 7 pixels per thread (element-wise path)
  constexpr int GPR = 10;                        // 8-pixel groups per patch row (vector path): px [ix0-3, ix0+77)
  constexpr int NQ = LAY == TN_LAYOUT_NCHW_F32 ? 6 : 3;   // 16-B (8-B for u8) loads per group
  float raw[VEC ? 1 : NPX][3];
  uint4 vq[VEC ? NQ : 1];
  const long plane = (long)a.H * a.W;
  const int vpr = t / GPR, vg = t - vpr * GPR;   // vector path: this thread's patch row and group
  auto request = [&](int ct) {
    const int ix0 = 2 * (2 * ct * PC - 1) - 3;
    if constexpr (VEC) {
      // ix0 = 56 ct - 5: the groups start at the 8-aligned pixel ix0 - 3 (the frame width is a multiple of 8,
      // so a group lies entirely inside or entirely outside the frame)
      const int iy = iy0 + vpr, gx = ix0 - 3 + 8 * vg;
      const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = gx < 0 ? 0 : (gx > a.W - 8 ? a.W - 8 : gx);
      if (t < IR * GPR) {
        if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
          const uint4 *src = (const uint4 *)((const f16 *)a.x + (((long)b * a.H + cy) * a.W + cx) * 3);
          vq[0] = src[0]; vq[1] = src[1]; vq[2] = src[2];
        } else if constexpr (LAY == TN_LAYOUT_NHWC_U8) {
          const uint2 *src = (const uint2 *)((const uint8_t *)a.x + (((long)b * a.H + cy) * a.W + cx) * 3);
          const uint2 q0 = src[0], q1 = src[1], q2 = src[2];
          vq[0] = make_uint4(q0.x, q0.y, q1.x, q1.y); vq[1] = make_uint4(q2.x, q2.y, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const uint4 *src = (const uint4 *)((const float *)a.x + ((long)b * 3 + c) * plane + (long)cy * a.W + cx);
            vq[2 * c] = src[0]; vq[2 * c + 1] = src[1];
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (VEC ? 0 : NPX); ++i) {
      const int p = t + 256 * i;
      const int pr = p / 72, pc = p - pr * 72;
      const int iy = iy0 + pr, ix = ix0 + pc;
      const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      const long pix = LAY == TN_LAYOUT_NCHW_F32 ? (long)b * 3 * plane + (long)cy * a.W + cx
                                                  : ((long)b * a.H + cy) * a.W + cx;
      load_raw<LAY>(a, pix, plane, raw[i]);
    }
  };
  auto commit = [&](int ct) {
    const int ix0 = 2 * (2 * ct * PC - 1) - 3;
    if constexpr (VEC) {
      const int iy = iy0 + vpr, gx = ix0 - 3 + 8 * vg;
      const bool in = (unsigned)iy < (unsigned)a.H && gx >= 0 && gx <= a.W - 8;
      f16 v[8][3];
      if constexpr (LAY == TN_LAYOUT_NHWC_F16) {
        const f16x8 h0 = __builtin_bit_cast(f16x8, vq[0]), h1 = __builtin_bit_cast(f16x8, vq[1]), h2 = __builtin_bit_cast(f16x8, vq[2]);
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i / 3][i % 3] = i < 8 ? h0[i] : (i < 16 ? h1[i - 8] : h2[i - 16]);
      } else if constexpr (LAY == TN_LAYOUT_NHWC_U8) {
        const unsigned w[6] = {vq[0].x, vq[0].y, vq[0].z, vq[0].w, vq[1].x, vq[1].y};
        const float mean[3] = {0.485f, 0.456f, 0.406f}, sdev[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          const float u = (float)((w[i >> 2] >> ((i & 3) * 8)) & 255u);
          // ToTensor (/255) then Normalize (mean,std) -- reference evaluate.py:96-97 (same arithmetic as load_raw)
          v[i / 3][i % 3] = (f16)((u / 255.0f - mean[i % 3]) / sdev[i % 3]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float f[8] = {__builtin_bit_cast(float, vq[2 * c].x), __builtin_bit_cast(float, vq[2 * c].y),
                              __builtin_bit_cast(float, vq[2 * c].z), __builtin_bit_cast(float, vq[2 * c].w),
                              __builtin_bit_cast(float, vq[2 * c + 1].x), __builtin_bit_cast(float, vq[2 * c + 1].y),
                              __builtin_bit_cast(float, vq[2 * c + 1].z), __builtin_bit_cast(float, vq[2 * c + 1].w)};
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i][c] = (f16)f[i];
        }
      }
      if (t < IR * GPR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = 8 * vg - 3 + i;          // patch pixel index (0 = ix0)
          f16x4 o = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          if (in) { o[0] = v[i][0]; o[1] = v[i][1]; o[2] = v[i][2]; }
          if (q >= 0 && q < 72) *(f16x4 *)(patch + vpr * IPITCH + q * 8) = o;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (VEC ? 0 : NPX); ++i) {
      const int p = t + 256 * i;
      const int pr = p / 72, pc = p - pr * 72;
      const int iy = iy0 + pr, ix = ix0 + pc;
      const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      f16x4 v;
      v[0] = in ? (f16)raw[i][0] : (f16)0.f;
      v[1] = in ? (f16)raw[i][1] : (f16)0.f;
      v[2] = in ? (f16)raw[i][2] : (f16)0.f;
      v[3] = (f16)0.f;
      if (p < IR * 72) *(f16x4 *)(patch + pr * IPITCH + pc * 8) = v;
    }
  };

  request(0);
  commit(0);
  __syncthreads();
  for (int ct = 0; ct < ntc; ++ct) {
    const int pc0 = ct * PC, cx0 = 2 * pc0 - 1;
    if (ct + 1 < ntc) request(ct + 1);
    // conv + BN + ReLU -> LDS; wave pair p = wid>>1 takes the pixel fragments p, p+2, ..., three at a time so that
    // the patch reads of one kernel row hide behind the MFMAs of the previous one
    auto conv_frags = [&](auto nft, int fa, int fb, int fc) {
      constexpr int NFR = decltype(nft)::value;
      const int fr[3] = {fa, fb, fc};
      f32x4 acc[NFR][2];
#pragma unroll
      for (int q = 0; q < NFR; ++q)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[q][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        f16x8 xb[NFR];
#pragma unroll
        for (int q = 0; q < NFR; ++q) {
          const int r = fr[q] >> 1, c = (fr[q] & 1) * 16 + pl;
          xb[q] = *(const f16x8 *)(patch + (2 * r + ky) * IPITCH + (c + kc) * 16);
        }
#pragma unroll
        for (int q = 0; q < NFR; ++q)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf) acc[q][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ky][nf], xb[q], acc[q][nf], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < NFR; ++q) {
        const int r = fr[q] >> 1, c = (fr[q] & 1) * 16 + pl;          // conv row / column inside the tile
        const bool valid = (unsigned)(cy0 + r) < (unsigned)a.Ho && (unsigned)(cx0 + c) < (unsigned)a.Wo;
        unsigned char *dst = ctile + (r * CC + c) * CPX + kc * 8 + nh * 64;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          f16x4 h;
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = (f16)fmaxf(fmaf(acc[q][nf][j], sc[nf][j], sh[nf][j]), 0.f);
          if (!valid) h = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          *(f16x4 *)(dst + nf * 32) = h;
        }
      }
    };
    if (!(TN_STEM_EXP & 1)) {
      static_assert(NFRAG == 18, "9 fragments per wave pair = 3 triples");
      for (int f = wid >> 1; f < NFRAG; f += 6) conv_frags(std::integral_constant<int, 3>{}, f, f + 2, f + 4);
    }
    __syncthreads();                      // conv tile complete; nobody reads the patch any more
    if (ct + 1 < ntc) commit(ct + 1);
    // 3x3/2 max pool out of LDS: one (pooled pixel, 4-channel group) per work item
    for (int id = t; id < ((TN_STEM_EXP & 2) ? 0 : PR * PC * 16); id += 256) {
      const int pp = id >> 4, cg = id & 15;
      const int pr = pp / PC, pc = pp - pr * PC;
      if (pr0 + pr >= Hp || pc0 + pc >= Wp) continue;
      f16x4 o = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};      // packed fp16 max (exact: values are fp16 already)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          o = __builtin_elementwise_max(o, *(const f16x4 *)(ctile + ((2 * pr + dy) * CC + 2 * pc + dx) * CPX + cg * 8));
      *(f16x4 *)(out + (((long)b * Hp + pr0 + pr) * Wp + pc0 + pc) * ldy + cg * 4) = o;
    }
    __syncthreads();                      // pooling done before the next tile overwrites ctile
  }
}

}  // namespace

// conv output (Ho,Wo) is implied by a.Ho/a.Wo; pooled output (Hp,Wp) = ((Ho-1)/2+1, (Wo-1)/2+1)
int launch_stem_pool(const StemArgs &a, f16 *out, int ldy, int Hp, int Wp, hipStream_t s) {
  TN_REQUIRE(a.layout >= 0 && a.layout <= 2, "stem: unknown input layout");
  TN_REQUIRE(ldy % 4 == 0, "stem: output stride must be a multiple of 4");
  const dim3 grid(1, (Hp + PR - 1) / PR, a.B), block(256);   // a workgroup walks one strip of 4 pooled rows
  static const bool novec = getenv("TN_STEM_NOVEC") != nullptr;   // tuning hook
  const bool vec = (a.W % 8 == 0) && a.W >= 80 && !novec;
#define TN_STEM(L) do { if (vec) hipLaunchKernelGGL((stem_pool_kernel<L, true>), grid, block, 0, s, a, out, ldy, Hp, Wp); \
                        else hipLaunchKernelGGL((stem_pool_kernel<L, false>), grid, block, 0, s, a, out, ldy, Hp, Wp); } while (0)
  if (a.layout == TN_LAYOUT_NCHW_F32) TN_STEM(TN_LAYOUT_NCHW_F32);
  else if (a.layout == TN_LAYOUT_NHWC_F16) TN_STEM(TN_LAYOUT_NHWC_F16);
  else TN_STEM(TN_LAYOUT_NHWC_U8);
#undef TN_STEM
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

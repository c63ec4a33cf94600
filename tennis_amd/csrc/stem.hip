// K1 — stem: Convolution 7x7/2 pad 3 (3 -> 64) + BatchNorm + ReLU, the first
// three operators of gluoncv's DenseNet .features (reference call site
// models/vision/definitions.py:30).  Accepts the reference
// layout (NCHW fp32 = ToTensor+Normalize output, evaluate.py:96-97), native
// NHWC fp16, or raw NHWC uint8 with the normalisation fused into the load.
//
// Implicit GEMM on v_mfma_f32_16x16x32_f16: one k-step per kernel row ky, the
// 32 k-slots of a step are 8 x-taps x 4 channels (tap 7 and channel 3 carry
// zero weights), so a lane's 8 operand values are 2 adjacent NHWC4 pixels =
// one aligned ds_read_b128 from the staged input patch.  BN (fp32 scale and
// shift per output channel) and ReLU are applied to the fp32 accumulators in
// the epilogue.  The staged operand is x - 255 mean (uint8 frames: an exact integer),
// the weights carry 1 / (255 std): common.h "the stem's operand".
// A workgroup owns 8 output rows of one frame and walks the row in 16-column
// tiles, keeping all 28 weight fragments in registers.
#include "common.h"

namespace {

constexpr int PITCH = 320;        // bytes per patch row: 40 px * 8 B
constexpr int PROWS = 21, PCOLS = 38;

// one staged operand pixel (common.h "the stem's operand"): x - q_c for uint8 frames (integers, exact), v * 255 std_c for
// normalised input; outside the frame the normalised zero
__device__ __forceinline__ f16x4 load_px(const StemArgs &a, int b, int iy, int ix) {
  const bool u8 = a.layout == TN_LAYOUT_NHWC_U8;
  f16x4 v = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
  if (u8) { v[0] = (f16)(float)stem_pad(0); v[1] = (f16)(float)stem_pad(1); v[2] = (f16)(float)stem_pad(2); }
  if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
    if (a.layout == TN_LAYOUT_NCHW_F32) {
      const float *x = (const float *)a.x;
      const long plane = (long)a.H * a.W;
      const long o = ((long)b * 3) * plane + (long)iy * a.W + ix;
      v[0] = (f16)(x[o] * stem_unscale(0));
      v[1] = (f16)(x[o + plane] * stem_unscale(1));
      v[2] = (f16)(x[o + 2 * plane] * stem_unscale(2));
    } else if (a.layout == TN_LAYOUT_NHWC_F16) {
      const f16 *x = (const f16 *)a.x + (((long)b * a.H + iy) * a.W + ix) * 3;
      v[0] = (f16)((float)x[0] * stem_unscale(0)); v[1] = (f16)((float)x[1] * stem_unscale(1)); v[2] = (f16)((float)x[2] * stem_unscale(2));
    } else {
      const uint8_t *x = (const uint8_t *)a.x + (((long)b * a.H + iy) * a.W + ix) * 3;
      // ToTensor (/255) then Normalize (mean,std) - reference evaluate.py:96-97 - with 1 / (255 std) in the weights
      v[0] = (f16)((float)x[0] - kStemQ[0]);
      v[1] = (f16)((float)x[1] - kStemQ[1]);
      v[2] = (f16)((float)x[2] - kStemQ[2]);
    }
  }
  return v;
}

__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[PROWS * PITCH];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wid = t >> 6;
  const int b = blockIdx.y;
  const int oy0 = blockIdx.x * 8;
  const int pl = lane & 15;   // pixel (column) within the 16-wide tile / weight row
  const int kc = lane >> 4;   // k chunk: x-taps 2*kc, 2*kc+1

  // all weight fragments stay in registers: [ky][nfrag]
  f16x8 wa[7][4];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) wa[ky][nf] = ((const f16x8 *)a.wp)[(ky * 4 + nf) * 64 + lane];
  float sc[4][4], sh[4][4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[nf][r] = a.scale[nf * 16 + kc * 4 + r];
      sh[nf][r] = (a.layout == TN_LAYOUT_NHWC_U8 ? a.shift_u8 : a.shift)[nf * 16 + kc * 4 + r];
    }

  const int ntiles = (a.Wo + 15) / 16;
  for (int ct = 0; ct < ntiles; ++ct) {
    const int ox0 = ct * 16;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    for (int p = t; p < PROWS * PCOLS; p += 256) {
      const int pr = p / PCOLS, pc = p - pr * PCOLS;
      *(f16x4 *)(patch + pr * PITCH + pc * 8) = load_px(a, b, iy0 + pr, ix0 + pc);
    }
    __syncthreads();

    f32x4 acc[2][4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        const int orow = wid * 2 + mf;  // output row within the strip
        const f16x8 xb = *(const f16x8 *)(patch + (2 * orow + ky) * PITCH + (pl + kc) * 16);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ky][nf], xb, acc[mf][nf], 0, 0, 0);
      }
    }
    // epilogue: D[i=n][j=pixel]; lane: pixel = pl, n = nf*16 + kc*4 + r
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int oy = oy0 + wid * 2 + mf, ox = ox0 + pl;
      if (oy < a.Ho && ox < a.Wo) {
        f16 *dst = a.y + (((long)b * a.Ho + oy) * a.Wo + ox) * 64 + kc * 4;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          f16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r)       // (the ReLU's floor is -m_c of the centred output: StemArgs::floor)
            h[r] = (f16)fmaxf(fmaf(acc[mf][nf][r], sc[nf][r], sh[nf][r]), a.floor ? a.floor[nf * 16 + kc * 4 + r] : 0.f);
          *(f16x4 *)(dst + nf * 16) = h;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_stem(const StemArgs &a, hipStream_t s) {
  TN_REQUIRE(a.layout >= 0 && a.layout <= 2, "stem: unknown input layout");
  TN_REQUIRE(a.shift_u8 != nullptr, "stem: the uint8 shift is missing");
  const dim3 grid((a.Ho + 7) / 8, a.B), block(256);
  hipLaunchKernelGGL(stem_kernel, grid, block, 0, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

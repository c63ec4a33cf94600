// Input side of the path: the geometric half of the reference's test transform
//   transforms.Resize(data_shape + 32)   -> mx.image.imresize(w=s, h=s, interp=1)  = cv::resize INTER_LINEAR, 8-bit
//   transforms.CenterCrop(data_shape)    -> crop at int((s - c) / 2)
// (reference evaluate.py:93-96, train.py transform_test) on decoded RGB uint8 frames, one launch per batch.
// ToTensor + Normalize (evaluate.py:96-97) stay fused into the stem's u8 load (stem_pool.hip).
//
// The arithmetic is OpenCV's published fixed-point bilinear for 8-bit images [EXT]: coordinates
// f = (d + 0.5) * (src / dst) - 0.5 in float, 11-bit coefficients rounded half-to-even, horizontal pass in int32,
// vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2x reduction takes the
// 2x2 box average OpenCV substitutes for it.  Only the cropped window is computed.
#include <cmath>
#include <vector>

#include "common.h"

namespace {

struct Tap { int32_t ofs; int16_t c0, c1; };   // source index of the first tap, two coefficients (sum ~2048)

__global__ __launch_bounds__(256) void resize_crop_u8_kernel(const uint8_t *__restrict__ src, int Hs, int Ws,
                                                             const Tap *__restrict__ xt, const Tap *__restrict__ yt,
                                                             int x0, int y0, int crop, int box2,
                                                             uint8_t *__restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= crop) return;
  const uint8_t *img = src + (size_t)b * Hs * Ws * 3;
  uint8_t *o = dst + (((size_t)b * crop + y) * crop + x) * 3;
  if (box2) {   // exact 2x reduction: INTER_AREA fast path
    const uint8_t *p = img + ((size_t)(2 * (y + y0)) * Ws + 2 * (x + x0)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((p[c] + p[3 + c] + p[(size_t)Ws * 3 + c] + p[(size_t)Ws * 3 + 3 + c] + 2) >> 2);
    return;
  }
  const Tap tx = xt[x + x0], ty = yt[y + y0];
  const int r0 = ty.ofs < 0 ? 0 : (ty.ofs < Hs ? ty.ofs : Hs - 1);
  const int r1 = ty.ofs + 1 < 0 ? 0 : (ty.ofs + 1 < Hs ? ty.ofs + 1 : Hs - 1);
  const int c1 = tx.ofs + 1 < Ws ? tx.ofs + 1 : Ws - 1;          // its coefficient is 0 when clamped
  const uint8_t *p0 = img + (size_t)r0 * Ws * 3, *p1 = img + (size_t)r1 * Ws * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = p0[tx.ofs * 3 + c] * tx.c0 + p0[c1 * 3 + c] * tx.c1;
    const int S1 = p1[tx.ofs * 3 + c] * tx.c0 + p1[c1 * 3 + c] * tx.c1;
    o[c] = (uint8_t)((((ty.c0 * (S0 >> 4)) >> 16) + ((ty.c1 * (S1 >> 4)) >> 16) + 2) >> 2);
  }
}

inline int16_t sat_short(float v) {
  const long r = lrintf(v);   // round half to even, as cvRound
  return (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

// cv::resize's coefficient tables for one axis; clamp_ofs: the x axis clamps the offset and zeroes the fraction at
// the borders, the y axis keeps the raw offset (rows are clipped when they are read)
void make_taps(int src, int dst, bool clamp_ofs, std::vector<Tap> &out) {
  out.resize(dst);
  const double scale = (double)src / dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (clamp_ofs) {
      if (s < 0) { f = 0.f; s = 0; }
      if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    out[d] = Tap{s, sat_short((1.f - f) * 2048.f), sat_short(f * 2048.f)};
  }
}

}  // namespace

struct tn_preproc {
  tn_ctx *ctx;
  int Hs, Ws, resize, crop, x0, y0, box2;
  Tap *xt, *yt;
};

extern "C" int tn_preproc_create(tn_ctx *ctx, int src_h, int src_w, int resize, int crop, tn_preproc **out) {
  TN_REQUIRE(ctx && out, "tn_preproc_create: null argument");
  TN_REQUIRE(src_h > 0 && src_w > 0 && resize > 0 && crop > 0, "tn_preproc_create: bad shape");
  TN_REQUIRE(crop <= resize, "tn_preproc_create: crop larger than the resized frame (CenterCrop would rescale: unsupported)");
  TN_ON_DEVICE(ctx->device);
  std::vector<Tap> xt, yt;
  make_taps(src_w, resize, true, xt);
  make_taps(src_h, resize, false, yt);
  tn_preproc *p = new tn_preproc();
  p->ctx = ctx; p->Hs = src_h; p->Ws = src_w; p->resize = resize; p->crop = crop;
  p->x0 = (resize - crop) / 2; p->y0 = (resize - crop) / 2;      // image.center_crop: int((w - new_w) / 2)
  p->box2 = (src_w == 2 * resize && src_h == 2 * resize) ? 1 : 0;
  if (hipMalloc((void **)&p->xt, sizeof(Tap) * resize) != hipSuccess || hipMalloc((void **)&p->yt, sizeof(Tap) * resize) != hipSuccess) {
    if (p->xt) (void)hipFree(p->xt);
    delete p;
    tn_set_error("device allocation failed");
    return TN_ERR_NOMEM;
  }
  TN_HIP_CHECK(hipMemcpy(p->xt, xt.data(), sizeof(Tap) * resize, hipMemcpyHostToDevice));
  TN_HIP_CHECK(hipMemcpy(p->yt, yt.data(), sizeof(Tap) * resize, hipMemcpyHostToDevice));
  *out = p;
  return TN_OK;
}

extern "C" int tn_preproc_forward(tn_preproc *p, const uint8_t *src, int batch, uint8_t *dst) {
  TN_REQUIRE(p && src && dst, "tn_preproc_forward: null argument");
  TN_REQUIRE(batch > 0 && batch <= 65535, "tn_preproc_forward: batch must be in 1..65535");
  TN_ON_DEVICE(p->ctx->device);
  hipLaunchKernelGGL(resize_crop_u8_kernel, dim3((p->crop + 255) / 256, p->crop, batch), dim3(256), 0, p->ctx->stream, src,
                     p->Hs, p->Ws, (const Tap *)p->xt, (const Tap *)p->yt, p->x0, p->y0, p->crop, p->box2, dst);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// ToTensor + Normalize of the transform (evaluate.py:96-97, train.py:138) for consumers that want fp32 frames (the
// fine-tuning step): y = (x / 255 - mean[c]) / std[c], NHWC in, NHWC out.  4 pixels (12 bytes) per thread.
__global__ __launch_bounds__(256) void to_tensor_normalize_kernel(const uint8_t *__restrict__ src, long npix, float m0, float m1, float m2,
                                                                   float i0, float i1, float i2, float *__restrict__ dst) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;     // group of 4 pixels
  const long p0 = q * 4;
  if (p0 >= npix) return;
  const float mean[3] = {m0, m1, m2}, inv[3] = {i0, i1, i2};
  if (p0 + 4 <= npix) {
    const uint32_t *s32 = (const uint32_t *)(src + p0 * 3);
    const uint32_t a = s32[0], b = s32[1], c = s32[2];
    const uint8_t v[12] = {(uint8_t)a, (uint8_t)(a >> 8), (uint8_t)(a >> 16), (uint8_t)(a >> 24), (uint8_t)b, (uint8_t)(b >> 8),
                           (uint8_t)(b >> 16), (uint8_t)(b >> 24), (uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
    float o[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) o[j] = ((float)v[j] / 255.0f - mean[j % 3]) * inv[j % 3];
    float4 *d4 = (float4 *)(dst + p0 * 3);
    d4[0] = make_float4(o[0], o[1], o[2], o[3]);
    d4[1] = make_float4(o[4], o[5], o[6], o[7]);
    d4[2] = make_float4(o[8], o[9], o[10], o[11]);
  } else {
    for (long p = p0; p < npix; ++p)
      for (int c = 0; c < 3; ++c) dst[p * 3 + c] = ((float)src[p * 3 + c] / 255.0f - mean[c]) * inv[c];
  }
}

extern "C" int tn_to_tensor_normalize(tn_ctx *ctx, const uint8_t *src, long pixels, const float *mean3, const float *std3, float *dst) {
  TN_REQUIRE(ctx && src && dst && mean3 && std3, "tn_to_tensor_normalize: null argument");
  TN_REQUIRE(pixels > 0, "tn_to_tensor_normalize: no pixels");
  TN_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "tn_to_tensor_normalize: zero std");
  TN_ON_DEVICE(ctx->device);
  const long groups = (pixels + 3) / 4;
  hipLaunchKernelGGL(to_tensor_normalize_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, ctx->stream, src, pixels, mean3[0],
                     mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], dst);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

extern "C" int tn_preproc_destroy(tn_preproc *p) {
  if (!p) return TN_OK;
  TnDeviceGuard tn_dg_(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  (void)hipFree(p->xt);
  (void)hipFree(p->yt);
  delete p;
  return TN_OK;
}

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

// ---- train-time augmentation (reference train.py:127-136: RandomResizedCrop, RandomFlipLeftRight, RandomColorJitter,
// RandomLighting in front of ToTensor / Normalize) ------------------------------------------------------------------------
// The random draws are the host's (tennis_amd/transforms.py: MXNet's generator cannot be reproduced); what runs here is the
// image arithmetic for a batch of frames with one parameter record each.  All of it stays uint8 between the operators, as
// the Gluon transforms do on the uint8 HWC image ToTensor receives [EXT: mxnet src/operator/image/image_random-inl.h,
// python/mxnet/image/image.py; unpinned]:
//   random_size_crop -> fixed_crop: crop (x0, y0, cw, ch), cv::resize(INTER_LINEAR) to size x size (8-bit fixed point as above,
//     the 2 x 2 box average for an exact 2x reduction); flip: mirror in x;
//   colour jitter: brightness / contrast / saturation (/ hue = 0: nothing) in the record's order, each
//     saturate_cast<uint8>(float): brightness v * a; contrast v * a + (1 - a) * mean over the image of
//     0.299 R + 0.587 G + 0.114 B; saturation v * a + (1 - a) * (0.299 R + 0.587 G + 0.114 B) of the pixel;
//   lighting: v + pca[c] (AlexNet's eigenvectors times the drawn alphas, worked out on the host).
namespace {

__device__ __forceinline__ uint8_t sat_u8(float v) { return (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v)); }
// Un-fused arithmetic: the host code these kernels restate rounds every product and every sum (HIP's __fmul_rn / __fadd_rn are
// plain operators, which hipcc contracts into fmas: a grey pixel's saturation result 123.0 came out as 122.99999 here and 123.0
// there).  Operations built under `fp contract(off)` carry no contract flag and stay separate after inlining.
__device__ __forceinline__ float mul_nf(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_nf(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_nf(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double dmul_nf(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double dsub_nf(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}

__device__ __forceinline__ void aug_tap(int d, int src, int dst, bool clamp_ofs, int &ofs, int &c0, int &c1) {
  // (explicitly un-fused: hipcc contracts a * b - c into an fma, cv::resize's host code does not)
  float f = (float)dsub_nf(dmul_nf((double)d + 0.5, (double)src / (double)dst), 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_ofs) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
  }
  ofs = s;
  c0 = (int)max(-32768l, min(32767l, lrintf(mul_nf(sub_nf(1.f, f), 2048.f))));
  c1 = (int)max(-32768l, min(32767l, lrintf(mul_nf(f, 2048.f))));
}

// crop + resize + flip: one thread per output pixel
__global__ __launch_bounds__(256) void aug_crop_resize_kernel(const uint8_t *__restrict__ src, int Hs, int Ws,
                                                              const tn_aug_frame *__restrict__ fr, int size, uint8_t *__restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= size) return;
  const tn_aug_frame f = fr[b];
  const uint8_t *img = src + ((size_t)b * Hs + f.y0) * Ws * 3 + (size_t)f.x0 * 3;      // the crop window, row pitch Ws * 3
  const int xo = f.flip ? size - 1 - x : x;
  uint8_t *o = dst + (((size_t)b * size + y) * size + xo) * 3;
  const size_t pitch = (size_t)Ws * 3;
  if (f.cw == 2 * size && f.ch == 2 * size) {      // exact 2x reduction: INTER_AREA fast path
    const uint8_t *p = img + (size_t)(2 * y) * pitch + (size_t)(2 * x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((p[c] + p[3 + c] + p[pitch + c] + p[pitch + 3 + c] + 2) >> 2);
    return;
  }
  int tx, tx0, tx1, ty, ty0, ty1;
  aug_tap(x, f.cw, size, true, tx, tx0, tx1);
  aug_tap(y, f.ch, size, false, ty, ty0, ty1);
  const int r0 = ty < 0 ? 0 : (ty < f.ch ? ty : f.ch - 1);
  const int r1 = ty + 1 < 0 ? 0 : (ty + 1 < f.ch ? ty + 1 : f.ch - 1);
  const int c1 = tx + 1 < f.cw ? tx + 1 : f.cw - 1;
  const uint8_t *p0 = img + (size_t)r0 * pitch, *p1 = img + (size_t)r1 * pitch;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = p0[tx * 3 + c] * tx0 + p0[c1 * 3 + c] * tx1;
    const int S1 = p1[tx * 3 + c] * tx0 + p1[c1 * 3 + c] * tx1;
    o[c] = (uint8_t)((((ty0 * (S0 >> 4)) >> 16) + ((ty1 * (S1 >> 4)) >> 16) + 2) >> 2);
  }
}

__device__ __forceinline__ float aug_gray(const uint8_t v[3]) {
  return add_nf(add_nf(mul_nf((float)v[0], 0.299f), mul_nf((float)v[1], 0.587f)), mul_nf((float)v[2], 0.114f));
}
// operator `op` of the jitter on one pixel (0 brightness, 1 contrast, 2 saturation, 3 hue = nothing)
__device__ __forceinline__ void aug_op(int op, const tn_aug_frame &f, float gray_mean, uint8_t v[3]) {
  if (op == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = sat_u8(mul_nf((float)v[c], f.brightness));
  } else if (op == 1) {
    const float beta = mul_nf(sub_nf(1.f, f.contrast), gray_mean);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = sat_u8(add_nf(mul_nf((float)v[c], f.contrast), beta));
  } else if (op == 2) {
    const float g = mul_nf(aug_gray(v), sub_nf(1.f, f.saturation));
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = sat_u8(add_nf(mul_nf((float)v[c], f.saturation), g));
  }
}

// mean of 0.299 R + 0.587 G + 0.114 B over the image AS THE CONTRAST OPERATOR SEES IT (the operators in front of it applied):
// one workgroup per frame; every per-pixel grey value is a float, their sum in double is exact in any order (multiples of 2^-27
// below 2^26), so the result does not depend on how the reduction is arranged
__global__ __launch_bounds__(256) void aug_gray_mean_kernel(const uint8_t *__restrict__ img, const tn_aug_frame *__restrict__ fr, int size,
                                                            float *__restrict__ gray_mean) {
  __shared__ double red[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const tn_aug_frame f = fr[b];
  const long n = (long)size * size;
  const uint8_t *p = img + (size_t)b * n * 3;
  double acc = 0.0;
  for (long i = t; i < n; i += 256) {
    uint8_t v[3] = {p[i * 3], p[i * 3 + 1], p[i * 3 + 2]};
    for (int k = 0; k < 4; ++k) {
      const int op = (f.order >> (2 * k)) & 3;
      if (op == 1) break;
      aug_op(op, f, 0.f, v);
    }
    acc += (double)aug_gray(v);
  }
  red[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) gray_mean[b] = (float)(red[0] / (double)n);
}

__global__ __launch_bounds__(256) void aug_color_kernel(const uint8_t *__restrict__ img, const tn_aug_frame *__restrict__ fr, int size,
                                                        const float *__restrict__ gray_mean, uint8_t *__restrict__ dst) {
  const long n = (long)size * size;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  const tn_aug_frame f = fr[b];
  const uint8_t *p = img + ((size_t)b * n + i) * 3;
  uint8_t v[3] = {p[0], p[1], p[2]};
  const float gm = gray_mean[b];
  for (int k = 0; k < 4; ++k) aug_op((f.order >> (2 * k)) & 3, f, gm, v);
  uint8_t *o = dst + ((size_t)b * n + i) * 3;
  o[0] = sat_u8(add_nf((float)v[0], f.light[0]));
  o[1] = sat_u8(add_nf((float)v[1], f.light[1]));
  o[2] = sat_u8(add_nf((float)v[2], f.light[2]));
}

}  // namespace

extern "C" int tn_augment_forward(tn_ctx *ctx, const uint8_t *src, int batch, int src_h, int src_w, const tn_aug_frame *frames_dev,
                                  const tn_aug_frame *frames_host, int size, uint8_t *tmp, float *gray_tmp, uint8_t *dst) {
  TN_REQUIRE(ctx && src && frames_dev && frames_host && tmp && gray_tmp && dst, "tn_augment_forward: null argument");
  TN_REQUIRE(batch > 0 && batch <= 65535 && src_h > 0 && src_w > 0 && size > 0 && size <= 65535, "tn_augment_forward: bad shape");
  for (int b = 0; b < batch; ++b) {
    const tn_aug_frame &f = frames_host[b];
    TN_REQUIRE(f.x0 >= 0 && f.y0 >= 0 && f.cw > 0 && f.ch > 0 && f.x0 + f.cw <= src_w && f.y0 + f.ch <= src_h,
               "tn_augment_forward: a crop window leaves the frame");
    int seen = 0;
    for (int k = 0; k < 4; ++k) seen |= 1 << ((f.order >> (2 * k)) & 3);
    TN_REQUIRE(seen == 15 && (f.order >> 8) == 0, "tn_augment_forward: order must be a permutation of the four jitter operators, two bits each");
  }
  TN_ON_DEVICE(ctx->device);
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(aug_crop_resize_kernel, dim3((size + 255) / 256, size, batch), dim3(256), 0, s, src, src_h, src_w, frames_dev, size, tmp);
  hipLaunchKernelGGL(aug_gray_mean_kernel, dim3(batch), dim3(256), 0, s, (const uint8_t *)tmp, frames_dev, size, gray_tmp);
  hipLaunchKernelGGL(aug_color_kernel, dim3((unsigned)(((long)size * size + 255) / 256), batch), dim3(256), 0, s, (const uint8_t *)tmp,
                     frames_dev, size, (const float *)gray_tmp, dst);
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

// Test / tuning hooks: run a single encoder kernel on caller-provided device
// buffers so the -m gpu parity tests can pin each kernel against the oracle on
// its own (ragged M, channel offsets) and scripts/kbench.py can time one kernel
// at one layer shape.  Host-weight variants fold/pack exactly as
// tn_densenet121_create does.
#include <cstring>
#include <vector>

#include "common.h"
#include "linear.h"

namespace {
template <typename T>
T *up(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
}  // namespace

std::vector<f16> pack_conv3x3(const float *w);  // api.hip

// fp32 (32,128,3,3) -> the packed fp16 fragment image the conv3x3 kernel consumes (72*64*8 halfs).
extern "C" int tn_dbg_pack_conv3x3(const float *w_host, uint16_t *out_host) {
  TN_REQUIRE(w_host && out_host, "tn_dbg_pack_conv3x3: null argument");
  const std::vector<f16> p = pack_conv3x3(w_host);
  memcpy(out_host, p.data(), p.size() * sizeof(f16));
  return TN_OK;
}

// Asynchronous single launches on device-resident operands (weights already fp16 / packed).
extern "C" int tn_dbg_conv1x1_dev(tn_ctx *ctx, const void *x_f16, int ldx, int K, const float *scale,
                                  const float *shift, const void *w_f16, int N, void *y_f16, int ldy, int yoff, int M,
                                  int pool, int H, int W, int variant) {
  TN_REQUIRE(ctx && x_f16 && y_f16 && scale && shift && w_f16, "tn_dbg_conv1x1_dev: null argument");
  Conv1x1Args a{(const f16 *)x_f16, ldx, K, scale, shift, (const f16 *)w_f16, N, (f16 *)y_f16, ldy, yoff, M, pool, H, W};
  a.variant = variant & 0xffff;
  a.exact = (variant >> 17) & 1;      // bit 17: w is [N][2 K] = [hi | lo] (exact-weights mode)
  return launch_conv1x1(a, ctx->stream);
}

extern "C" int tn_dbg_conv3x3_dev(tn_ctx *ctx, const void *x_f16, const float *scale, const float *shift,
                                  const void *wp_f16, void *y_f16, int ldy, int yoff, int B, int H, int W,
                                  int variant) {
  TN_REQUIRE(ctx && x_f16 && y_f16 && scale && shift && wp_f16, "tn_dbg_conv3x3_dev: null argument");
  Conv3x3Args a{(const f16 *)x_f16, scale, shift, (const f16 *)wp_f16, (f16 *)y_f16, ldy, yoff, B * H * W, H, W};
  a.variant = variant;
  return launch_conv3x3(a, ctx->stream);
}

// y[m][yoff+n] = sum_k relu(scale[k]*x[m][k]+shift[k]) * w[n][k]   (pool: 2x2 mean first)
extern "C" int tn_dbg_conv1x1(tn_ctx *ctx, const void *x_f16, int ldx, int K, const float *scale_host,
                              const float *shift_host, const float *w_host /*[N][K]*/, int N, void *y_f16, int ldy,
                              int yoff, int M, int pool, int H, int W) {
  TN_REQUIRE(ctx && x_f16 && y_f16 && scale_host && shift_host && w_host, "tn_dbg_conv1x1: null argument");
  TN_ON_DEVICE(ctx->device);
  std::vector<f16> wh((size_t)N * K);
  for (size_t i = 0; i < wh.size(); ++i) wh[i] = (f16)w_host[i];
  f16 *w = up(wh);
  float *s = up(std::vector<float>(scale_host, scale_host + K));
  float *t = up(std::vector<float>(shift_host, shift_host + K));
  TN_REQUIRE(w && s && t, "tn_dbg_conv1x1: device allocation failed");
  const int rc = tn_dbg_conv1x1_dev(ctx, x_f16, ldx, K, s, t, w, N, y_f16, ldy, yoff, M, pool, H, W, 0);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(w); (void)hipFree(s); (void)hipFree(t);
  if (rc) return rc;
  TN_HIP_CHECK(e);
  return TN_OK;
}

// y[m][yoff+n] = conv3x3(relu(scale*x+shift), w (32,128,3,3), pad 1) over (B,H,W,128) NHWC
extern "C" int tn_dbg_conv3x3(tn_ctx *ctx, const void *x_f16, const float *scale_host, const float *shift_host,
                              const float *w_host, void *y_f16, int ldy, int yoff, int B, int H, int W) {
  TN_REQUIRE(ctx && x_f16 && y_f16 && scale_host && shift_host && w_host, "tn_dbg_conv3x3: null argument");
  TN_ON_DEVICE(ctx->device);
  f16 *w = up(pack_conv3x3(w_host));
  float *s = up(std::vector<float>(scale_host, scale_host + 128));
  float *t = up(std::vector<float>(shift_host, shift_host + 128));
  TN_REQUIRE(w && s && t, "tn_dbg_conv3x3: device allocation failed");
  const int rc = tn_dbg_conv3x3_dev(ctx, x_f16, s, t, w, y_f16, ldy, yoff, B, H, W, 0);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(w); (void)hipFree(s); (void)hipFree(t);
  if (rc) return rc;
  TN_HIP_CHECK(e);
  return TN_OK;
}

// One fused dense layer in place on buf (B,H,W,ldc): device-resident, pre-converted operands.
extern "C" int tn_dbg_dense_layer_dev(tn_ctx *ctx, void *buf_f16, int ldc, int K, const float *s1, const float *t1,
                                      const void *w1_f16, const float *s2, const float *t2, const void *w3p_f16,
                                      int B, int H, int W, unsigned long long *ts, int variant) {
  TN_REQUIRE(ctx && buf_f16 && s1 && t1 && w1_f16 && s2 && t2 && w3p_f16, "tn_dbg_dense_layer_dev: null argument");
  DenseLayerArgs a{(f16 *)buf_f16, ldc, K, s1, t1, (const f16 *)w1_f16, s2, t2, (const f16 *)w3p_f16, B, H, W};
  a.ts = ts;
  a.variant = variant & 0xffff;
  a.exact = (variant >> 17) & 1;      // bit 17: w1 is [128][2 Kp] = [hi | lo], w3p the hi image followed by the lo image
  return launch_dense_layer(a, ctx->stream);
}

// fp32 linear: y = x W^T + b
extern "C" int tn_dbg_linear(tn_ctx *ctx, const float *x, const float *w, const float *bias, float *y, int M, int N,
                             int K) {
  TN_REQUIRE(ctx && x && w && y, "tn_dbg_linear: null argument");
  TN_ON_DEVICE(ctx->device);
  return launch_linear_f32(x, K, w, K, bias, y, N, M, N, K, 0, ctx->stream);
}

// ---- strip-streaming fused dense layer (dense_strip.hip) ----
// fp32 (128,K) 1x1 weights with BN2's folded scale / shift (128 each) and (32,128,3,3) 3x3 weights -> the fragment images the
// strip kernel keeps in LDS ((K+16)*128 and 36864 halfs); the scale is multiplied into the weights before the fp16 rounding
extern "C" int tn_dbg_pack_strip(const float *w1_host, int K, const float *s2_host, const float *t2_host, uint16_t *w1s_out,
                                 const float *w3_host, uint16_t *w3s_out) {
  TN_REQUIRE(K > 0 && K % 32 == 0, "tn_dbg_pack_strip: K must be a multiple of 32");
  if (w1_host && w1s_out) {
    TN_REQUIRE(s2_host && t2_host, "tn_dbg_pack_strip: the 1x1 image needs BN2's scale and shift");
    std::vector<float> wf((size_t)128 * K);
    for (int n = 0; n < 128; ++n)
      for (int k = 0; k < K; ++k) wf[(size_t)n * K + k] = w1_host[(size_t)n * K + k] * s2_host[n];
    const std::vector<f16> p = pack_w1_strip(wf.data(), K, t2_host);
    memcpy(w1s_out, p.data(), p.size() * sizeof(f16));
  }
  if (w3_host && w3s_out) {
    const std::vector<f16> p = pack_w3_strip(w3_host);
    memcpy(w3s_out, p.data(), p.size() * sizeof(f16));
  }
  return TN_OK;
}

// One fused dense layer in place on buf (B,H,W,ldc), asynchronous, device-resident packed operands.
extern "C" int tn_dbg_dense_strip_dev(tn_ctx *ctx, void *buf_f16, int ldc, int K, const float *s1, const float *t1,
                                      const void *w1s_f16, const void *w3s_f16, int B, int H, int W, unsigned long long *ts) {
  TN_REQUIRE(ctx && buf_f16 && s1 && t1 && w1s_f16 && w3s_f16, "tn_dbg_dense_strip_dev: null argument");
  DenseStripArgs a{(f16 *)buf_f16, ldc, K, s1, t1, (const f16 *)w1s_f16, (const f16 *)w3s_f16, B, H, W};
  a.ts = ts;
  return launch_dense_strip(a, ctx->stream);
}

// Test hooks: run a single encoder kernel on caller-provided device activations
// so the -m gpu parity tests can pin each kernel against the oracle on its own
// (per-kernel shapes, ragged M, channel offsets).  Weights arrive as host fp32 in
// Gluon layout and are folded/packed exactly as tn_densenet121_create does.
#include <vector>

#include "common.h"

namespace {
template <typename T>
T *up(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
}  // namespace

// y[m][yoff+n] = sum_k relu(scale[k]*x[m][k]+shift[k]) * w[n][k]   (pool: 2x2 mean first)
extern "C" int tn_dbg_conv1x1(tn_ctx *ctx, const void *x_f16, int ldx, int K, const float *scale_host,
                              const float *shift_host, const float *w_host /*[N][K]*/, int N, void *y_f16, int ldy,
                              int yoff, int M, int pool, int H, int W) {
  TN_REQUIRE(ctx && x_f16 && y_f16 && scale_host && shift_host && w_host, "tn_dbg_conv1x1: null argument");
  TN_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<f16> wh((size_t)N * K);
  for (size_t i = 0; i < wh.size(); ++i) wh[i] = (f16)w_host[i];
  f16 *w = up(wh);
  float *s = up(std::vector<float>(scale_host, scale_host + K));
  float *t = up(std::vector<float>(shift_host, shift_host + K));
  TN_REQUIRE(w && s && t, "tn_dbg_conv1x1: device allocation failed");
  Conv1x1Args a{(const f16 *)x_f16, ldx, K, s, t, w, N, (f16 *)y_f16, ldy, yoff, M, pool, H, W};
  const int rc = launch_conv1x1(a, ctx->stream);
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
  TN_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<f16> p((size_t)72 * 64 * 8);
  for (int s = 0; s < 72; ++s) {
    const int tap = s >> 3, kk = s & 7, ky = tap / 3, kx = tap % 3;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int n = l & 31, c = kk * 16 + (l >> 5) * 8 + j;
        p[((size_t)s * 64 + l) * 8 + j] = (f16)w_host[(((size_t)n * 128 + c) * 3 + ky) * 3 + kx];
      }
  }
  f16 *w = up(p);
  float *s = up(std::vector<float>(scale_host, scale_host + 128));
  float *t = up(std::vector<float>(shift_host, shift_host + 128));
  TN_REQUIRE(w && s && t, "tn_dbg_conv3x3: device allocation failed");
  Conv3x3Args a{(const f16 *)x_f16, s, t, w, (f16 *)y_f16, ldy, yoff, B * H * W, H, W};
  const int rc = launch_conv3x3(a, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(w); (void)hipFree(s); (void)hipFree(t);
  if (rc) return rc;
  TN_HIP_CHECK(e);
  return TN_OK;
}

// fp32 linear: y = x W^T + b
#include "linear.h"
extern "C" int tn_dbg_linear(tn_ctx *ctx, const float *x, const float *w, const float *bias, float *y, int M, int N,
                             int K) {
  TN_REQUIRE(ctx && x && w && y, "tn_dbg_linear: null argument");
  TN_HIP_CHECK(hipSetDevice(ctx->device));
  return launch_linear_f32(x, K, w, K, bias, y, N, M, N, K, 0, ctx->stream);
}

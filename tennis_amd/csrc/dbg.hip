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
  if ((variant >> 18) & 1) {          // bit 18: the warp-specialised transition kernel (trans_ws.hip); the fragment image of w is built per call
    TN_REQUIRE(trans_ws_supported(a), "tn_dbg_conv1x1_dev: geometry not supported by trans_ws");
    TN_ON_DEVICE(ctx->device);
    static f16 *frags[64] = {};             // (a test hook: one scratch image per device, grown on demand, never freed; not thread-safe)
    static size_t frag_halves[64] = {};
    TN_REQUIRE(ctx->device >= 0 && ctx->device < 64, "tn_dbg_conv1x1_dev: device index out of range");
    f16 *&frag = frags[ctx->device];
    if (frag_halves[ctx->device] < (size_t)N * K) {
      if (frag) (void)hipFree(frag);
      frag = nullptr; frag_halves[ctx->device] = 0;
      TN_HIP_CHECK(hipMalloc((void **)&frag, (size_t)N * K * sizeof(f16)));
      frag_halves[ctx->device] = (size_t)N * K;
    }
    const int rc = launch_pack_trans_frags(a.w, N, K, frag, ctx->stream);
    if (rc) return rc;
    a.wfrag = frag;
  }
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

// ---- strip-streaming fused dense layer (dense_strip_impl.h) ----
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


// ---- the LDS-resident 7x7 dense block (dense_block7.hip) ----
// nl layers from K0 input channels: w1_all = the (128, K_l) 1x1 weights one after the other, s1_all / t1_all the folded BN1
// scale / shift (K_l each), s2_all / t2_all the folded BN2 scale / shift (128 per layer), w3_all nl x (32,128,3,3).
struct tn_dbg_block7 {
  tn_ctx *ctx;
  void *wa = nullptr, *wb = nullptr, *tab = nullptr;
  DenseBlock7Args args;
};

extern "C" int tn_dbg_block7_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                                    const float *s2_all, const float *t2_all, const float *w3_all, void **out) {
  TN_REQUIRE(ctx && w1_all && s1_all && t1_all && s2_all && t2_all && w3_all && out, "tn_dbg_block7_create: null argument");
  TN_REQUIRE(dense_block7_supported(7, 7, K0, nl), "tn_dbg_block7_create: unsupported geometry");
  TN_ON_DEVICE(ctx->device);
  std::vector<std::vector<float>> folded(nl);
  std::vector<Block7Layer> layers(nl);
  size_t o1 = 0, ok = 0;
  for (int l = 0; l < nl; ++l) {
    const int K = K0 + 32 * l;
    folded[l].resize((size_t)128 * K);
    for (int n = 0; n < 128; ++n)
      for (int k = 0; k < K; ++k) folded[l][(size_t)n * K + k] = w1_all[o1 + (size_t)n * K + k] * s2_all[(size_t)l * 128 + n];
    layers[l] = Block7Layer{folded[l].data(), w3_all + (size_t)l * 32 * 128 * 9, s1_all + ok, t1_all + ok, t2_all + (size_t)l * 128};
    o1 += (size_t)128 * K;
    ok += K;
  }
  const Block7Image img = pack_block7(layers, K0);
  tn_dbg_block7 *b = new tn_dbg_block7();
  b->ctx = ctx;
  auto up = [&](void **dst, const void *src, size_t bytes) {
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up(&b->wa, img.wa.data(), img.wa.size() * sizeof(f16)) || !up(&b->wb, img.wb.data(), img.wb.size() * sizeof(f16)) ||
      !up(&b->tab, img.tab.data(), img.tab.size() * sizeof(float))) {
    tn_set_error("tn_dbg_block7_create: device allocation failed");
    return TN_ERR_NOMEM;
  }
  b->args = DenseBlock7Args{nullptr, 0, K0, nl, 0, (const f16 *)b->wa, (const f16 *)b->wb, (const float *)b->tab};
  for (int w = 0; w < 4; ++w) { b->args.a_off[w] = img.a_off[w]; b->args.b_off[w] = img.b_off[w]; }
  *out = b;
  return TN_OK;
}

extern "C" int tn_dbg_block7_run_ts(void *handle, void *buf_f16, int ldc, int B, unsigned long long *ts);
extern "C" int tn_dbg_block7_run(void *handle, void *buf_f16, int ldc, int B) { return tn_dbg_block7_run_ts(handle, buf_f16, ldc, B, nullptr); }
extern "C" int tn_dbg_block7_run_ts(void *handle, void *buf_f16, int ldc, int B, unsigned long long *ts) {
  tn_dbg_block7 *b = (tn_dbg_block7 *)handle;
  TN_REQUIRE(b && buf_f16, "tn_dbg_block7_run: null argument");
  TN_ON_DEVICE(b->ctx->device);
  DenseBlock7Args a = b->args;
  a.buf = (f16 *)buf_f16; a.ldc = ldc; a.B = B; a.ts = ts;
  return launch_dense_block7(a, b->ctx->stream);
}

extern "C" void tn_dbg_block7_destroy(void *handle) {
  tn_dbg_block7 *b = (tn_dbg_block7 *)handle;
  if (!b) return;
  (void)hipFree(b->wa); (void)hipFree(b->wb); (void)hipFree(b->tab);
  delete b;
}


// ---- the streamed 14x14 dense block (dense_block14.hip) ----
// same operand convention as tn_dbg_block7_create
struct tn_dbg_block14 {
  tn_ctx *ctx;
  void *stream = nullptr, *scratch = nullptr;
  int scratch_frames = 0;
  DenseBlock14Args args;
};

extern "C" int tn_dbg_block14_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                                     const float *s2_all, const float *t2_all, const float *w3_all, void **out) {
  TN_REQUIRE(ctx && w1_all && s1_all && t1_all && s2_all && t2_all && w3_all && out, "tn_dbg_block14_create: null argument");
  TN_REQUIRE(dense_block14_supported(14, 14, K0, nl), "tn_dbg_block14_create: unsupported geometry");
  TN_ON_DEVICE(ctx->device);
  std::vector<std::vector<float>> folded(nl);
  std::vector<Block14Layer> layers(nl);
  size_t o1 = 0, ok = 0;
  for (int l = 0; l < nl; ++l) {
    const int K = K0 + 32 * l;
    folded[l].resize((size_t)128 * K);
    for (int n = 0; n < 128; ++n)
      for (int k = 0; k < K; ++k) folded[l][(size_t)n * K + k] = w1_all[o1 + (size_t)n * K + k] * s2_all[(size_t)l * 128 + n];
    layers[l] = Block14Layer{folded[l].data(), w3_all + (size_t)l * 32 * 128 * 9, s1_all + ok, t1_all + ok, t2_all + (size_t)l * 128};
    o1 += (size_t)128 * K;
    ok += K;
  }
  const std::vector<unsigned char> img = pack_block14(layers, K0);
  tn_dbg_block14 *b = new tn_dbg_block14();
  b->ctx = ctx;
  if (hipMalloc(&b->stream, img.size()) != hipSuccess || hipMemcpy(b->stream, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) {
    tn_set_error("tn_dbg_block14_create: device allocation failed");
    delete b;
    return TN_ERR_NOMEM;
  }
  b->args = DenseBlock14Args{nullptr, 0, K0, nl, 0, (const unsigned char *)b->stream, dense_block14_units(K0, nl)};
  *out = b;
  return TN_OK;
}

extern "C" int tn_dbg_block14_run_ts(void *handle, void *buf_f16, int ldc, int B, unsigned long long *ts) {
  tn_dbg_block14 *b = (tn_dbg_block14 *)handle;
  TN_REQUIRE(b && buf_f16, "tn_dbg_block14_run: null argument");
  TN_ON_DEVICE(b->ctx->device);
  if (b->scratch_frames < B) {
    (void)hipFree(b->scratch);
    b->scratch = nullptr; b->scratch_frames = 0;
    if (hipMalloc(&b->scratch, (size_t)B * dense_block14_scratch_halfs() * sizeof(f16)) != hipSuccess) {
      tn_set_error("tn_dbg_block14_run: device allocation failed");
      return TN_ERR_NOMEM;
    }
    b->scratch_frames = B;
    TN_HIP_CHECK(hipMemset(b->scratch, 0, (size_t)B * dense_block14_scratch_halfs() * sizeof(f16)));
  }
  DenseBlock14Args a = b->args;
  a.buf = (f16 *)buf_f16; a.ldc = ldc; a.B = B; a.ts = ts; a.scratch = (f16 *)b->scratch;
  return launch_dense_block14(a, b->ctx->stream);
}
extern "C" int tn_dbg_block14_run(void *handle, void *buf_f16, int ldc, int B) { return tn_dbg_block14_run_ts(handle, buf_f16, ldc, B, nullptr); }

extern "C" void tn_dbg_block14_destroy(void *handle) {
  tn_dbg_block14 *b = (tn_dbg_block14 *)handle;
  if (!b) return;
  (void)hipFree(b->stream);
  (void)hipFree(b->scratch);
  delete b;
}

// ---- the streamed 28x28 dense block (dense_block28.hip) ----
// same operand convention as tn_dbg_block7_create
struct tn_dbg_block28 {
  tn_ctx *ctx;
  void *stream = nullptr, *scratch = nullptr;
  int scratch_frames = 0;
  DenseBlock28Args args;
};

extern "C" int tn_dbg_block28_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                                     const float *s2_all, const float *t2_all, const float *w3_all, void **out) {
  TN_REQUIRE(ctx && w1_all && s1_all && t1_all && s2_all && t2_all && w3_all && out, "tn_dbg_block28_create: null argument");
  TN_REQUIRE(dense_block28_supported(28, 28, K0, nl), "tn_dbg_block28_create: unsupported geometry");
  TN_ON_DEVICE(ctx->device);
  std::vector<std::vector<float>> folded(nl);
  std::vector<Block14Layer> layers(nl);
  size_t o1 = 0, ok = 0;
  for (int l = 0; l < nl; ++l) {
    const int K = K0 + 32 * l;
    folded[l].resize((size_t)128 * K);
    for (int n = 0; n < 128; ++n)
      for (int k = 0; k < K; ++k) folded[l][(size_t)n * K + k] = w1_all[o1 + (size_t)n * K + k] * s2_all[(size_t)l * 128 + n];
    layers[l] = Block14Layer{folded[l].data(), w3_all + (size_t)l * 32 * 128 * 9, s1_all + ok, t1_all + ok, t2_all + (size_t)l * 128};
    o1 += (size_t)128 * K;
    ok += K;
  }
  const std::vector<unsigned char> img = pack_block28(layers, K0);
  tn_dbg_block28 *b = new tn_dbg_block28();
  b->ctx = ctx;
  if (hipMalloc(&b->stream, img.size()) != hipSuccess || hipMemcpy(b->stream, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) {
    tn_set_error("tn_dbg_block28_create: device allocation failed");
    delete b;
    return TN_ERR_NOMEM;
  }
  b->args = DenseBlock28Args{nullptr, 0, K0, nl, 0, (const unsigned char *)b->stream, dense_block28_units(K0, nl)};
  *out = b;
  return TN_OK;
}

extern "C" int tn_dbg_block28_run_ts(void *handle, void *buf_f16, int ldc, int B, unsigned long long *ts) {
  tn_dbg_block28 *b = (tn_dbg_block28 *)handle;
  TN_REQUIRE(b && buf_f16, "tn_dbg_block28_run: null argument");
  TN_ON_DEVICE(b->ctx->device);
  if (b->scratch_frames < B) {
    (void)hipFree(b->scratch);
    b->scratch = nullptr; b->scratch_frames = 0;
    if (hipMalloc(&b->scratch, (size_t)B * dense_block28_scratch_halfs() * sizeof(f16)) != hipSuccess) {
      tn_set_error("tn_dbg_block28_run: device allocation failed");
      return TN_ERR_NOMEM;
    }
    b->scratch_frames = B;
    TN_HIP_CHECK(hipMemset(b->scratch, 0, (size_t)B * dense_block28_scratch_halfs() * sizeof(f16)));
  }
  DenseBlock28Args a = b->args;
  a.buf = (f16 *)buf_f16; a.ldc = ldc; a.B = B; a.ts = ts; a.scratch = (f16 *)b->scratch;
  return launch_dense_block28(a, b->ctx->stream);
}
extern "C" int tn_dbg_block28_run(void *handle, void *buf_f16, int ldc, int B) { return tn_dbg_block28_run_ts(handle, buf_f16, ldc, B, nullptr); }

extern "C" void tn_dbg_block28_destroy(void *handle) {
  tn_dbg_block28 *b = (tn_dbg_block28 *)handle;
  if (!b) return;
  (void)hipFree(b->stream);
  (void)hipFree(b->scratch);
  delete b;
}

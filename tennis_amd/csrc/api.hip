// C ABI of libtennis_hip.so (see include/tennis_hip.h): context, DenseNet-121
// frame encoder (weight folding/packing + launch schedule), Dense, bi-RNN,
// temporal pooling and the PRF1 histogram.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "linear.h"
#include "rnn.h"
#include "train.h"

bool dense_layer_big_supported(int H, int W);   // dense_layer_big.hip

// ---------------------------------------------------------------------------
static thread_local std::string g_err;
void tn_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *tn_last_error(void) { return g_err.c_str(); }
extern "C" int tn_version(void) { return 100; }

// ---- context ----------------------------------------------------------------
extern "C" int tn_ctx_create(int device, void *stream, int own_stream, tn_ctx **out) {
  TN_REQUIRE(out != nullptr, "tn_ctx_create: out is null");
  int n = 0;
  TN_HIP_CHECK(hipGetDeviceCount(&n));
  TN_REQUIRE(device >= 0 && device < n, "tn_ctx_create: no such device");
  TN_ON_DEVICE(device);
  tn_ctx *c = new tn_ctx();
  c->device = device;
  c->own_stream = own_stream != 0;
  if (c->own_stream) {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      tn_set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
      return TN_ERR_HIP;
    }
  } else {
    c->stream = (hipStream_t)stream;
  }
  *out = c;
  return TN_OK;
}
extern "C" void *tn_ctx_stream(tn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int tn_ctx_sync(tn_ctx *ctx) {
  TN_REQUIRE(ctx, "tn_ctx_sync: null ctx");
  TN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return TN_OK;
}
extern "C" int tn_ctx_destroy(tn_ctx *ctx) {
  if (!ctx) return TN_OK;
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return TN_OK;
}

// ---- small host helpers -------------------------------------------------------
namespace {

struct DevPool {  // owns device allocations of one handle
  std::vector<void *> ptrs;
  size_t bytes = 0;
  bool failed = false;
  void *alloc(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, n ? n : 16) != hipSuccess) { failed = true; return nullptr; }
    ptrs.push_back(p);
    bytes += n;
    return p;
  }
  template <typename T>
  T *upload(const std::vector<T> &h) {
    T *d = (T *)alloc(h.size() * sizeof(T));
    if (d && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { failed = true; return nullptr; }
    return d;
  }
  void release() {
    for (void *p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

struct ParamMap {
  std::map<std::string, const tn_param *> m;
  ParamMap(const tn_param *p, int n) {
    for (int i = 0; i < n; ++i) m[p[i].name] = &p[i];
  }
  const float *get(const std::string &name, int64_t numel) const {
    auto it = m.find(name);
    if (it == m.end()) {
      tn_set_error("missing parameter: " + name);
      return nullptr;
    }
    if (it->second->numel != numel) {
      tn_set_error("parameter " + name + " has " + std::to_string(it->second->numel) + " elements, expected " +
                   std::to_string(numel));
      return nullptr;
    }
    return it->second->data_host;
  }
};

constexpr float kBnEps = 1e-5f;

// Folded inference BatchNorm: y = x*scale + shift.
bool fold_bn(const ParamMap &pm, const std::string &name, int c, std::vector<float> &scale, std::vector<float> &shift) {
  const float *g = pm.get(name + "_gamma", c), *b = pm.get(name + "_beta", c);
  const float *mu = pm.get(name + "_running_mean", c), *var = pm.get(name + "_running_var", c);
  if (!g || !b || !mu || !var) return false;
  scale.resize(c);
  shift.resize(c);
  bn_scale_shift(g, b, mu, var, c, kBnEps, scale.data(), shift.data());
  return true;
}

// fp16 copy with 64 trailing zeros: the LDS-DMA k-tile of the fused dense layer may read up
// to 32 halfs past the last row when K % 64 == 32
std::vector<f16> to_f16(const float *w, size_t n) {
  std::vector<f16> h(n + 64, (f16)0.f);
  for (size_t i = 0; i < n; ++i) h[i] = (f16)w[i];
  return h;
}

// Exact-weights mode: w = hi + lo with hi = fp16(w), lo = fp16(w - hi) (22 bits of the fp32 weight survive).
// rows x k fp32 -> [rows][2 kp] fp16 = [hi (k, zero-padded to kp) | lo (...)], + 64 halves of slack like to_f16
std::vector<f16> split_hi_lo_rows(const float *w, int rows, int k, int kp) {
  std::vector<f16> h((size_t)rows * 2 * kp + 64, (f16)0.f);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < k; ++c) {
      const float v = w[(size_t)r * k + c];
      const f16 hi = (f16)v;
      h[(size_t)r * 2 * kp + c] = hi;
      h[(size_t)r * 2 * kp + kp + c] = (f16)(v - (float)hi);
    }
  return h;
}

}  // namespace
// 3x3 weights (32,128,3,3) -> MFMA B fragments [72 k-steps][64 lanes][8]:
// k-step s = tap*8 + kk; lane l: n = l&31, channel = kk*16 + (l>>5)*8 + j.
std::vector<f16> pack_conv3x3(const float *w) {
  // two MFMA operand layouts back to back (72*64*8 halves each):
  //  [0]     v_mfma_f32_32x32x16_f16 A fragments [9 taps x 8 k16-steps][64 lanes][8]   (conv3x3.hip)
  //  [36864] v_mfma_f32_16x16x32_f16 A fragments [9 taps][4 k32-steps][2 n-frags][64 lanes][8]   (dense_layer_big.hip):
  //          lane l: out channel nf*16 + (l&15), in channel kk*32 + (l>>4)*8 + j
  std::vector<f16> p((size_t)2 * 72 * 64 * 8);
  for (int s = 0; s < 72; ++s) {
    const int tap = s >> 3, kk = s & 7, ky = tap / 3, kx = tap % 3;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int n = l & 31, c = kk * 16 + (l >> 5) * 8 + j;
        p[((size_t)s * 64 + l) * 8 + j] = (f16)w[(((size_t)n * 128 + c) * 3 + ky) * 3 + kx];
      }
  }
  f16 *q = p.data() + (size_t)72 * 64 * 8;
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap % 3;
    for (int kk = 0; kk < 4; ++kk)
      for (int nf = 0; nf < 2; ++nf)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int n = nf * 16 + (l & 15), c = kk * 32 + (l >> 4) * 8 + j;
            q[((((size_t)tap * 4 + kk) * 2 + nf) * 64 + l) * 8 + j] = (f16)w[(((size_t)n * 128 + c) * 3 + ky) * 3 + kx];
          }
  }
  return p;
}
namespace {

// stem weights (64,3,7,7) -> MFMA A fragments [7 ky][4 nfrag][64 lanes][8]:
// lane l: n = nf*16 + (l&15); k slot (l>>4)*8 + j -> x-tap kx = slot>>2, channel c = slot&3 (the eighth tap is zero);
// zero_first: the zero tap comes first, kx = (slot>>2) - 1 (operand alignment of the fused stem + maxpool kernel).
std::vector<f16> pack_stem(const float *w, bool zero_first) {
  std::vector<f16> p((size_t)7 * 4 * 64 * 8);
  for (int ky = 0; ky < 7; ++ky)
    for (int nf = 0; nf < 4; ++nf)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int n = nf * 16 + (l & 15), slot = (l >> 4) * 8 + j, kx = (slot >> 2) - (zero_first ? 1 : 0), c = slot & 3;
          float v = 0.f;
          if (kx >= 0 && kx < 7 && c < 3) v = w[(((size_t)n * 3 + c) * 7 + ky) * 7 + kx];
          p[(((size_t)ky * 4 + nf) * 64 + l) * 8 + j] = (f16)v;
        }
  return p;
}

struct EventTimer {  // brackets launches with HIP events when enabled
  bool on = false;
  hipStream_t s = nullptr;
  struct Rec { hipEvent_t a, b; int fam; };
  std::vector<Rec> recs;
  std::vector<tn_kernel_stat> fams;
  int family(const char *name) {
    for (size_t i = 0; i < fams.size(); ++i)
      if (!strcmp(fams[i].name, name)) return (int)i;
    tn_kernel_stat st;
    memset(&st, 0, sizeof(st));
    strncpy(st.name, name, sizeof(st.name) - 1);
    fams.push_back(st);
    return (int)fams.size() - 1;
  }
  void begin(const char *name, double flops, double bytes) {
    if (!on) return;
    Rec r;
    r.fam = family(name);
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    fams[r.fam].launches += 1;
    fams[r.fam].flops += flops;
    fams[r.fam].bytes += bytes;
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
  }
  void end() {
    if (!on) return;
    (void)hipEventRecord(recs.back().b, s);
  }
  void finish() {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    for (auto &r : recs) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, r.a, r.b);
      fams[r.fam].ms += ms;
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
};

}  // namespace

// ---- DenseNet-121 encoder --------------------------------------------------------
struct tn_encoder {
  tn_ctx *ctx;
  DevPool pool;
  int H, W, maxB, feat_dim;
  int Hs, Ws;              // stem conv output
  int Hb[4], Wb[4];        // dense block spatial sizes
  int Cin[4], Cb[4];       // block input / total channels
  int PH, PW;
  f16 *stem_wp, *stem_wp_zf, *stem_wp_zf_lo = nullptr;
  float *stem_scale, *stem_shift, *stem_shift_u8;
  float *stem_floor = nullptr;                               // centred stem output (round 6): the ReLU's floor -m_c on the device
  std::vector<float> stem_centre;                            // m_c on the host (read_tap / input_means add it back)
  struct DenseLayer { float *s1, *t1; f16 *w1; float *s2, *t2; f16 *w3p; int cin; f16 *w1s = nullptr, *w3s = nullptr; };   // w1s / w3s: fragment images of the strip kernel
  std::vector<DenseLayer> layers[4];
  struct Trans { float *s, *t; f16 *w; int cin, cout; f16 *wfrag = nullptr; } trans[3];      // wfrag: w in MFMA operand order (trans_ws.hip)
  float *head_s, *head_t;
  float *zeros128 = nullptr;   // a BatchNorm shift of zeros (un-fused dense layers: the shift was added by the 1x1)
  f16 *stem_out, *bott, *blockbuf[4];
  float *head32 = nullptr;     // the last block's map once more in fp32 (what the head reads): written by the last transition and the 7x7 block kernel
  size_t workspace_bytes;
  int last_batch;
  bool fuse, split;
  int nsplit;                 // side streams in use (TN_SPLIT, default 2)
  int dl_variant;             // tuning hook: TN_DL_VARIANT -> DenseLayerArgs.variant
  bool chain;                 // whole-frame blocks (14x14, 7x7) run all their layers in one launch (TN_NO_CHAIN disables)
  bool exact = false;         // TN_ENC_EXACT_WEIGHTS: dense-layer and transition weights as hi + lo fp16 pairs
  bool strip = true;          // 56x56 / 28x28 layers with K <= 320 run on the strip-streaming kernel (TN_NO_STRIP disables)
  int strip_min_batch = 64;   // ... from this many frames per launch on (one workgroup per frame: small batches leave CUs idle)
  DenseLayerDev *chain_dev[4] = {nullptr, nullptr, nullptr, nullptr};
  float *calib_dev = nullptr;   // tn_densenet121_input_means: where the layer-wise pass leaves the mean of every convolution's input
  double *calib_scratch = nullptr;
  float *ones128 = nullptr;
  bool block7 = true;         // a 7x7 block runs on the LDS-resident kernel of dense_block7.hip (TN_NO_BLOCK7 disables)
  DenseBlock7Args b7[4] = {};  // its packed operands per block (buf == nullptr: not packed)
  bool block14 = true;        // a 14x14 block runs on the streamed kernel of dense_block14.hip (TN_NO_BLOCK14 disables)
  DenseBlock14Args b14[4] = {};
  f16 *b14_scratch[4] = {nullptr, nullptr, nullptr, nullptr};   // its k-step-major working copy of the block's frames
  bool block28 = false;       // a 28x28 block runs on the streamed kernel of dense_block28.hip (TN_BLOCK28=1 enables: measured, not the default)
  DenseBlock28Args b28[4] = {};
  f16 *b28_scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t side[4];
  hipEvent_t ev_in, ev_done[2][4];   // completion of the side streams, alternating per forward call
  bool pipelined = false;            // tn_densenet121_set_pipelined: the caller's stream is not made to wait inside forward
  bool last_interleaved = false;     // the previous split call took the whole-batch form
  int last_ws0 = 0;                  // first workspace frame slot of the last forward (read_tap)
  bool interleave = false;           // pipelined calls run WHOLE batches on alternating side streams (round 6, encoder_run), on two workspace sets
  long calls = 0;                    // forward calls so far
  int split_of[2] = {0, 0};          // side streams the call of each parity used (0: it ran on the caller's stream)
  int split_batch = 0;               // batch size of the last split call (pipelined calls share the workspace by row range)
};

static const int kBlockCfg[4] = {6, 12, 24, 16};

extern "C" int tn_densenet121_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c,
                                     int height, int width, int max_batch, tn_encoder **out) {
  return tn_densenet121_create_ex(ctx, params, n_params, prefix_c, height, width, max_batch, 0, out);
}

extern "C" int tn_densenet121_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c,
                                        int height, int width, int max_batch, int flags, tn_encoder **out) {
  TN_REQUIRE(ctx && params && out && prefix_c, "tn_densenet121_create: null argument");
  TN_REQUIRE((flags & ~TN_ENC_EXACT_WEIGHTS) == 0, "tn_densenet121_create_ex: unknown flag");
  TN_REQUIRE(max_batch > 0, "tn_densenet121_create: max_batch must be positive");
  TN_REQUIRE(height >= 224 && width >= 224 && height <= 1024 && width <= 1024,
             "tn_densenet121_create: input size must be in [224,1024] (AvgPool2D(7) needs a >=7x7 final map)");
  TN_ON_DEVICE(ctx->device);
  const std::string pre(prefix_c);
  ParamMap pm(params, n_params);
  tn_encoder *e = new tn_encoder();
  e->ctx = ctx;
  e->H = height; e->W = width; e->maxB = max_batch; e->last_batch = 0;
  e->fuse = getenv("TN_NO_FUSE") == nullptr;
  e->split = getenv("TN_NO_SPLIT") == nullptr;
  e->nsplit = getenv("TN_SPLIT") ? atoi(getenv("TN_SPLIT")) : 2;
  if (e->nsplit != 4) e->nsplit = 2;
  e->block7 = getenv("TN_NO_BLOCK7") == nullptr;
  e->block14 = getenv("TN_NO_BLOCK14") == nullptr;
  e->block28 = getenv("TN_BLOCK28") != nullptr && atoi(getenv("TN_BLOCK28")) != 0;
  e->chain = getenv("TN_NO_CHAIN") == nullptr;   // measured: -20% on the 14x14 / 7x7 blocks, +2.8% end to end
  e->dl_variant = getenv("TN_DL_VARIANT") ? atoi(getenv("TN_DL_VARIANT")) : 0;
  e->exact = (flags & TN_ENC_EXACT_WEIGHTS) != 0;
  e->strip = getenv("TN_NO_STRIP") == nullptr && !e->exact && e->fuse;
  if (getenv("TN_STRIP_MIN_BATCH")) e->strip_min_batch = atoi(getenv("TN_STRIP_MIN_BATCH"));
  for (int i = 0; i < 4; ++i) {
    if (hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_done[0][i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_done[1][i], hipEventDisableTiming) != hipSuccess) {
      tn_set_error("could not create the side streams");
      delete e;
      return TN_ERR_HIP;
    }
  }
  if (hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) != hipSuccess) { tn_set_error("hipEventCreate failed"); delete e; return TN_ERR_HIP; }
  e->Hs = (height + 6 - 7) / 2 + 1; e->Ws = (width + 6 - 7) / 2 + 1;
  int h = (e->Hs + 2 - 3) / 2 + 1, w = (e->Ws + 2 - 3) / 2 + 1, c = 64;
  for (int b = 0; b < 4; ++b) {
    e->Hb[b] = h; e->Wb[b] = w; e->Cin[b] = c; e->Cb[b] = c + 32 * kBlockCfg[b];
    c = e->Cb[b] / 2; h /= 2; w /= 2;
  }
  e->PH = e->Hb[3] / 7; e->PW = e->Wb[3] / 7;
  e->feat_dim = e->Cb[3] * e->PH * e->PW;
  auto fail = [&](int code) { e->pool.release(); delete e; return code; };
  if (e->PH < 1 || e->PW < 1) { tn_set_error("input too small for AvgPool2D(7)"); return fail(TN_ERR_INVALID); }
  if (e->Wb[0] > 240) { tn_set_error("input too wide for the conv3x3 LDS tile"); return fail(TN_ERR_INVALID); }
  if (e->exact) {      // the hi + lo weight passes: the 8-wave fused layer, the transition kernel and (round 6) the un-fused layer kernels
    const bool ok = e->fuse && (e->dl_variant & ~256) == 0;
    if (!ok) { tn_set_error("TN_ENC_EXACT_WEIGHTS needs the default kernels (no TN_NO_FUSE, no TN_DL_VARIANT)"); return fail(TN_ERR_INVALID); }
  }

  std::vector<float> s, t;
  // ---- centred stem output (round 6) ----
  // The pooled stem map is what every dense layer of block 1 and the first transition read, through a BatchNorm each.  Stored
  // as it is, a channel whose values sit many standard deviations from zero loses its information to fp16's RELATIVE precision:
  // the extreme case is a (near-)dead channel of batchnorm0 (gamma ~ 0: the output is the constant relu(beta)), whose
  // consumers normalise with a running variance at the epsilon floor - scale gamma / sqrt(1e-5) = 316 gamma on the rounding
  // error of a constant, 1e-2 on the features of EVERY frame (tests/tools/trained_like.py, scripts/round_study.py).  All of a
  // channel's consumers carry an estimate of its mean, their running_mean; the map is stored as relu(bn(conv)) - m_c with m_c
  // their average, and the constant goes into the consumers' shifts, shift' = shift + scale m_c (exact: no kernel knows).
  e->stem_centre.assign(64, 0.f);
  if (getenv("TN_NO_STEM_CENTRE") == nullptr) {
    std::vector<double> acc(64, 0.0);
    int cnt = 0;
    for (int l = 0; l <= kBlockCfg[0]; ++l) {
      const std::string bn = l < kBlockCfg[0] ? pre + "stage1_batchnorm" + std::to_string(2 * l) : pre + "batchnorm1";
      const float *mu = pm.get(bn + "_running_mean", l < kBlockCfg[0] ? 64 + 32 * l : 64 + 32 * kBlockCfg[0]);
      if (!mu) return fail(TN_ERR_MISSING);
      for (int c = 0; c < 64; ++c) acc[c] += mu[c];
      ++cnt;
    }
    for (int c = 0; c < 64; ++c) {
      const float m = (float)(acc[c] / cnt);
      e->stem_centre[c] = std::isfinite(m) ? m : 0.f;
    }
  }
  auto centre_shift = [&](std::vector<float> &sc, std::vector<float> &sh) {       // a consumer of block 1's channels 0 .. 63
    for (int c = 0; c < 64; ++c) sh[c] = (float)((double)sh[c] + (double)sc[c] * (double)e->stem_centre[c]);
  };
  {  // stem: conv0 + batchnorm0
    const float *w0 = pm.get(pre + "conv0_weight", 64 * 3 * 7 * 7);
    if (!w0 || !fold_bn(pm, pre + "batchnorm0", 64, s, t)) return fail(TN_ERR_MISSING);
    // the input normalisation's 1 / (255 std_c) goes into the weights before they are rounded (common.h "the stem's operand")
    std::vector<float> w0s((size_t)64 * 3 * 49), tu(64);
    for (int n = 0; n < 64; ++n) {
      double bias = 0.0;
      for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 49; ++k) {
          const size_t i = ((size_t)n * 3 + c) * 49 + k;
          w0s[i] = w0[i] * stem_wfactor(c);
          bias -= (double)(float)(f16)w0s[i] * stem_pad(c);
        }
      s[n] = (float)((double)s[n] / kStemWScale);
      tu[n] = (float)((double)t[n] + (double)s[n] * bias);
    }
    e->stem_wp = e->pool.upload(pack_stem(w0s.data(), false));
    e->stem_wp_zf = e->pool.upload(pack_stem(w0s.data(), true));
    if (e->exact) {     // w = hi + lo: the second fragment image (the constant of the integer staging then uses hi + lo as well)
      std::vector<float> lo(w0s.size());
      for (int n = 0; n < 64; ++n) {
        double bias = 0.0;
        for (int c = 0; c < 3; ++c)
          for (int k = 0; k < 49; ++k) {
            const size_t i = ((size_t)n * 3 + c) * 49 + k;
            const float hi = (float)(f16)w0s[i];
            lo[i] = w0s[i] - hi;
            bias -= ((double)hi + (double)(float)(f16)lo[i]) * stem_pad(c);
          }
        tu[n] = (float)((double)t[n] + (double)s[n] * bias);
      }
      e->stem_wp_zf_lo = e->pool.upload(pack_stem(lo.data(), true));
    }
    std::vector<float> fl(64);
    for (int n = 0; n < 64; ++n) {
      t[n] = (float)((double)t[n] - (double)e->stem_centre[n]);
      tu[n] = (float)((double)tu[n] - (double)e->stem_centre[n]);
      fl[n] = -e->stem_centre[n];
    }
    e->stem_scale = e->pool.upload(s);
    e->stem_shift = e->pool.upload(t);
    e->stem_shift_u8 = e->pool.upload(tu);
    e->stem_floor = e->pool.upload(fl);
  }
  e->zeros128 = e->pool.upload(std::vector<float>(128, 0.0f));
  e->ones128 = e->pool.upload(std::vector<float>(128, 1.0f));
  int outer = 1;
  for (int b = 0; b < 4; ++b) {
    const std::string sp = pre + "stage" + std::to_string(b + 1) + "_";
    const bool pack14 = e->fuse && e->block14 && !e->exact && dense_block14_supported(e->Hb[b], e->Wb[b], e->Cin[b], kBlockCfg[b]);
    const bool pack28 = e->fuse && e->block28 && !e->exact && dense_block28_supported(e->Hb[b], e->Wb[b], e->Cin[b], kBlockCfg[b]);
    const bool pack7 = (e->fuse && e->block7 && !e->exact && dense_block7_supported(e->Hb[b], e->Wb[b], e->Cin[b], kBlockCfg[b])) || pack14 || pack28;
    std::vector<std::vector<float>> h7[4];     // host copies for pack_block7: folded 1x1 weights, s1, t1, t2 per layer
    std::vector<const float *> h7w3;
    for (int l = 0; l < kBlockCfg[b]; ++l) {
      tn_encoder::DenseLayer L;
      L.cin = e->Cin[b] + 32 * l;
      const float *w1 = pm.get(sp + "conv" + std::to_string(2 * l) + "_weight", (int64_t)128 * L.cin);
      const float *w3 = pm.get(sp + "conv" + std::to_string(2 * l + 1) + "_weight", 32 * 128 * 9);
      if (!w1 || !w3) return fail(TN_ERR_MISSING);
      if (!fold_bn(pm, sp + "batchnorm" + std::to_string(2 * l), L.cin, s, t)) return fail(TN_ERR_MISSING);
      if (b == 0) centre_shift(s, t);
      // BN1 + ReLU as relu(s x + t) = sw clamp(x, lo, hi) + tc with lo, hi fp16 numbers (calib_host.hip::bn_relu_clamp_fold: no
      // arithmetic and no rounding in front of the 1x1); every kernel of the layer gets (lo, hi) as its constants, sw[k] w[n][k] as
      // its weights and sum_k w[n][k] tc[k] inside BN2's shift
      std::vector<float> sw1(L.cin), tc1(L.cin);
      bn_relu_clamp_fold(std::vector<float>(s).data(), std::vector<float>(t).data(), L.cin, s.data(), t.data(), sw1.data(), tc1.data());
      L.s1 = e->pool.upload(s); L.t1 = e->pool.upload(t);
      // x0[k] = clamp(0, lo, hi): the operand's value on the CLIPPED side of a channel whose ReLU is off at x = 0 (round 6, below)
      std::vector<float> x0(L.cin);
      for (int k = 0; k < L.cin; ++k) x0[k] = std::fmin(std::fmax(0.f, s[k]), t[k]);
      if (pack7) { h7[1].push_back(s); h7[2].push_back(t); h7w3.push_back(w3); }
      // The scale of the BatchNorm BEHIND the 1x1 convolution is folded into its weights before they are rounded to fp16
      // (or split into hi + lo): bn2(conv(a)) = conv'(a) + shift with w'[n][k] = scale[n] w[n][k].  That is how the fp16
      // model is defined (weights.as_fp16_model hands over w with scale[n] w[n][k] fp16-representable); the kernels that
      // still apply a scale get ones.
      if (!fold_bn(pm, sp + "batchnorm" + std::to_string(2 * l + 1), 128, s, t)) return fail(TN_ERR_MISSING);
      std::vector<float> w1f((size_t)128 * L.cin);
      bool in_range = true;
      // The constant of the clamp form, sum_k w[n][k] tc[k], is computed here in double from the exact weights, while the matrix
      // pipe multiplies the ROUNDED folded weight with clamp(x).  On the clipped side of a channel the two have to cancel
      // (sw lo + tc = 0), and they only do so to the precision of the rounded weight times |lo|: a near-dead BatchNorm channel
      // with a negative shift (scale 1e-5, threshold 1e4, folded weight in fp16's subnormals) left 3e-4 per weight that way
      // (scripts/dead_debug.py: 4.8e-3 on the features in the exact-weights mode).  Round 6: the operand is split at
      // x0 = clamp(0, lo, hi) - sw clamp(x) + tc = sw (clamp(x) - x0) + (tc + sw x0) - and the x0 part of the matrix product is
      // taken out of the shift with the SAME rounded weights the pipe uses: what is left of a weight's rounding error multiplies
      // clamp(x) - x0 (zero on the clipped side), what the exact weights multiply is tc + sw x0 (= s (x0 - c): no cancellation).
      for (int n = 0; n < 128; ++n) {
        double bias = 0.0, corr = 0.0;
        for (int k = 0; k < L.cin; ++k) {
          const float wf = s[n] * sw1[k] * w1[(size_t)n * L.cin + k];
          w1f[(size_t)n * L.cin + k] = wf;
          in_range = in_range && std::fabs(wf) <= 65504.0f;
          bias += (double)w1[(size_t)n * L.cin + k] * ((double)tc1[k] + (double)sw1[k] * (double)x0[k]);
          if (x0[k] != 0.f) {
            const float hi = (float)(f16)wf;
            const double weff = e->exact ? (double)hi + (double)(float)(f16)(wf - hi) : (double)hi;      // what the kernels multiply (split_hi_lo_rows / to_f16)
            corr += weff * (double)x0[k];
          }
        }
        t[n] = (float)((double)t[n] + (double)s[n] * bias - corr);
      }
      if (!in_range) { tn_set_error("a 1x1 weight leaves the fp16 range once its BatchNorm scales are folded in (" + sp + "conv" + std::to_string(2 * l) + ")"); return fail(TN_ERR_INVALID); }
      if (e->exact) {
        // k-tile of the kernel that will run the layer (the row pitch of [hi | lo] is twice K rounded up to it): the block's
        // fused kernel (dense_layer_big.hip: 64 channels at 14 x 14 and 7 x 7, 32 elsewhere), or conv1x1.hip's 64 where no fused
        // kernel tiles the map or holds the layer's K (encoder_run takes the same decision)
        const bool tile = dense_layer_supported(e->Hb[b], e->Wb[b]) && L.cin <= dense_layer_kmax(e->Wb[b]);
        const int bk = !tile ? 64 : (e->Hb[b] == 14 || e->Hb[b] == 7) ? 64 : 32;
        L.w1 = e->pool.upload(split_hi_lo_rows(w1f.data(), 128, L.cin, (L.cin + bk - 1) / bk * bk));
      } else {
        L.w1 = e->pool.upload(to_f16(w1f.data(), (size_t)128 * L.cin));
      }
      if (pack7) { h7[0].push_back(w1f); h7[3].push_back(t); }
      if (e->strip && dense_strip_supported(e->Hb[b], e->Wb[b], L.cin)) L.w1s = e->pool.upload(pack_w1_strip(w1f.data(), L.cin, t.data()));
      L.s2 = e->pool.upload(std::vector<float>(128, 1.0f)); L.t2 = e->pool.upload(t);
      if (e->exact) {       // packed image of hi, then packed image of lo
        std::vector<float> hi(32 * 128 * 9), lo(32 * 128 * 9);
        for (size_t i = 0; i < hi.size(); ++i) {
          hi[i] = (float)(f16)w3[i];
          lo[i] = w3[i] - hi[i];
        }
        std::vector<f16> img = pack_conv3x3(hi.data());
        const std::vector<f16> img_lo = pack_conv3x3(lo.data());
        img.insert(img.end(), img_lo.begin(), img_lo.end());
        L.w3p = e->pool.upload(img);
      } else {
        L.w3p = e->pool.upload(pack_conv3x3(w3));
      }
      if (L.w1s) L.w3s = e->pool.upload(pack_w3_strip(w3));
      e->layers[b].push_back(L);
    }
    if (pack14) {
      std::vector<Block14Layer> bl;
      for (int l = 0; l < kBlockCfg[b]; ++l)
        bl.push_back(Block14Layer{h7[0][l].data(), h7w3[l], h7[1][l].data(), h7[2][l].data(), h7[3][l].data()});
      DenseBlock14Args &a14 = e->b14[b];
      a14.stream = e->pool.upload(pack_block14(bl, e->Cin[b]));
      a14.total_units = dense_block14_units(e->Cin[b], kBlockCfg[b]);
      a14.ldc = e->Cb[b]; a14.K0 = e->Cin[b]; a14.nl = kBlockCfg[b];
    } else if (pack28) {
      std::vector<Block14Layer> bl;
      for (int l = 0; l < kBlockCfg[b]; ++l)
        bl.push_back(Block14Layer{h7[0][l].data(), h7w3[l], h7[1][l].data(), h7[2][l].data(), h7[3][l].data()});
      DenseBlock28Args &a28 = e->b28[b];
      a28.stream = e->pool.upload(pack_block28(bl, e->Cin[b]));
      a28.total_units = dense_block28_units(e->Cin[b], kBlockCfg[b]);
      a28.ldc = e->Cb[b]; a28.K0 = e->Cin[b]; a28.nl = kBlockCfg[b];
    } else if (pack7) {
      std::vector<Block7Layer> bl;
      for (int l = 0; l < kBlockCfg[b]; ++l)
        bl.push_back(Block7Layer{h7[0][l].data(), h7w3[l], h7[1][l].data(), h7[2][l].data(), h7[3][l].data()});
      const Block7Image img = pack_block7(bl, e->Cin[b]);
      DenseBlock7Args &a7 = e->b7[b];
      a7.wa = e->pool.upload(img.wa); a7.wb = e->pool.upload(img.wb); a7.tab = e->pool.upload(img.tab);
      for (int w = 0; w < 4; ++w) { a7.a_off[w] = img.a_off[w]; a7.b_off[w] = img.b_off[w]; }
      a7.ldc = e->Cb[b]; a7.K0 = e->Cin[b]; a7.nl = kBlockCfg[b];
    }
    {
      std::vector<DenseLayerDev> cd;
      for (auto &L : e->layers[b]) cd.push_back(DenseLayerDev{L.s1, L.t1, L.w1, L.s2, L.t2, L.w3p});
      e->chain_dev[b] = e->pool.upload(cd);
    }
    if (b < 3) {
      auto &T = e->trans[b];
      T.cin = e->Cb[b]; T.cout = e->Cb[b] / 2;
      const float *wt = pm.get(pre + "conv" + std::to_string(outer) + "_weight", (int64_t)T.cout * T.cin);
      if (!wt || !fold_bn(pm, pre + "batchnorm" + std::to_string(outer), T.cin, s, t)) return fail(TN_ERR_MISSING);
      if (b == 0) centre_shift(s, t);
      T.s = e->pool.upload(s); T.t = e->pool.upload(t);
      T.w = e->exact ? e->pool.upload(split_hi_lo_rows(wt, T.cout, T.cin, T.cin))
                     : e->pool.upload(to_f16(wt, (size_t)T.cout * T.cin));
      if (!e->exact && (T.cout == 512 || T.cout == 256) && T.cin % 128 == 0) {      // the warp-specialised kernel of the last two transitions (trans_ws.hip decides at launch)
        const std::vector<f16> wh = to_f16(wt, (size_t)T.cout * T.cin);
        T.wfrag = e->pool.upload(pack_trans_frags(wh.data(), T.cout, T.cin));
      }
      ++outer;
    }
  }
  if (!fold_bn(pm, pre + "batchnorm" + std::to_string(outer), e->Cb[3], s, t)) return fail(TN_ERR_MISSING);
  e->head_s = e->pool.upload(s); e->head_t = e->pool.upload(t);

  const size_t weights_bytes = e->pool.bytes;
  // Round 6: with pipelined forwards a batch is no longer cut in two halves that run side by side - consecutive WHOLE batches run
  // side by side on the two side streams (encoder_run), each in its own workspace set: twice the frames per launch
  // (bench.py --batch 512 measured what that is worth before it was built: +2.1 %, the per-launch drain / fill of the chip
  // amortised over two frames per CU).  TN_NO_INTERLEAVE: the half-batch form of rounds 1 - 5.
  e->interleave = getenv("TN_NO_INTERLEAVE") == nullptr && e->split && max_batch >= 64;
  const size_t B = (size_t)max_batch * (e->interleave ? 2 : 1);
  e->stem_out = (f16 *)e->pool.alloc(B * e->Hs * e->Ws * 64 * sizeof(f16));
  e->bott = (f16 *)e->pool.alloc(B * e->Hb[0] * e->Wb[0] * 128 * sizeof(f16));
  for (int b = 0; b < 4; ++b)
    e->blockbuf[b] = (f16 *)e->pool.alloc(B * e->Hb[b] * e->Wb[b] * e->Cb[b] * sizeof(f16));
  if (e->b7[3].wa) e->head32 = (float *)e->pool.alloc(B * e->Hb[3] * e->Wb[3] * e->Cb[3] * sizeof(float));
  for (int b = 0; b < 4; ++b)
    if (e->b14[b].stream) e->b14_scratch[b] = (f16 *)e->pool.alloc(B * dense_block14_scratch_halfs() * sizeof(f16));
  for (int b = 0; b < 4; ++b)
    if (e->b28[b].stream) e->b28_scratch[b] = (f16 *)e->pool.alloc(B * dense_block28_scratch_halfs() * sizeof(f16));
  e->workspace_bytes = e->pool.bytes - weights_bytes;
  if (e->pool.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  // dense_block14.hip reads the 32 channels a layer is about to write as the zero-weighted pad of its last 64-channel super-step:
  // whatever is there must be finite, so the buffer does not start as whatever the allocator left in it
  for (int b = 0; b < 4; ++b)
    if (e->b14[b].stream && (hipMemset(e->blockbuf[b], 0, B * e->Hb[b] * e->Wb[b] * e->Cb[b] * sizeof(f16)) != hipSuccess ||
                             hipMemset(e->b14_scratch[b], 0, B * dense_block14_scratch_halfs() * sizeof(f16)) != hipSuccess)) {
      tn_set_error("hipMemset failed");
      return fail(TN_ERR_HIP);
    }
  for (int b = 0; b < 4; ++b)      // (dense_block28.hip reads rows 28 .. 31 of a plane and the zero-weighted pad of a last super-step)
    if (e->b28[b].stream && (hipMemset(e->blockbuf[b], 0, B * e->Hb[b] * e->Wb[b] * e->Cb[b] * sizeof(f16)) != hipSuccess ||
                             hipMemset(e->b28_scratch[b], 0, B * dense_block28_scratch_halfs() * sizeof(f16)) != hipSuccess)) {
      tn_set_error("hipMemset failed");
      return fail(TN_ERR_HIP);
    }
  *out = e;
  return TN_OK;
}

extern "C" int tn_densenet121_feature_dim(const tn_encoder *enc) { return enc ? enc->feat_dim : 0; }
extern "C" size_t tn_densenet121_workspace_bytes(const tn_encoder *enc) { return enc ? enc->workspace_bytes : 0; }

// Frames [b0, b0+B) of the batch on stream s.  Every buffer is per-frame contiguous, so a
// sub-batch is just a pointer offset; weights are shared read-only.
static int encoder_run_range(tn_encoder *e, const void *x0, tn_layout layout, int b0, int B, float *feat0,
                             hipStream_t s, EventTimer &tm, int ws0 = -1) {
  int rc;
  const double fB = (double)B;
  const size_t frame_bytes = (size_t)e->H * e->W * 3 * (layout == TN_LAYOUT_NCHW_F32 ? 4 : layout == TN_LAYOUT_NHWC_F16 ? 2 : 1);
  const void *x = (const unsigned char *)x0 + (size_t)b0 * frame_bytes;
  float *feat = feat0 + (size_t)b0 * e->feat_dim;
  const int w0 = ws0 >= 0 ? ws0 : b0;        // first frame slot of the workspace (the second workspace set starts at maxB)
  f16 *stem_out = e->stem_out + (size_t)w0 * e->Hs * e->Ws * 64;
  f16 *bott = e->bott + (size_t)w0 * e->Hb[0] * e->Wb[0] * 128;
  f16 *bbuf[4];
  for (int b = 0; b < 4; ++b) bbuf[b] = e->blockbuf[b] + (size_t)w0 * e->Hb[b] * e->Wb[b] * e->Cb[b];
  // the head reads the last block un-rounded when the kernels that produce it write the fp32 side copy: the LDS-resident 7x7
  // block kernel and the transition in front of it (not in the layer-wise calibration pass, not with a tuning variant)
  float *h32 = (e->head32 && e->calib_dev == nullptr && e->dl_variant == 0)
                   ? e->head32 + (size_t)w0 * e->Hb[3] * e->Wb[3] * e->Cb[3] : nullptr;
  {
    StemArgs a{x, (int)layout, B, e->H, e->W, e->stem_wp, e->stem_wp_zf, e->stem_scale, e->stem_shift, stem_out, e->Hs, e->Ws};
    a.shift_u8 = e->stem_shift_u8;
    a.wp_zf_lo = e->stem_wp_zf_lo;
    a.floor = e->stem_floor;
    const double px = fB * e->Hs * e->Ws;
    if (e->fuse) {
      tm.begin("stem_conv_bn_relu_maxpool", 2.0 * px * 64 * 147, fB * e->H * e->W * 3 * 2 + fB * e->Hb[0] * e->Wb[0] * 64 * 2);
      rc = launch_stem_pool(a, bbuf[0], e->Cb[0], e->Hb[0], e->Wb[0], s);
      tm.end();
      if (rc) return rc;
    } else {
      tm.begin("stem_conv7x7_bn_relu", 2.0 * px * 64 * 147, fB * e->H * e->W * 3 * 2 + px * 64 * 2);
      rc = launch_stem(a, s);
      tm.end();
      if (rc) return rc;
      tm.begin("maxpool3x3s2", 0.0, px * 64 * 2 + fB * e->Hb[0] * e->Wb[0] * 64 * 2);
      rc = launch_maxpool3x3s2(stem_out, B, e->Hs, e->Ws, 64, bbuf[0], e->Cb[0], e->Hb[0], e->Wb[0], s);      // (the stem map is centred already: max commutes with the constant)
      tm.end();
      if (rc) return rc;
    }
  }
  // calibration pass (tn_densenet121_input_means): layer-wise kernels only, and behind every BatchNorm + ReLU that feeds a
  // convolution the per-channel mean of that input, in execution order
  const bool cal = e->calib_dev != nullptr;
  float *cal_out = e->calib_dev;
  auto cal_mean = [&](const f16 *xin, int ld, int K, const float *sc, const float *sh, long rows, int clamp = 0) {
    const int rc2 = launch_channel_mean(xin, ld, K, sc, sh, rows, e->calib_scratch, cal_out, s, clamp);
    cal_out += K;
    return rc2;
  };
  for (int b = 0; b < 4; ++b) {
    const int Hh = e->Hb[b], Ww = e->Wb[b];
    const int M = B * Hh * Ww;
    const bool fused = !cal && e->fuse && dense_layer_supported(Hh, Ww);
    if (!cal && e->b14[b].stream && e->dl_variant == 0) {
      // pixel-owning waves, all weights streamed through an LDS ring (dense_block14.hip)
      DenseBlock14Args a14 = e->b14[b];
      a14.buf = bbuf[b]; a14.B = B;
      a14.scratch = e->b14_scratch[b] + (size_t)w0 * dense_block14_scratch_halfs();
      double fl = 0, by = 0;
      for (auto &L : e->layers[b]) {
        fl += 2.0 * M * (128.0 * L.cin + 32.0 * 1152);
        by += (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2;
      }
      tm.begin("dense_block_stream_14x14", fl, by);
      rc = launch_dense_block14(a14, s);
      tm.end();
      if (rc) return rc;
    } else if (!cal && e->b28[b].stream && e->dl_variant == 0) {
      // the 28x28 block in four passes of eight rows, all weights streamed once per pass (dense_block28.hip)
      DenseBlock28Args a28 = e->b28[b];
      a28.buf = bbuf[b]; a28.B = B;
      a28.scratch = e->b28_scratch[b] + (size_t)w0 * dense_block28_scratch_halfs();
      double fl = 0, by = 0;
      for (auto &L : e->layers[b]) {
        fl += 2.0 * M * (128.0 * L.cin + 32.0 * 1152);
        by += (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2;
      }
      tm.begin("dense_block_stream_28x28", fl, by);
      rc = launch_dense_block28(a28, s);
      tm.end();
      if (rc) return rc;
    } else if (!cal && e->b7[b].wa && e->dl_variant == 0) {
      // the frame's concat buffer stays in LDS for the whole block; only the weights stream (dense_block7.hip)
      DenseBlock7Args a7 = e->b7[b];
      a7.buf = bbuf[b]; a7.B = B;
      a7.side = b == 3 ? h32 : nullptr;
      double fl = 0, by = 0;
      for (auto &L : e->layers[b]) {
        fl += 2.0 * M * (128.0 * L.cin + 32.0 * 1152);
        by += (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2;
      }
      tm.begin("dense_block_lds_7x7", fl, by);
      rc = launch_dense_block7(a7, s);
      tm.end();
      if (rc) return rc;
    } else if (fused && e->chain && (e->dl_variant & ~(32 | 64 | 128 | 256 | 512)) == 0 && Hh == Ww && (Hh == 14 || Hh == 7 || Hh == 16) &&
               e->layers[b].back().cin <= dense_layer_kmax(Ww)) {
      // one workgroup per frame walks the whole block: no launch gaps, no cold prologue per layer
      auto &L0 = e->layers[b][0];
      const int nl = (int)e->layers[b].size();
      DenseLayerArgs af{bbuf[b], e->Cb[b], L0.cin, L0.s1, L0.t1, L0.w1, L0.s2, L0.t2, L0.w3p, B, Hh, Ww, nullptr, e->dl_variant, e->chain_dev[b], nl};
      af.exact = e->exact;
      double fl = 0, by = 0;
      for (auto &L : e->layers[b]) {
        fl += 2.0 * M * (128.0 * L.cin + 32.0 * 1152);
        by += (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2;
      }
      tm.begin(Hh == 14 ? "dense_block_chained_14x14" : Hh == 16 ? "dense_block_chained_16x16" : "dense_block_chained_7x7", fl, by);
      rc = launch_dense_layer(af, s);
      tm.end();
      if (rc) return rc;
    } else
    for (auto &L : e->layers[b]) {
      // (one workgroup per 56 x 56 / 28 x 28 frame, 5 / 3 per 128 x 128 / 64 x 64 frame: enough of them to fill the chip?)
      if (e->fuse && L.w1s && B * (Hh == 128 ? 5 : Hh == 64 ? 3 : 1) >= e->strip_min_batch && e->dl_variant == 0 && !cal) {
        DenseStripArgs as{bbuf[b], e->Cb[b], L.cin, L.s1, L.t1, L.w1s, L.w3s, B, Hh, Ww};
        const std::string fam = "dense_layer_strip_" + std::to_string(Hh) + "x" + std::to_string(Ww);
        tm.begin(fam.c_str(), 2.0 * M * (128.0 * L.cin + 32.0 * 1152),
                 (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2);
        rc = launch_dense_strip(as, s);
        tm.end();
        if (rc) return rc;
        continue;
      }
      if (fused && L.cin <= dense_layer_kmax(Ww)) {
        DenseLayerArgs af{bbuf[b], e->Cb[b], L.cin, L.s1, L.t1, L.w1, L.s2, L.t2, L.w3p, B, Hh, Ww, nullptr, e->dl_variant};
        af.exact = e->exact;
        const std::string fam = "dense_layer_fused_" + std::to_string(Hh) + "x" + std::to_string(Ww);
        tm.begin(fam.c_str(), 2.0 * M * (128.0 * L.cin + 32.0 * 1152),
                 (double)M * (L.cin + 32) * 2 + 128.0 * L.cin * 2 + 32.0 * 1152 * 2);
        rc = launch_dense_layer(af, s);
        tm.end();
        if (rc) return rc;
        continue;
      }
      // un-fused: BN2 (scale folded into the weights) adds its shift in the 1x1's epilogue, the 3x3 only applies the ReLU
      // (the 1x1's operand is clamp(x, lo, hi): that is what its folded weights multiply, and what the calibration averages)
      if (cal && (rc = cal_mean(bbuf[b], e->Cb[b], L.cin, L.s1, L.t1, M, 1))) return rc;
      Conv1x1Args a1{bbuf[b], e->Cb[b], L.cin, L.s1, L.t1, L.w1, 128, bott, 128, 0, M, 0, Hh, Ww};
      a1.bias = L.t2;
      a1.clamp = 1;
      a1.exact = e->exact;
      tm.begin("conv1x1_bnrelu", 2.0 * M * 128.0 * L.cin, (double)M * (L.cin + 128) * 2 + 128.0 * L.cin * 2);
      rc = launch_conv1x1(a1, s);
      tm.end();
      if (rc) return rc;
      if (cal && (rc = cal_mean(bott, 128, 128, e->ones128, e->zeros128, M))) return rc;
      Conv3x3Args a3{bott, L.s2, e->zeros128, L.w3p, bbuf[b], e->Cb[b], L.cin, M, Hh, Ww};
      a3.exact = e->exact;
      tm.begin("conv3x3_bnrelu", 2.0 * M * 32.0 * 1152, (double)M * (128 + 32) * 2 + 32.0 * 1152 * 2);
      rc = launch_conv3x3(a3, s);
      tm.end();
      if (rc) return rc;
    }
    if (b < 3) {
      auto &T = e->trans[b];
      const int Mo = B * e->Hb[b + 1] * e->Wb[b + 1];
      if (cal && (rc = cal_mean(bbuf[b], e->Cb[b], T.cin, T.s, T.t, M))) return rc;
      Conv1x1Args at{bbuf[b], e->Cb[b], T.cin, T.s, T.t, T.w, T.cout, bbuf[b + 1], e->Cb[b + 1], 0, Mo, 1, Hh, Ww};
      at.exact = e->exact;
      at.wfrag = cal ? nullptr : T.wfrag;
      if (b == 2 && h32) { at.y32 = h32; at.ld32 = e->Cb[3]; }
      tm.begin("transition_conv1x1_avgpool", 2.0 * M * (double)T.cout * T.cin,
               (double)M * T.cin * 2 + (double)Mo * T.cout * 2 + (double)T.cout * T.cin * 2);
      rc = launch_conv1x1(at, s);
      tm.end();
      if (rc) return rc;
    }
  }
  tm.begin("head_bnrelu_avgpool7", 0.0, fB * e->Hb[3] * e->Wb[3] * e->Cb[3] * 2 + fB * e->feat_dim * 4);
  rc = launch_head(bbuf[3], B, e->Hb[3], e->Wb[3], e->Cb[3], e->head_s, e->head_t, feat, e->PH, e->PW, s, h32);
  tm.end();
  return rc;
}

static int encoder_run(tn_encoder *e, const void *x, tn_layout layout, int B, float *feat, EventTimer &tm) {
  TN_REQUIRE(e && x && feat, "tn_densenet121_forward: null argument");
  TN_REQUIRE(B > 0 && B <= e->maxB, "tn_densenet121_forward: batch exceeds max_batch");
  TN_ON_DEVICE(e->ctx->device);
  hipStream_t s = e->ctx->stream;
  e->last_batch = B;
  // Large batches run as two half-batches on two side streams: the halves drift apart, so one
  // half's load-bound kernels (56^2 block) overlap the other's MFMA-bound ones (measured +6%).
  // The caller's stream is fenced with events on both sides, so stream order is preserved.
  const int ns = e->nsplit;
  const bool split = e->split && !tm.on && B >= 32 * ns && (B % (8 * ns)) == 0;
  const int par = (int)(e->calls & 1);
  e->calls++;
  e->last_ws0 = 0;
  if (!split) {
    // (a pipelined encoder: earlier calls may still run on the side streams and share the workspace)
    if (e->pipelined) {
      for (int h = 0; h < e->split_of[par ^ 1]; ++h) TN_HIP_CHECK(hipStreamWaitEvent(s, e->ev_done[par ^ 1][h], 0));
      for (int h = 0; h < ns; ++h) TN_HIP_CHECK(hipStreamWaitEvent(s, e->ev_done[par][h], 0));   // (the call before that one)
    }
    e->split_of[par] = 0;
    return encoder_run_range(e, x, layout, 0, B, feat, s, tm);
  }
  if (e->pipelined && e->interleave && B / ns >= e->strip_min_batch) {      // (a batch whose halves would run the small-batch kernels keeps them: one kernel family per batch size, pipelined or not)
    // Whole batch on side stream `par`, workspace set `par`: the call before runs on the other stream in the other set, the
    // call before that was on this stream (stream order separates the two users of a set).  A call of the half-batch form
    // may still be in flight on either stream (the mode was switched, or a small batch came in between): wait for it.
    TN_HIP_CHECK(hipEventRecord(e->ev_in, s));
    TN_HIP_CHECK(hipStreamWaitEvent(e->side[par], e->ev_in, 0));
    if (!e->last_interleaved)
      for (int p2 = 0; p2 < 2; ++p2)
        for (int g = 0; g < e->split_of[p2]; ++g) TN_HIP_CHECK(hipStreamWaitEvent(e->side[par], e->ev_done[p2][g], 0));
    e->last_interleaved = true;
    e->last_ws0 = par * e->maxB;
    const int r = encoder_run_range(e, x, layout, 0, B, feat, e->side[par], tm, par * e->maxB);
    TN_HIP_CHECK(hipEventRecord(e->ev_done[par][0], e->side[par]));
    e->split_of[par] = 1;
    e->split_batch = 0;
    return r;
  }
  if (e->last_interleaved) {      // back to the half-batch form: its row ranges cut across both workspace sets' users
    for (int h = 0; h < ns; ++h)
      for (int p2 = 0; p2 < 2; ++p2)
        for (int g = 0; g < e->split_of[p2]; ++g) TN_HIP_CHECK(hipStreamWaitEvent(e->side[h], e->ev_done[p2][g], 0));
    e->last_interleaved = false;
  }
  TN_HIP_CHECK(hipEventRecord(e->ev_in, s));
  // Pipelined calls overlap on the side streams, and half h of every call works in workspace rows [h B / ns, (h + 1) B / ns):
  // equal batch sizes keep each row range on one stream, whose order then separates consecutive calls.  When the batch size
  // changes (a corpus' ragged last batch) the ranges shift across streams: every side stream first waits for everything the
  // earlier calls left on ANY side stream.
  if (e->pipelined && e->split_batch != 0 && e->split_batch != B) {
    for (int h = 0; h < ns; ++h)
      for (int p2 = 0; p2 < 2; ++p2)
        for (int g = 0; g < e->split_of[p2]; ++g) TN_HIP_CHECK(hipStreamWaitEvent(e->side[h], e->ev_done[p2][g], 0));
  }
  e->split_batch = B;
  int rc = TN_OK;
  for (int h = 0; h < ns; ++h) {
    TN_HIP_CHECK(hipStreamWaitEvent(e->side[h], e->ev_in, 0));
    const int r = encoder_run_range(e, x, layout, h * (B / ns), B / ns, feat, e->side[h], tm);
    if (r) rc = r;
    TN_HIP_CHECK(hipEventRecord(e->ev_done[par][h], e->side[h]));
    // pipelined: the join is the caller's (tn_densenet121_join), so that the next call's first half can start beside
    // the tail of this call's second half (the last chained block of a half batch runs on half of the CUs)
    if (!e->pipelined) TN_HIP_CHECK(hipStreamWaitEvent(s, e->ev_done[par][h], 0));
  }
  e->split_of[par] = ns;
  return rc;
}

static int encoder_join(tn_encoder *e, int lag) {
  if (e->calls - 1 - lag < 0) return TN_OK;
  const int par = (int)((e->calls - 1 - lag) & 1);
  for (int h = 0; h < e->split_of[par]; ++h) TN_HIP_CHECK(hipStreamWaitEvent(e->ctx->stream, e->ev_done[par][h], 0));
  return TN_OK;
}

extern "C" int tn_densenet121_set_pipelined(tn_encoder *enc, int on) {
  TN_REQUIRE(enc, "tn_densenet121_set_pipelined: null handle");
  TN_ON_DEVICE(enc->ctx->device);
  if (enc->pipelined && !on) {            // leaving the mode: everything issued so far is joined
    int rc = encoder_join(enc, 0);
    if (rc == TN_OK) rc = encoder_join(enc, 1);
    if (rc) return rc;
  }
  enc->pipelined = on != 0;
  return TN_OK;
}

extern "C" int tn_densenet121_join(tn_encoder *enc, int lag) {
  TN_REQUIRE(enc, "tn_densenet121_join: null handle");
  TN_REQUIRE(lag == 0 || lag == 1, "tn_densenet121_join: lag must be 0 (the last forward) or 1 (the one before)");
  TN_ON_DEVICE(enc->ctx->device);
  return encoder_join(enc, lag);
}

extern "C" int tn_densenet121_forward(tn_encoder *enc, const void *x, tn_layout layout, int batch, float *feat) {
  EventTimer tm;
  return encoder_run(enc, x, layout, batch, feat, tm);
}

extern "C" int tn_densenet121_profile(tn_encoder *enc, const void *x, tn_layout layout, int batch, float *feat,
                                      tn_kernel_stat *stats, int max_stats, int *n_stats) {
  TN_REQUIRE(enc && stats && n_stats, "tn_densenet121_profile: null argument");
  EventTimer tm;
  tm.on = true;
  tm.s = enc->ctx->stream;
  const int rc = encoder_run(enc, x, layout, batch, feat, tm);
  tm.finish();
  if (rc) return rc;
  const int n = (int)tm.fams.size() < max_stats ? (int)tm.fams.size() : max_stats;
  for (int i = 0; i < n; ++i) stats[i] = tm.fams[i];
  *n_stats = n;
  return TN_OK;
}

// Calibration statistics for weights.as_fp16_model(input_means=...): runs `batch` frames through the LAYER-WISE kernels and
// returns, for the 119 convolutions behind the stem in execution order (per dense layer: the 1x1's K inputs, then the 3x3's
// 128; a transition's inputs behind its block), the per-input-channel mean of the activation the convolution reads.
extern "C" int tn_densenet121_input_means(tn_encoder *e, const void *x, tn_layout layout, int batch, float *means_host,
                                          int64_t capacity, int64_t *numel) {
  TN_REQUIRE(e && x && means_host && numel, "tn_densenet121_input_means: null argument");
  TN_REQUIRE(batch > 0 && batch <= e->maxB, "tn_densenet121_input_means: batch exceeds max_batch");
  TN_REQUIRE(!e->exact, "tn_densenet121_input_means: not for an exact-weights encoder");
  TN_ON_DEVICE(e->ctx->device);
  int64_t n = 0;
  for (int b = 0; b < 4; ++b) {
    for (auto &L : e->layers[b]) n += L.cin + 128;
    if (b < 3) n += e->trans[b].cin;
  }
  *numel = n;
  TN_REQUIRE(capacity >= n, "tn_densenet121_input_means: host buffer too small");
  if (int rc = encoder_join(e, 0)) return rc;
  if (int rc = encoder_join(e, 1)) return rc;     // (the call before the last one may still run, in the workspace set this pass uses)
  hipStream_t s = e->ctx->stream;
  float *dev = nullptr, *feat = nullptr;
  double *scratch = nullptr;
  auto release = [&]() { (void)hipFree(dev); (void)hipFree(feat); (void)hipFree(scratch); e->calib_dev = nullptr; e->calib_scratch = nullptr; };
  if (hipMalloc((void **)&dev, sizeof(float) * n) != hipSuccess || hipMalloc((void **)&feat, sizeof(float) * (size_t)batch * e->feat_dim) != hipSuccess ||
      hipMalloc((void **)&scratch, sizeof(double) * 32 * 1024) != hipSuccess) {
    release();
    tn_set_error("tn_densenet121_input_means: device allocation failed");
    return TN_ERR_NOMEM;
  }
  e->calib_dev = dev; e->calib_scratch = scratch;
  EventTimer tm;
  int rc = encoder_run_range(e, x, layout, 0, batch, feat, s, tm);
  if (!rc && hipMemcpyAsync(means_host, dev, sizeof(float) * n, hipMemcpyDeviceToHost, s) != hipSuccess) rc = TN_ERR_HIP;
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = TN_ERR_HIP;
  if (!rc) {      // block 1's 1x1 operands are clamps of the CENTRED stem channels: hand out the means in the reference graph's units
    int64_t o = 0;
    for (auto &L : e->layers[0]) {
      for (int c = 0; c < 64; ++c) means_host[o + c] += e->stem_centre[c];
      o += L.cin + 128;
    }
  }
  release();
  if (rc == TN_ERR_HIP) tn_set_error("tn_densenet121_input_means: HIP error");
  e->last_batch = batch;
  return rc;
}

extern "C" int tn_densenet121_read_tap(tn_encoder *e, const char *tap_c, int batch, float *out_host, size_t capacity,
                                       size_t *numel) {
  TN_REQUIRE(e && tap_c && out_host && numel, "tn_densenet121_read_tap: null argument");
  TN_REQUIRE(batch > 0 && batch <= e->last_batch, "tn_densenet121_read_tap: batch exceeds the last forward");
  const std::string tap(tap_c);
  const f16 *src = nullptr;
  int hh = 0, ww = 0, cc = 0, ld = 0;
  if (tap == "stem") {
    TN_REQUIRE(!e->fuse, "read_tap: the stem map is not materialised when stem+maxpool are fused (use pool0)");
    src = e->stem_out; hh = e->Hs; ww = e->Ws; cc = 64; ld = 64;
  }
  else if (tap == "pool0") { src = e->blockbuf[0]; hh = e->Hb[0]; ww = e->Wb[0]; cc = 64; ld = e->Cb[0]; }
  else if (tap.rfind("stage", 0) == 0 && tap.size() == 6) {
    const int b = tap[5] - '1';
    TN_REQUIRE(b >= 0 && b < 4, "read_tap: bad stage");
    src = e->blockbuf[b]; hh = e->Hb[b]; ww = e->Wb[b]; cc = e->Cb[b]; ld = cc;
  } else if (tap.rfind("trans", 0) == 0 && tap.size() == 6) {
    const int b = tap[5] - '1';
    TN_REQUIRE(b >= 0 && b < 3, "read_tap: bad transition");
    src = e->blockbuf[b + 1]; hh = e->Hb[b + 1]; ww = e->Wb[b + 1]; cc = e->Cb[b] / 2; ld = e->Cb[b + 1];
  } else {
    TN_REQUIRE(false, "read_tap: unknown tap");
  }
  const size_t px = (size_t)batch * hh * ww;
  *numel = px * cc;
  TN_REQUIRE(capacity >= *numel, "read_tap: host buffer too small");
  src += (size_t)e->last_ws0 * hh * ww * ld;      // (the workspace set the last forward ran in)
  if (int rc = encoder_join(e, 0)) return rc;
  if (int rc = encoder_join(e, 1)) return rc;
  TN_HIP_CHECK(hipStreamSynchronize(e->ctx->stream));
  std::vector<f16> tmp(px * ld);
  TN_HIP_CHECK(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(f16), hipMemcpyDeviceToHost));
  // (the stem's output is stored centred: the tap hands out the values the reference's graph has)
  const bool centred = tap == "stem" || tap == "pool0" || tap == "stage1";
  for (size_t p = 0; p < px; ++p)
    for (int c = 0; c < cc; ++c) out_host[p * cc + c] = (float)tmp[p * ld + c] + (centred && c < 64 ? e->stem_centre[c] : 0.f);
  return TN_OK;
}

extern "C" int tn_densenet121_destroy(tn_encoder *enc) {
  if (!enc) return TN_OK;
  TnDeviceGuard tn_dg_(enc->ctx->device);
  for (int i = 0; i < 4; ++i) { (void)hipStreamSynchronize(enc->side[i]); (void)hipStreamDestroy(enc->side[i]); (void)hipEventDestroy(enc->ev_done[0][i]); (void)hipEventDestroy(enc->ev_done[1][i]); }
  (void)hipEventDestroy(enc->ev_in);
  enc->pool.release();
  delete enc;
  return TN_OK;
}

// ---- Dense -----------------------------------------------------------------------
struct tn_dense {
  tn_ctx *ctx;
  DevPool pool;
  float *w, *b;
  int units, in_units;
};

extern "C" int tn_dense_create(tn_ctx *ctx, const float *weight_host, const float *bias_host, int units,
                               int in_units, tn_dense **out) {
  TN_REQUIRE(ctx && weight_host && out, "tn_dense_create: null argument");
  TN_REQUIRE(units > 0 && in_units > 0, "tn_dense_create: bad shape");
  TN_ON_DEVICE(ctx->device);
  tn_dense *d = new tn_dense();
  d->ctx = ctx; d->units = units; d->in_units = in_units;
  d->w = d->pool.upload(std::vector<float>(weight_host, weight_host + (size_t)units * in_units));
  d->b = bias_host ? d->pool.upload(std::vector<float>(bias_host, bias_host + units)) : nullptr;
  if (d->pool.failed) { d->pool.release(); delete d; tn_set_error("device allocation failed"); return TN_ERR_NOMEM; }
  *out = d;
  return TN_OK;
}
extern "C" int tn_dense_forward(tn_dense *d, const float *x, int rows, float *y) {
  TN_REQUIRE(d && x && y, "tn_dense_forward: null argument");
  TN_REQUIRE(rows >= 0, "tn_dense_forward: negative rows");
  TN_ON_DEVICE(d->ctx->device);
  return launch_linear_f32(x, d->in_units, d->w, d->in_units, d->b, y, d->units, rows, d->units, d->in_units, 0,
                           d->ctx->stream);
}
extern "C" int tn_dense_destroy(tn_dense *d) {
  if (!d) return TN_OK;
  TnDeviceGuard tn_dg_(d->ctx->device);
  d->pool.release();
  delete d;
  return TN_OK;
}

// ---- bi-RNN ----------------------------------------------------------------------
struct tn_birnn {
  tn_ctx *ctx;
  DevPool pool;
  int gates, F, H, dirs, max_rows;
  float *wi;   // [dirs*G*H][F]   both directions stacked -> one i2h GEMM
  float *bi;   // [dirs*G*H]
  float *whT;  // [dirs][H][G*H]
  float *bh;   // [dirs][G*H]
  float *gi;   // workspace [max_rows][dirs*G*H]
};

extern "C" int tn_birnn_create(tn_ctx *ctx, tn_rnn_kind kind, int input_size, int hidden, const tn_param *params,
                               int n_params, const char *prefix_c, int bidirectional, int max_rows, tn_birnn **out) {
  TN_REQUIRE(ctx && params && prefix_c && out, "tn_birnn_create: null argument");
  TN_REQUIRE(kind == TN_RNN_GRU || kind == TN_RNN_LSTM, "tn_birnn_create: unknown cell kind");
  TN_REQUIRE(input_size > 0 && hidden > 0 && hidden % 4 == 0 && max_rows > 0, "tn_birnn_create: bad shape");
  const int G = kind == TN_RNN_GRU ? 3 : 4;
  TN_REQUIRE(G * hidden <= 1024, "tn_birnn_create: gates*hidden must be <= 1024");
  TN_ON_DEVICE(ctx->device);
  const std::string pre(prefix_c);
  ParamMap pm(params, n_params);
  const int dirs = bidirectional ? 2 : 1, GH = G * hidden;
  std::vector<float> wi((size_t)dirs * GH * input_size), bi((size_t)dirs * GH), whT((size_t)dirs * hidden * GH),
      bh((size_t)dirs * GH);
  for (int d = 0; d < dirs; ++d) {
    const std::string dp = pre + (d == 0 ? "l0_" : "r0_");
    const float *a = pm.get(dp + "i2h_weight", (int64_t)GH * input_size);
    const float *b = pm.get(dp + "h2h_weight", (int64_t)GH * hidden);
    const float *c = pm.get(dp + "i2h_bias", GH);
    const float *e = pm.get(dp + "h2h_bias", GH);
    if (!a || !b || !c || !e) return TN_ERR_MISSING;
    memcpy(&wi[(size_t)d * GH * input_size], a, sizeof(float) * GH * input_size);
    memcpy(&bi[(size_t)d * GH], c, sizeof(float) * GH);
    memcpy(&bh[(size_t)d * GH], e, sizeof(float) * GH);
    for (int j = 0; j < GH; ++j)
      for (int k = 0; k < hidden; ++k) whT[((size_t)d * hidden + k) * GH + j] = b[(size_t)j * hidden + k];
  }
  tn_birnn *r = new tn_birnn();
  r->ctx = ctx; r->gates = G; r->F = input_size; r->H = hidden; r->dirs = dirs; r->max_rows = max_rows;
  r->wi = r->pool.upload(wi); r->bi = r->pool.upload(bi); r->whT = r->pool.upload(whT); r->bh = r->pool.upload(bh);
  r->gi = (float *)r->pool.alloc((size_t)max_rows * dirs * GH * sizeof(float));
  if (r->pool.failed) {
    r->pool.release(); delete r; tn_set_error("device allocation failed"); return TN_ERR_NOMEM;
  }
  *out = r;
  return TN_OK;
}

extern "C" int tn_birnn_forward(tn_birnn *r, const float *x, int batch, int steps, const int32_t *valid_len,
                                float *seq, float *h_last, float *c_last) {
  TN_REQUIRE(r && x && seq, "tn_birnn_forward: null argument");
  TN_REQUIRE(batch > 0 && steps > 0 && (long)batch * steps <= r->max_rows, "tn_birnn_forward: B*T exceeds max_rows");
  TN_ON_DEVICE(r->ctx->device);
  hipStream_t s = r->ctx->stream;
  const int GH = r->gates * r->H, N = r->dirs * GH, rows = batch * steps;
  int rc = launch_linear_f32(x, r->F, r->wi, r->F, r->bi, r->gi, N, rows, N, r->F, 0, s);
  if (rc) return rc;
  if (valid_len) TN_HIP_CHECK(hipMemsetAsync(seq, 0, (size_t)rows * r->dirs * r->H * sizeof(float), s));
  return launch_rnn_recurrent(r->gates, r->gi, N, r->whT, r->bh, valid_len, seq, r->dirs * r->H, h_last, c_last,
                              batch, steps, r->H, r->dirs, s);
}
extern "C" int tn_birnn_destroy(tn_birnn *r) {
  if (!r) return TN_OK;
  TnDeviceGuard tn_dg_(r->ctx->device);
  r->pool.release();
  delete r;
  return TN_OK;
}

// ---- temporal-head training step -----------------------------------------------------
struct tn_head {
  tn_ctx *ctx;
  DevPool pool;
  int F, H, C, maxB, maxT;
  int G;                           // gates per cell: 3 GRU, 4 LSTM
  long n;                          // parameters in the flat buffers
  long o_wi, o_bi, o_wh, o_bh, o_wd, o_bd;
  float *w, *g, *mom;              // [n] parameters, gradients, momentum
  float *whT;                      // [2][H][G*H] transposed h2h for the forward recurrence
  float *gi, *seq, *gates, *pooled, *dlog, *dpool, *dseq, *dgi, *dgh, *hprev, *logits, *loss;
  int32_t *arg;
  std::string rnn_prefix, dense_prefix;
};

static int head_refresh_whT(tn_head *h) {
  for (int d = 0; d < 2; ++d) {
    const int rc = launch_transpose_f32(h->w + h->o_wh + (long)d * h->G * h->H * h->H, h->G * h->H, h->H,
                                        h->whT + (long)d * h->H * h->G * h->H, h->ctx->stream);
    if (rc) return rc;
  }
  return TN_OK;
}

extern "C" int tn_head_create(tn_ctx *ctx, tn_rnn_kind kind, int input_size, int hidden, int classes, const tn_param *params,
                              int n_params, const char *rnn_prefix, const char *dense_prefix, int max_batch,
                              int max_steps, tn_head **out) {
  TN_REQUIRE(ctx && params && rnn_prefix && dense_prefix && out, "tn_head_create: null argument");
  TN_REQUIRE(kind == TN_RNN_GRU || kind == TN_RNN_LSTM, "tn_head_create: type must be 'gru' or 'lstm'");
  const int G = kind == TN_RNN_GRU ? 3 : 4;
  TN_REQUIRE(input_size > 0 && hidden > 0 && hidden % 4 == 0 && G * hidden <= 1024 && classes > 0 && max_batch > 0 &&
                 max_steps > 0, "tn_head_create: bad shape (gates*hidden must be <= 1024, hidden % 4 == 0)");
  TN_ON_DEVICE(ctx->device);
  ParamMap pm(params, n_params);
  const int F = input_size, H = hidden, C = classes, GH = G * hidden;
  tn_head *h = new tn_head();
  h->ctx = ctx; h->G = G; h->F = F; h->H = H; h->C = C; h->maxB = max_batch; h->maxT = max_steps;
  h->rnn_prefix = rnn_prefix; h->dense_prefix = dense_prefix;
  h->o_wi = 0; h->o_bi = h->o_wi + 2L * GH * F; h->o_wh = h->o_bi + 2L * GH; h->o_bh = h->o_wh + 2L * GH * H;
  h->o_wd = h->o_bh + 2L * GH; h->o_bd = h->o_wd + (long)C * 2 * H; h->n = h->o_bd + C;
  std::vector<float> w(h->n);
  auto fail = [&](int rc) { h->pool.release(); delete h; return rc; };
  for (int d = 0; d < 2; ++d) {
    const std::string dp = h->rnn_prefix + (d == 0 ? "l0_" : "r0_");
    const float *a = pm.get(dp + "i2h_weight", (int64_t)GH * F), *b = pm.get(dp + "h2h_weight", (int64_t)GH * H);
    const float *c = pm.get(dp + "i2h_bias", GH), *e = pm.get(dp + "h2h_bias", GH);
    if (!a || !b || !c || !e) return fail(TN_ERR_MISSING);
    memcpy(&w[h->o_wi + (long)d * GH * F], a, sizeof(float) * GH * F);
    memcpy(&w[h->o_bi + (long)d * GH], c, sizeof(float) * GH);
    memcpy(&w[h->o_wh + (long)d * GH * H], b, sizeof(float) * GH * H);
    memcpy(&w[h->o_bh + (long)d * GH], e, sizeof(float) * GH);
  }
  const float *wd = pm.get(h->dense_prefix + "weight", (int64_t)C * 2 * H), *bd = pm.get(h->dense_prefix + "bias", C);
  if (!wd || !bd) return fail(TN_ERR_MISSING);
  memcpy(&w[h->o_wd], wd, sizeof(float) * C * 2 * H);
  memcpy(&w[h->o_bd], bd, sizeof(float) * C);
  h->w = h->pool.upload(w);
  const size_t rows = (size_t)max_batch * max_steps;
  auto fl = [&](size_t n) { return (float *)h->pool.alloc(n * sizeof(float)); };
  h->g = fl(h->n); h->mom = fl(h->n); h->whT = fl(2L * H * GH);
  h->gi = fl(rows * 2 * GH); h->seq = fl(rows * 2 * H); h->gates = fl(2 * rows * (G + 1) * H);
  h->pooled = fl((size_t)max_batch * 2 * H); h->dlog = fl((size_t)max_batch * C); h->dpool = fl((size_t)max_batch * 2 * H);
  h->dseq = fl(rows * 2 * H); h->dgi = fl(rows * 2 * GH); h->dgh = fl(rows * 2 * GH); h->hprev = fl(2 * rows * H);
  h->logits = fl((size_t)max_batch * C); h->loss = fl(max_batch);
  h->arg = (int32_t *)h->pool.alloc((size_t)max_batch * 2 * H * sizeof(int32_t));
  if (h->pool.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  TN_HIP_CHECK(hipMemsetAsync(h->mom, 0, h->n * sizeof(float), ctx->stream));
  TN_HIP_CHECK(hipMemsetAsync(h->g, 0, h->n * sizeof(float), ctx->stream));
  const int rc = head_refresh_whT(h);
  if (rc) return fail(rc);
  *out = h;
  return TN_OK;
}

extern "C" int tn_head_forward_backward(tn_head *h, const float *x, const int32_t *labels, int B, int T, float *loss,
                                        float *logits) {
  TN_REQUIRE(h && x && labels, "tn_head_forward_backward: null argument");
  TN_REQUIRE(B > 0 && B <= h->maxB && T > 0 && T <= h->maxT, "tn_head_forward_backward: batch / steps exceed the maxima");
  TN_ON_DEVICE(h->ctx->device);
  hipStream_t s = h->ctx->stream;
  const int F = h->F, H = h->H, C = h->C, GH = h->G * h->H, M = B * T;
  const bool lstm = h->G == 4;
  float *w = h->w, *g = h->g;
  int rc;
#define TN_TRY(e) do { rc = (e); if (rc) return rc; } while (0)
  // forward: one i2h GEMM for both directions, recurrence with saved gates, max over T (argmax kept), Dense
  TN_TRY(launch_linear_f32(x, F, w + h->o_wi, F, w + h->o_bi, h->gi, 2 * GH, M, 2 * GH, F, 0, s));
  TN_TRY(launch_rnn_recurrent(h->G, h->gi, 2 * GH, h->whT, w + h->o_bh, nullptr, h->seq, 2 * H, nullptr, nullptr, B, T, H, 2, s,
                              h->gates));
  TN_TRY(launch_pool_max_arg(h->seq, B, T, 2 * H, h->pooled, h->arg, s));
  TN_TRY(launch_linear_f32(h->pooled, 2 * H, w + h->o_wd, 2 * H, w + h->o_bd, h->logits, C, B, C, 2 * H, 0, s));
  TN_TRY(launch_softmax_ce(h->logits, labels, B, C, h->loss, h->dlog, s));
  // backward
  TN_TRY(launch_dense_bwd(h->dlog, h->pooled, w + h->o_wd, B, C, 2 * H, g + h->o_wd, g + h->o_bd, h->dpool, s));
  TN_TRY(launch_scatter_pool_grad(h->dpool, h->arg, B, T, 2 * H, h->dseq, s));
  if (lstm) TN_TRY(launch_lstm_train_bwd(h->seq, h->gates, h->dseq, w + h->o_wh, h->dgi, h->hprev, B, T, H, s));
  else TN_TRY(launch_gru_train_bwd(h->seq, h->gates, h->dseq, w + h->o_wh, h->dgi, h->dgh, h->hprev, B, T, H, s));
  const float *dgh = lstm ? h->dgi : h->dgh;   // LSTM: one pre-activation gradient feeds both branches
  TN_TRY(launch_gemm_tn_f32(h->dgi, 2 * GH, x, F, g + h->o_wi, F, 2 * GH, F, M, s));      // dW_ih = dGI^T X
  TN_TRY(launch_colsum_f32(h->dgi, 2 * GH, M, 2 * GH, g + h->o_bi, s));
  for (int d = 0; d < 2; ++d)                                                              // dW_hh = dGH^T H_prev
    TN_TRY(launch_gemm_tn_f32(dgh + d * GH, 2 * GH, h->hprev + (long)d * M * H, H, g + h->o_wh + (long)d * GH * H, H,
                              GH, H, M, s));
  TN_TRY(launch_colsum_f32(dgh, 2 * GH, M, 2 * GH, g + h->o_bh, s));
#undef TN_TRY
  if (loss) TN_HIP_CHECK(hipMemcpyAsync(loss, h->loss, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
  if (logits) TN_HIP_CHECK(hipMemcpyAsync(logits, h->logits, sizeof(float) * B * C, hipMemcpyDeviceToDevice, s));
  return TN_OK;
}

extern "C" int tn_head_buffers(tn_head *h, float **params_dev, float **grads_dev, int64_t *numel) {
  TN_REQUIRE(h, "tn_head_buffers: null handle");
  if (params_dev) *params_dev = h->w;
  if (grads_dev) *grads_dev = h->g;
  if (numel) *numel = h->n;
  return TN_OK;
}

extern "C" int tn_head_sgd_step(tn_head *h, float lr, float momentum, float wd, float rescale_grad) {
  TN_REQUIRE(h, "tn_head_sgd_step: null handle");
  TN_ON_DEVICE(h->ctx->device);
  int rc = launch_sgd_momentum(h->w, h->g, h->mom, h->n, lr, momentum, wd, rescale_grad, h->ctx->stream);
  if (rc) return rc;
  return head_refresh_whT(h);
}

extern "C" int tn_head_read_param(tn_head *h, const char *name_c, int gradient, float *out_host, int64_t capacity,
                                  int64_t *numel) {
  TN_REQUIRE(h && name_c && out_host && numel, "tn_head_read_param: null argument");
  const std::string name(name_c);
  const long GH = (long)h->G * h->H;
  long off = -1, cnt = 0;
  for (int d = 0; d < 2; ++d) {
    const std::string dp = h->rnn_prefix + (d == 0 ? "l0_" : "r0_");
    if (name == dp + "i2h_weight") { off = h->o_wi + d * GH * h->F; cnt = GH * h->F; }
    if (name == dp + "i2h_bias") { off = h->o_bi + d * GH; cnt = GH; }
    if (name == dp + "h2h_weight") { off = h->o_wh + d * GH * h->H; cnt = GH * h->H; }
    if (name == dp + "h2h_bias") { off = h->o_bh + d * GH; cnt = GH; }
  }
  if (name == h->dense_prefix + "weight") { off = h->o_wd; cnt = (long)h->C * 2 * h->H; }
  if (name == h->dense_prefix + "bias") { off = h->o_bd; cnt = h->C; }
  TN_REQUIRE(off >= 0, "tn_head_read_param: unknown parameter name");
  TN_REQUIRE(capacity >= cnt, "tn_head_read_param: host buffer too small");
  TN_ON_DEVICE(h->ctx->device);
  TN_HIP_CHECK(hipStreamSynchronize(h->ctx->stream));
  TN_HIP_CHECK(hipMemcpy(out_host, (gradient ? h->g : h->w) + off, sizeof(float) * cnt, hipMemcpyDeviceToHost));
  *numel = cnt;
  return TN_OK;
}

extern "C" int tn_head_destroy(tn_head *h) {
  if (!h) return TN_OK;
  TnDeviceGuard tn_dg_(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  h->pool.release();
  delete h;
  return TN_OK;
}

// ---- temporal pooling / PRF1 -------------------------------------------------------
extern "C" int tn_temporal_pool(tn_ctx *ctx, const float *x, int batch, int steps, int feat, tn_pool_kind kind,
                                float *y) {
  TN_REQUIRE(ctx && x && y, "tn_temporal_pool: null argument");
  TN_REQUIRE(batch > 0 && steps > 0 && feat > 0, "tn_temporal_pool: bad shape");
  TN_REQUIRE(kind == TN_POOL_MAX || kind == TN_POOL_MEAN, "tn_temporal_pool: unknown pool kind");
  TN_ON_DEVICE(ctx->device);
  return launch_temporal_pool(x, batch, steps, feat, (int)kind, y, ctx->stream);
}

extern "C" int tn_prf1_update(tn_ctx *ctx, const float *logits, const int32_t *labels, int rows, int classes,
                              int64_t *mat) {
  TN_REQUIRE(ctx && logits && labels && mat, "tn_prf1_update: null argument");
  TN_REQUIRE(rows >= 0 && classes > 0, "tn_prf1_update: bad shape");
  if (rows == 0) return TN_OK;
  TN_ON_DEVICE(ctx->device);
  return launch_prf1(logits, labels, rows, classes, mat, ctx->stream);
}

// Host side of the calibrated fp16 conversion (tennis_amd/calibrate.py, weights.as_fp16_model): the choice, weight by weight,
// between the two fp16 neighbours of an fp32 weight such that the rounding error of every output row is (nearly) orthogonal to
// the mean input activations of ALL calibration frames at once.  Pure host code (no GPU needed): 6.9 M weights x ~70 frames x
// a few sweeps is seconds here and minutes in numpy (weights._round_fp16_vector_feedback is the reference implementation the
// tests compare this against).
//
// The reference evaluates fp32 parameters (models/vision/definitions.py:27-33); this is what lets ONE fp16 number per weight
// stay within the path's 1e-3 bar of that evaluation (DESIGN.md "Numerics").
#include <cmath>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// one output row: minimises || A d ||^2 / F + ridge * sum_k (d[k] a_rms[k])^2 over d[k] in {d1[k], d2[k]} by a greedy pass along k
// and `sweeps` passes of coordinate descent (every weight re-decided against the residual of all the others)
void round_row(const float *w, int K, const std::vector<double> &An /*[K][F], A / sqrt(F)*/, const std::vector<double> &a2, int F, int sweeps,
               double ridge, float *out) {
  std::vector<double> d1(K), d2(K), r(F, 0.0);
  std::vector<float> lo(K), other(K);
  std::vector<char> use2(K, 0);
  for (int k = 0; k < K; ++k) {
    const f16 hk = (f16)w[k];                  // round to nearest even
    const float rt = (float)hk;
    lo[k] = rt;
    float ot = rt;
    if (w[k] != rt && std::isfinite(rt)) {      // the neighbour on the other side of w
      unsigned short bits = __builtin_bit_cast(unsigned short, hk);
      const bool up = w[k] > rt;
      if (rt == 0.0f) bits = up ? 0x0001 : 0x8001;
      else if ((rt > 0) == up) bits += 1;
      else bits -= 1;
      const float cand = (float)__builtin_bit_cast(f16, bits);
      if (std::isfinite(cand)) ot = cand;
    }
    other[k] = ot;
    d1[k] = (double)rt - (double)w[k];
    d2[k] = (double)ot - (double)w[k];
  }
  for (int sweep = 0; sweep <= sweeps; ++sweep)
    for (int k = 0; k < K; ++k) {
      const double *ak = &An[(size_t)k * F];
      if (sweep) {
        const double d = use2[k] ? d2[k] : d1[k];
        for (int f = 0; f < F; ++f) r[f] -= d * ak[f];
      }
      double ra = 0.0;
      for (int f = 0; f < F; ++f) ra += r[f] * ak[f];
      const double q = a2[k] * (1.0 + ridge);
      const double c1 = 2 * d1[k] * ra + d1[k] * d1[k] * q, c2 = 2 * d2[k] * ra + d2[k] * d2[k] * q;
      use2[k] = c2 < c1;
      const double d = use2[k] ? d2[k] : d1[k];
      for (int f = 0; f < F; ++f) r[f] += d * ak[f];
    }
  for (int k = 0; k < K; ++k) out[k] = use2[k] ? other[k] : lo[k];
}

}  // namespace

extern "C" int tn_round_fp16_calibrated(const float *w, int N, int K, const double *A, int F, int sweeps, double ridge, float *out) {
  TN_REQUIRE(w && A && out && N > 0 && K > 0 && F > 0 && sweeps >= 0 && ridge >= 0, "tn_round_fp16_calibrated: bad argument");
  std::vector<double> An((size_t)K * F), a2(K, 0.0);
  const double inv = 1.0 / std::sqrt((double)F);
  for (int f = 0; f < F; ++f)
    for (int k = 0; k < K; ++k) {
      const double v = A[(size_t)f * K + k] * inv;
      An[(size_t)k * F + f] = v;
      a2[k] += v * v;
    }
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
  if ((unsigned)N < nt) nt = (unsigned)N;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      for (int n = (int)t; n < N; n += (int)nt) round_row(w + (size_t)n * K, K, An, a2, F, sweeps, ridge, out + (size_t)n * K);
    });
  for (auto &x : th) x.join();
  return TN_OK;
}

// ---- BatchNorm + ReLU in front of a dense layer's 1x1 convolution as TWO packed-half instructions (round 4) ----
// The strip / streamed-block kernels are issue-bound: one wave per SIMD, and every 32-cycle MFMA of the 1x1 phase carried one
// BN1 + ReLU item of four VALU instructions (two v_fma_mix_f32, v_cvt_pk_f16_f32, v_pk_max_f16: fp32 constants).  With fp16
// constants the item is v_pk_fma_f16 + v_pk_max_f16 - measured 10 - 17 % on the layers with K >= 192 - but rounding s and t to
// fp16 is an error that is the same in every pixel (the average pool does not reduce it, like a weight-rounding error).  So
// the constants are not rounded, the layer is re-parametrised: for m > 0
//     relu(s x + t) = m relu(a x + b),   a = s / m,  b = t / m,
// and m goes into column k of the 1x1 weights BEFORE they are rounded to fp16 (where BN2's scale already goes).  m is chosen
// so that a is an fp16 number exactly (a = fp16(s) moved by j ulps, |j| <= 16, m = s / a) and b = t / m is as close to one as
// the 33 candidates allow (the fraction of b's ulp moves by ~ |b / a| per step: expected residual 1/33 of half an ulp).
// v_pk_fma_f16 is fused, so a x + b is rounded once - as the fp32 path rounded it.
void bn_scale_shift(const float *gamma, const float *beta, const float *mean, const float *var, int n, float eps, float *scale, float *shift) {
  for (int i = 0; i < n; ++i) {
    const float s = gamma[i] / std::sqrt(var[i] + eps);
    scale[i] = s;
    shift[i] = beta[i] - mean[i] * s;
  }
}

void bn_relu_fold_fp16(const float *scale, const float *shift, int n, float *a_out, float *b_out, float *m_out) {
  auto half_clamped = [](double v) {          // nearest fp16 number, +-65504 beyond the range, 0 for a NaN
    if (!(v == v)) return 0.f;
    if (v > 65504.0) return 65504.f;
    if (v < -65504.0) return -65504.f;
    return (float)(f16)(float)v;
  };
  for (int i = 0; i < n; ++i) {
    double s = scale[i], t = std::isfinite(shift[i]) ? (double)shift[i] : 0.0;
    if (!std::isfinite(scale[i]) || s == 0.0) {     // relu(t): a constant
      a_out[i] = 0.f; b_out[i] = half_clamped(t); m_out[i] = 1.f;
      continue;
    }
    // a scale outside fp16's comfortable range: a power of two of it goes into m first (exact)
    double m0 = 1.0;
    if (std::fabs(s) < 0x1p-10 || std::fabs(s) > 0x1p12) {
      int e;
      (void)std::frexp(s, &e);
      m0 = std::ldexp(1.0, e);
      s /= m0; t /= m0;
    }
    const f16 a0 = (f16)(float)s;
    const unsigned short bits0 = __builtin_bit_cast(unsigned short, a0);
    double best = 1e300;
    int bj = 0;
    for (int k = 0; k <= 32; ++k) {
      const int j = (k & 1) ? -((k + 1) >> 1) : (k >> 1);          // 0, -1, 1, -2, 2, ...: ties go to the smallest |j|
      const unsigned short bj_bits = (unsigned short)(bits0 + j);    // same sign, exponent stays normal (|s| in [2^-10, 2^12], |j| <= 16 < 1024)
      const double aj = (double)(float)__builtin_bit_cast(f16, bj_bits);
      const double bt = t * aj / s;
      const double err = std::fabs((double)half_clamped(bt) - bt);
      if (err < best) { best = err; bj = j; }
    }
    const unsigned short ab = (unsigned short)(bits0 + bj);
    const float a = (float)__builtin_bit_cast(f16, ab);
    a_out[i] = a;
    m_out[i] = (float)(m0 * s / (double)a);
    b_out[i] = half_clamped(t * (double)a / s);
  }
}

extern "C" int tn_bn_relu_fold_fp16(const float *gamma, const float *beta, const float *mean, const float *var, int n, float *a, float *b, float *m) {
  TN_REQUIRE(gamma && beta && mean && var && a && b && m && n > 0, "tn_bn_relu_fold_fp16: null argument");
  std::vector<float> s(n), t(n);
  bn_scale_shift(gamma, beta, mean, var, n, 1e-5f, s.data(), t.data());
  bn_relu_fold_fp16(s.data(), t.data(), n, a, b, m);
  return TN_OK;
}

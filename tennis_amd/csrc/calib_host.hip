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

// ---- BatchNorm + ReLU in front of a dense layer's 1x1 convolution WITHOUT a rounding (round 5) ----
// Rounds 1 - 4 evaluated relu(s x + t) per element and rounded the result to fp16 as the MFMA operand: one more rounding per
// consumed activation, and on flat image regions the SAME error in every pixel (scripts/round_study.py: this
// rounding alone is 9e-4 on a constant frame).  ReLU commutes with a positive scale:
//     s > 0:  relu(s x + t) = s max(x, c) + t,   c = -t / s         s < 0:  relu(s x + t) = s min(x, c) + t
// so the operand of the MFMA is u = clamp(x, lo, hi) - x itself (an fp16 number, stored by the producer) or the threshold: no
// arithmetic, no rounding, v_pk_max_f16 + v_pk_min_f16 - and the affine part leaves the element-wise path: s goes into column k
// of the 1x1 weights before they are rounded (where BN2's scale already goes), sum_k w[n][k] t[k] into BN2's shift.
// What is left of an error is the threshold's own rounding, c16 = fp16(c): pixels on the clipped side get s (c16 - c) instead
// of 0, at most 2^-11 |t| (the unclipped side is exact; scripts/round_study.py: 1.6e-4 on the features, every frame family).
//   out: lo, hi  fp16 numbers (as floats);  sw  the factor for column k of the weights;  tc  the constant of channel k
//   relu(scale x + shift) ~= sw * clamp(x, lo, hi) + tc
// Degenerate channels: scale 0 / not finite -> the constant relu(shift); a threshold beyond the fp16 range -> the channel is
// always clipped (constant 0) or never (lo = -65504 / hi = 65504: x is an fp16 number, the clamp does nothing).  A tiny scale
// (gamma 5e-6, beta 1) is nothing special here: sw = s, tc = t, the threshold far outside the range.
void bn_scale_shift(const float *gamma, const float *beta, const float *mean, const float *var, int n, float eps, float *scale, float *shift) {
  for (int i = 0; i < n; ++i) {
    const float s = gamma[i] / std::sqrt(var[i] + eps);
    scale[i] = s;
    shift[i] = beta[i] - mean[i] * s;
  }
}

void bn_relu_clamp_fold(const float *scale, const float *shift, int n, float *lo, float *hi, float *sw, float *tc) {
  constexpr double kMax = 65504.0;
  for (int i = 0; i < n; ++i) {
    const double s = scale[i], t = std::isfinite(shift[i]) ? (double)shift[i] : 0.0;
    auto constant = [&](double v) { lo[i] = 0.f; hi[i] = 0.f; sw[i] = 0.f; tc[i] = (float)v; };
    if (!std::isfinite(scale[i]) || s == 0.0) { constant(t > 0 ? t : 0.0); continue; }
    const double c = -t / s;
    if (s > 0) {
      if (c > kMax) { constant(0.0); continue; }                // never above the threshold
      lo[i] = (float)(f16)(float)(c < -kMax ? -kMax : c);
      hi[i] = (float)kMax;
    } else {
      if (c < -kMax) { constant(0.0); continue; }
      lo[i] = (float)-kMax;
      hi[i] = (float)(f16)(float)(c > kMax ? kMax : c);
    }
    sw[i] = scale[i];
    tc[i] = (float)t;
  }
}

extern "C" int tn_bn_relu_clamp_fold(const float *gamma, const float *beta, const float *mean, const float *var, int n, float *lo, float *hi,
                                     float *sw, float *tc) {
  TN_REQUIRE(gamma && beta && mean && var && lo && hi && sw && tc && n > 0, "tn_bn_relu_clamp_fold: null argument");
  std::vector<float> s(n), t(n);
  bn_scale_shift(gamma, beta, mean, var, n, 1e-5f, s.data(), t.data());
  bn_relu_clamp_fold(s.data(), t.data(), n, lo, hi, sw, tc);
  return TN_OK;
}

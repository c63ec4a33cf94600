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

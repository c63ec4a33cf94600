// Fused DenseNet dense layer, strip-streaming form: dispatch and host-side packing (the kernel is dense_strip_impl.h,
// instantiated per map width in dense_strip_w*.hip).
#include <array>
#include <vector>

#include "common.h"

int launch_dense_strip_w56(const DenseStripArgs &a, hipStream_t s);
int launch_dense_strip_w28(const DenseStripArgs &a, hipStream_t s);
int launch_dense_strip_w128(const DenseStripArgs &a, hipStream_t s);
int launch_dense_strip_w64(const DenseStripArgs &a, hipStream_t s);

bool dense_strip_supported(int H, int W, int K) {
  // (128 x 128: K <= 288 - the K = 320 instantiation of that width spills one register; a 512 x 512 input needs K <= 224 there)
  return H == W && (W == 56 || W == 28 || W == 128 || W == 64) && K % 32 == 0 && K >= 64 && K <= (W == 128 ? 288 : 320);
}

int launch_dense_strip(const DenseStripArgs &a, hipStream_t s) {
  TN_REQUIRE(dense_strip_supported(a.H, a.W, a.K), "dense_strip: unsupported geometry");
  TN_REQUIRE(a.buf && a.s1 && a.t1 && a.w1s && a.w3s, "dense_strip: null operand");
  TN_REQUIRE(a.ldc % 64 == 0 && a.K + 32 <= a.ldc, "dense_strip: bad channel geometry");
  if (a.W == 56) return launch_dense_strip_w56(a, s);
  if (a.W == 128) return launch_dense_strip_w128(a, s);
  if (a.W == 64) return launch_dense_strip_w64(a, s);
  return launch_dense_strip_w28(a, s);
}

// ---- host-side packing (api.hip, dbg.hip) ----
// 1x1 weights [128][K] (BN2's scale already folded in) + BN2's shift [128] -> v_mfma_f32_32x32x16_f16 A fragments
// [K/16 + 1 k-steps][4 blocks][64 lanes][8]: lane l: bottleneck channel 32 mb + (l & 31); input channels of 64-channel
// super-step u, k-step i: 64 u + 32 (l >> 5) + 8 i + j (a lane's four k-steps are 64 contiguous bytes of the pixel); the
// trailing 32-channel half super-step of an odd K/32: 64 u + 16 (l >> 5) + 8 i + j.  The last k-step multiplies the constant
// pixel fragment (1, 1, mask, 0, ...): columns fp16(shift), fp16(shift - fp16(shift)), 1.
std::vector<f16> pack_w1_strip(const float *w, int K, const float *shift) {
  const int ks = K / 32, kq = K / 16, nsu = (ks + 1) / 2;
  std::vector<f16> p((size_t)(kq + 1) * 4 * 64 * 8, (f16)0.f);
  for (int mb = 0; mb < 4; ++mb)
    for (int l = 0; l < 32; ++l) {       // (lanes 32 .. 63 hold k = 8 .. 15 of the shift step: zeros)
      f16 *d = &p[(((size_t)kq * 4 + mb) * 64 + l) * 8];
      const float t = shift[32 * mb + l];
      d[0] = (f16)t;
      d[1] = (f16)(t - (float)d[0]);
      d[2] = (f16)1.f;
    }
  for (int q = 0; q < kq; ++q) {
    const int u = q >> 2, i = q & 3;
    const bool half = (ks & 1) && u == nsu - 1;
    for (int mb = 0; mb < 4; ++mb)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int c = (half ? 64 * u + 16 * (l >> 5) : 64 * u + 32 * (l >> 5)) + 8 * i + j;
          p[(((size_t)q * 4 + mb) * 64 + l) * 8 + j] = (f16)w[(size_t)(32 * mb + (l & 31)) * K + c];
        }
  }
  return p;
}

// 3x3 weights (32,128,3,3) -> A fragments [3 dy][8 k-steps][3 dx][64 lanes][8]: lane l, m = l & 31: output channel
// 16 ((m >> 2) & 1) + (m & 3) + 4 (m >> 3) (so that a lane of the result holds 16 consecutive output channels of its pixel),
// bottleneck channel 16 t + 8 (j >> 2) + 4 (l >> 5) + (j & 3): the order in which the 1x1's accumulators hold them.
std::vector<f16> pack_w3_strip(const float *w) {
  constexpr int kW3Bytes = 3 * 8 * 3 * 1024;   // [dy][k16-step][dx] fragments of 1 KiB (dense_strip_impl.h)
  std::vector<f16> p((size_t)kW3Bytes / 2);
  for (int dy = 0; dy < 3; ++dy)
    for (int t = 0; t < 8; ++t)
      for (int dx = 0; dx < 3; ++dx)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int m = l & 31, o = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
            const int c = 16 * t + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            p[(((((size_t)dy * 8 + t) * 3 + dx) * 64) + l) * 8 + j] = (f16)w[(((size_t)o * 128 + c) * 3 + dy) * 3 + dx];
          }
  return p;
}

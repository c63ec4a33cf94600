// Memory-bound NHWC fp16 helpers of the frame encoder:
//   maxpool3x3s2  — stem MaxPool2D(3, 2, pad 1), writes straight into channels
//                   [0,64) of dense block 1's concat buffer;
//   head          — final BatchNorm + ReLU + AvgPool2D(7) + Flatten -> fp32
//                   feature rows in NCHW-flatten order (feature
//                   width 1024 @224, 4096 @512 — reference train.py:259).
// Both are the tail/head of gluoncv DenseNet .features (reference call site
// models/vision/definitions.py:30).  16-byte vector accesses, one 8-channel
// chunk per thread.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void maxpool_kernel(const f16 *__restrict__ x, int B, int H, int W, int C,
                                                      f16 *__restrict__ y, int ldy, int Ho, int Wo) {
  const int cpp = C >> 3;  // chunks per pixel
  const long total = (long)B * Ho * Wo * cpp;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(id % cpp);
    long p = id / cpp;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;  // MaxPool pads with -inf
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f16x8 v = *(const f16x8 *)(x + (((long)b * H + iy) * W + ix) * C + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)m[j];
    *(f16x8 *)(y + (((long)b * Ho + oy) * Wo + ox) * ldy + ch * 8) = o;
  }
}

// X32: the map comes un-rounded from the fp32 side buffer the last transition and the 7x7 block kernel write (round 5)
template <bool X32>
__global__ __launch_bounds__(256) void head_kernel(const f16 *__restrict__ x, const float *__restrict__ x32, int B, int H, int W, int C,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   float *__restrict__ feat, int PH, int PW) {
  const int cpp = C >> 3;
  const long total = (long)B * PH * PW * cpp;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int ch = (int)(id % cpp);
  long p = id / cpp;
  const int pw = (int)(p % PW);
  p /= PW;
  const int ph = (int)(p % PH);
  const int b = (int)(p / PH);
  float sc[8], sh[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[ch * 8 + j];
    sh[j] = shift[ch * 8 + j];
    acc[j] = 0.f;
  }
  for (int ky = 0; ky < 7; ++ky)
    for (int kx = 0; kx < 7; ++kx) {
      const int iy = ph * 7 + ky, ix = pw * 7 + kx;
      if constexpr (X32) {
        const float4 *q = (const float4 *)(x32 + (((long)b * H + iy) * W + ix) * C + ch * 8);
        const float4 v0 = q[0], v1 = q[1];
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
      } else {
        const f16x8 v = *(const f16x8 *)(x + (((long)b * H + iy) * W + ix) * C + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += fmaxf(fmaf((float)v[j], sc[j], sh[j]), 0.f);
      }
    }
  // Flatten of (B, C, PH, PW): index c*PH*PW + ph*PW + pw
  const int F = C * PH * PW;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    feat[(long)b * F + (long)(ch * 8 + j) * PH * PW + ph * PW + pw] = acc[j] * (1.0f / 49.0f);
}

}  // namespace

int launch_maxpool3x3s2(const f16 *x, int B, int H, int W, int C, f16 *y, int ldy, int Ho, int Wo, hipStream_t s) {
  TN_REQUIRE(C % 8 == 0 && ldy % 8 == 0, "maxpool: channels must be a multiple of 8");
  const long total = (long)B * Ho * Wo * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, C, y, ldy, Ho, Wo);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// Calibration statistic (tn_densenet121_input_means): per-channel mean over `rows` pixels of relu(scale * x + shift), i.e. the
// mean of what the next convolution reads.  Two deterministic stages: kMeanChunks row chunks x 64-channel groups of partial
// sums in double, then one pass over the chunks in index order.
constexpr int kMeanChunks = 32;
__global__ __launch_bounds__(256) void channel_mean_partial_kernel(const f16 *__restrict__ x, int ld, int K, const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, long rows, double *__restrict__ part, int clamp) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const long per = (rows + kMeanChunks - 1) / kMeanChunks, r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  double acc = 0.0;
  if (c < K) {
    const float sc = scale[c], sh = shift[c];
    // clamp: the operand of a fused dense layer's 1x1, clamp(x, lo = scale[c], hi = shift[c]) (calib_host.hip::bn_relu_clamp_fold)
    if (clamp) for (long r = r0 + sub; r < r1; r += 4) acc += (double)fminf(fmaxf((float)x[r * ld + c], sc), sh);
    else for (long r = r0 + sub; r < r1; r += 4) acc += (double)fmaxf(fmaf((float)x[r * ld + c], sc, sh), 0.f);
  }
  __shared__ double red[4][64];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && c < K) part[(long)blockIdx.y * K + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void channel_mean_final_kernel(const double *__restrict__ part, int K, long rows, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double a = 0.0;
  for (int j = 0; j < kMeanChunks; ++j) a += part[(long)j * K + c];
  out[c] = (float)(a / (double)rows);
}

int launch_channel_mean(const f16 *x, int ld, int K, const float *scale, const float *shift, long rows, double *scratch /* 32 * K */,
                        float *out, hipStream_t s, int clamp) {
  TN_REQUIRE(K > 0 && rows > 0, "channel_mean: empty input");
  hipLaunchKernelGGL(channel_mean_partial_kernel, dim3((K + 63) / 64, kMeanChunks), dim3(256), 0, s, x, ld, K, scale, shift, rows, scratch, clamp);
  hipLaunchKernelGGL(channel_mean_final_kernel, dim3((K + 255) / 256), dim3(256), 0, s, (const double *)scratch, K, rows, out);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_head(const f16 *x, int B, int H, int W, int C, const float *scale, const float *shift, float *feat,
                int PH, int PW, hipStream_t s, const float *x32) {
  TN_REQUIRE(C % 8 == 0, "head: channels must be a multiple of 8");
  const long total = (long)B * PH * PW * (C / 8);
  if (x32) hipLaunchKernelGGL(head_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x32, B, H, W, C, scale, shift, feat, PH, PW);
  else hipLaunchKernelGGL(head_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x32, B, H, W, C, scale, shift, feat, PH, PW);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// Fine-tuning step of the frame classifier (second half): FrameModel(DenseNet-121 .features, Dense(classes))
// trained end to end the way reference train.py drives it when the backbone is not frozen (model 0006 of
// models/README.md): BatchNorm in training mode (batch statistics, running statistics updated), SoftmaxCrossEntropyLoss
// per sample (:324), backward of the summed losses (:419-421), gluon.Trainer 'sgd' .step(batch_size) with momentum and
// weight decay (:298-299,424).  fp32 throughout, as MXNet trains by default.
//
// This is the CORRECT-FIRST version: every convolution is a GEMM on the exact-f32 matrix pipe (linear.hip / train.hip's
// transposed GEMM) over NHWC activations — 1x1 convolutions directly on the dense block's concat buffer, 3x3 and the 7x7
// stem through an explicit im2col buffer — with small element-wise / reduction kernels for BatchNorm, ReLU and the pools.
// The dense connectivity is what the inference path uses: one (B*H*W, C_total) buffer per block, a layer reads channels
// [0,K) and writes [K,K+32); the gradient buffer of a block has the same shape and every layer ACCUMULATES into [0,K).
// Activations that backward needs and that are cheap to rebuild (BN+ReLU outputs, im2col) are recomputed, the bottleneck
// convolution outputs and the batch statistics are kept.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "linear.h"
#include "train.h"

namespace {

constexpr float kEps = 1e-5f;        // gluon nn.BatchNorm(epsilon=1e-5)
constexpr float kBnMom = 0.9f;       // gluon nn.BatchNorm(momentum=0.9)

// ---- im2col ---------------------------------------------------------------------------------------------------------
// 7x7 stride 2 pad 3 on (B,H,W,3) -> (B*Ho*Wo, 147), column order (ky, kx, c)
__global__ void ft_im2col7_kernel(const float *__restrict__ x, int B, int H, int W, float *__restrict__ col) {
  const int Ho = H / 2, Wo = W / 2;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * Ho * Wo * 49) return;
  const int tap = (int)(id % 49);
  const long m = id / 49;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
  const int ky = tap / 7, kx = tap - ky * 7;
  const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
    const float *p = x + (((long)b * H + iy) * W + ix) * 3;
    v0 = p[0]; v1 = p[1]; v2 = p[2];
  }
  float *o = col + m * 147 + tap * 3;
  o[0] = v0; o[1] = v1; o[2] = v2;
}
// 3x3 pad 1 on (B,H,W,C) contiguous -> (B*H*W, 9*C), column order (ky, kx, c)
// sc / sh non-null: the source is a pre-activation, relu(a * sc[c] + sh[c]) is applied on the way (padding stays zero)
__global__ void ft_im2col3_kernel(const float *__restrict__ a, int B, int H, int W, int C, float *__restrict__ col,
                                  const float *__restrict__ sc = nullptr, const float *__restrict__ sh = nullptr) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C / 4;
  if (id >= (long)B * H * W * 9 * C4) return;
  const int c4 = (int)(id % C4);
  const int tap = (int)((id / C4) % 9);
  const long m = id / ((long)C4 * 9);
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long)W * H));
  const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
    v = *(const float4 *)(a + (((long)b * H + iy) * W + ix) * C + c4 * 4);
    if (sc) {
      const float4 s4 = *(const float4 *)(sc + c4 * 4), h4 = *(const float4 *)(sh + c4 * 4);
      v.x = fmaxf(fmaf(v.x, s4.x, h4.x), 0.f); v.y = fmaxf(fmaf(v.y, s4.y, h4.y), 0.f);
      v.z = fmaxf(fmaf(v.z, s4.z, h4.z), 0.f); v.w = fmaxf(fmaf(v.w, s4.w, h4.w), 0.f);
    }
  }
  *(float4 *)(col + m * 9 * C + tap * C + c4 * 4) = v;
}
// transpose of im2col3 (gather form, deterministic): da[m][c] = sum over taps of dcol[neighbour(m, tap)][tap][c]
__global__ void ft_col2im3_kernel(const float *__restrict__ dcol, int B, int H, int W, int C, float *__restrict__ da) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C / 4;
  if (id >= (long)B * H * W * C4) return;
  const int c4 = (int)(id % C4);
  const long m = id / C4;
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long)W * H));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // output pixel (oy, ox) read this pixel through tap (ky, kx) iff oy + ky - 1 == y, ox + kx - 1 == x
    const int oy = y - (tap / 3 - 1), ox = x - (tap % 3 - 1);
    if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
      const float4 v = *(const float4 *)(dcol + ((((long)b * H + oy) * W + ox) * 9 + tap) * C + c4 * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  *(float4 *)(da + m * C + c4 * 4) = acc;
}

// ---- BatchNorm (training mode) + ReLU ---------------------------------------------------------------------------------
// batch mean and biased variance of columns [0,C) of x (row stride ld).  Grid (C/64, RS): 64 columns x one of RS row slices
// per workgroup; one pass over the data with the column's first element as the shift (sums of (x - x0) and (x - x0)^2:
// no cancellation worth speaking of since x0 lies inside the data), slices added in order by the finish kernel.
__global__ __launch_bounds__(1024) void ft_bn_stats_kernel(const float *__restrict__ x, int ld, long M, int C,
                                                           float *__restrict__ part) {
  __shared__ float p1[16][64], p2[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const long stride = 16L * gridDim.y;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const float x0 = x[c];
    long r = (long)blockIdx.y * 16 + rg;
    for (; r + 15 * stride < M; r += 16 * stride) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = x[(r + i * stride) * ld + c] - x0;
#pragma unroll
      for (int i = 0; i < 16; ++i) { a1 += v[i]; a2 = fmaf(v[i], v[i], a2); }
    }
    for (; r < M; r += stride) { const float d = x[r * ld + c] - x0; a1 += d; a2 = fmaf(d, d, a2); }
  }
  p1[rg][cl] = a1; p2[rg][cl] = a2;
  __syncthreads();
  if (rg == 0 && c < C) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1 += p1[i][cl]; s2 += p2[i][cl]; }
    part[((long)blockIdx.y * 2 + 0) * C + c] = s1;
    part[((long)blockIdx.y * 2 + 1) * C + c] = s2;
  }
}
__global__ void ft_bn_stats_finish_kernel(const float *__restrict__ part, int RS, int C, const float *__restrict__ x, long M,
                                          float *__restrict__ mean, float *__restrict__ var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int z = 0; z < RS; ++z) { s1 += part[((long)z * 2 + 0) * C + c]; s2 += part[((long)z * 2 + 1) * C + c]; }
  const float d = s1 / (float)M;
  mean[c] = x[c] + d;
  var[c] = fmaxf(s2 / (float)M - d * d, 0.f);
}
// y (M,C contiguous) = relu(gamma * (x - mean) / sqrt(var + eps) + beta)
__global__ void ft_bn_relu_kernel(const float *__restrict__ x, int ld, long M, int C, const float *__restrict__ mean,
                                  const float *__restrict__ var, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, float *__restrict__ y) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= M * C) return;
  const long r = id / C;
  const int c = (int)(id - r * C);
  const float v = gamma[c] * (x[r * ld + c] - mean[c]) * rsqrtf(var[c] + kEps) + beta[c];
  y[id] = v > 0.f ? v : 0.f;
}
// column sums the BN backward needs: s1 = sum g, s2 = sum g * xhat with g = dy * [bn output > 0].  Grid (C/64, RS): 64
// columns x one of RS row slices per workgroup (16 row groups inside), partial sums to part[(slice*2 + {0,1})*C + c];
// ft_bn_bwd_finish_kernel adds the slices in order.
__global__ __launch_bounds__(1024) void ft_bn_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                int ld, long M, int C, const float *__restrict__ mean,
                                                                const float *__restrict__ var,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float *__restrict__ part) {
  __shared__ float p1[16][64], p2[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const long stride = 16L * gridDim.y;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const float m = mean[c], is = rsqrtf(var[c] + kEps), ga = gamma[c], be = beta[c];
    long r = (long)blockIdx.y * 16 + rg;
    for (; r + 7 * stride < M; r += 8 * stride) {
      float xv[8], dv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { xv[i] = x[(r + i * stride) * ld + c]; dv[i] = dy[(r + i * stride) * C + c]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[i] - m) * is;
        const float g = ga * xh + be > 0.f ? dv[i] : 0.f;
        a1 += g;
        a2 = fmaf(g, xh, a2);
      }
    }
    for (; r < M; r += stride) {
      const float xh = (x[r * ld + c] - m) * is;
      const float g = ga * xh + be > 0.f ? dy[r * C + c] : 0.f;
      a1 += g;
      a2 = fmaf(g, xh, a2);
    }
  }
  p1[rg][cl] = a1; p2[rg][cl] = a2;
  __syncthreads();
  if (rg == 0 && c < C) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1 += p1[i][cl]; s2 += p2[i][cl]; }
    part[((long)blockIdx.y * 2 + 0) * C + c] = s1;
    part[((long)blockIdx.y * 2 + 1) * C + c] = s2;
  }
}
__global__ void ft_bn_bwd_finish_kernel(const float *__restrict__ part, int RS, int C, float *__restrict__ dgamma,
                                        float *__restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int z = 0; z < RS; ++z) { s1 += part[((long)z * 2 + 0) * C + c]; s2 += part[((long)z * 2 + 1) * C + c]; }
  dbeta[c] = s1;
  dgamma[c] = s2;
}
// dx = gamma / sqrt(var + eps) * (g - dbeta / M - xhat * dgamma / M); assigned or accumulated into dx (row stride ldd)
__global__ void ft_bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x, int ld, long M, int C,
                                       const float *__restrict__ mean, const float *__restrict__ var,
                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                       const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                       float *__restrict__ dx, int ldd, int accumulate) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= M * C) return;
  const long r = id / C;
  const int c = (int)(id - r * C);
  const float is = rsqrtf(var[c] + kEps), xh = (x[r * ld + c] - mean[c]) * is;
  const float g = gamma[c] * xh + beta[c] > 0.f ? dy[id] : 0.f;
  const float v = gamma[c] * is * (g - dbeta[c] / (float)M - xh * dgamma[c] / (float)M);
  float *o = dx + r * ldd + c;
  *o = accumulate ? *o + v : v;
}
__global__ void ft_bn_running_kernel(float *__restrict__ rmean, float *__restrict__ rvar, const float *__restrict__ mean,
                                     const float *__restrict__ var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  rmean[c] = kBnMom * rmean[c] + (1.f - kBnMom) * mean[c];
  rvar[c] = kBnMom * rvar[c] + (1.f - kBnMom) * var[c];
}

// ---- pools ------------------------------------------------------------------------------------------------------------
// maxpool 3x3 stride 2 pad 1: a (B,H,W,C) -> y rows of stride ldy
__global__ void ft_maxpool_kernel(const float *__restrict__ a, int B, int H, int W, int C, float *__restrict__ y, int ldy) {
  const int Ho = H / 2, Wo = W / 2;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * Ho * Wo * C) return;
  const int c = (int)(id % C);
  const long m = id / C;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
  float best = -INFINITY;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) best = fmaxf(best, a[(((long)b * H + iy) * W + ix) * C + c]);
    }
  y[m * ldy + c] = best;
}
// gradient of the maxpool (first maximum of the window in scan order takes it), gather form over the <= 4 windows of a pixel
__global__ void ft_maxpool_bwd_kernel(const float *__restrict__ a, const float *__restrict__ dy, int ldy, int B, int H, int W,
                                      int C, float *__restrict__ da) {
  const int Ho = H / 2, Wo = W / 2;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * H * W * C) return;
  const int c = (int)(id % C);
  const long m = id / C;
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long)W * H));
  float acc = 0.f;
  for (int oy = (y + 1) / 2 - ((y + 1) % 2 == 0 ? 1 : 0); oy <= (y + 1) / 2; ++oy)
    for (int ox = (x + 1) / 2 - ((x + 1) % 2 == 0 ? 1 : 0); ox <= (x + 1) / 2; ++ox) {
      if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
      float best = -INFINITY;
      int by = -1, bx = -1;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float v = a[(((long)b * H + iy) * W + ix) * C + c];
            if (v > best) { best = v; by = iy; bx = ix; }
          }
        }
      if (by == y && bx == x) acc += dy[(((long)b * Ho + oy) * Wo + ox) * ldy + c];
    }
  da[id] = acc;
}
// avgpool 2x2 stride 2: z (B,H,W,C) -> y rows of stride ldy ; and its gradient (dy rows of stride ldy -> dz contiguous)
__global__ void ft_avgpool2_kernel(const float *__restrict__ z, int B, int H, int W, int C, float *__restrict__ y, int ldy) {
  const int Ho = H / 2, Wo = W / 2;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * Ho * Wo * C) return;
  const int c = (int)(id % C);
  const long m = id / C;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
  const float *p = z + (((long)b * H + 2 * oy) * W + 2 * ox) * C + c;
  y[m * ldy + c] = 0.25f * (p[0] + p[C] + p[(long)W * C] + p[(long)W * C + C]);
}
__global__ void ft_avgpool2_bwd_kernel(const float *__restrict__ dy, int ldy, int B, int H, int W, int C, float *__restrict__ dz) {
  const int Ho = H / 2, Wo = W / 2;
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * H * W * C) return;
  const int c = (int)(id % C);
  const long m = id / C;
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long)W * H));
  dz[id] = 0.25f * dy[(((long)b * Ho + y / 2) * Wo + x / 2) * ldy + c];
}
// global average pool over the P pixels of a frame: a (B,P,C) -> f (B,C); gradient: da = df / P
__global__ void ft_gap_kernel(const float *__restrict__ a, int B, int P, int C, float *__restrict__ f) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * C) return;
  const int b = id / C, c = id - b * C;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += a[((long)b * P + p) * C + c];
  f[id] = s / (float)P;
}
__global__ void ft_gap_bwd_kernel(const float *__restrict__ df, int B, int P, int C, float *__restrict__ da) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * P * C) return;
  const int c = (int)(id % C);
  const int b = (int)(id / ((long)P * C));
  da[id] = df[(long)b * C + c] / (float)P;
}

struct Pool {
  std::vector<void *> ptrs;
  bool failed = false;
  float *fl(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(float)) != hipSuccess) { failed = true; return nullptr; }
    ptrs.push_back(p);
    return (float *)p;
  }
  void release() { for (void *p : ptrs) (void)hipFree(p); ptrs.clear(); }
};

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// One BatchNorm: offsets of gamma / beta in the flat parameter buffer, of the running statistics and batch statistics
struct FtBn { long o_gamma, o_beta; long o_rm, o_rv; int C; float *mean, *var; std::string name; float *sc = nullptr, *sh = nullptr; };
// (sc, sh: the folded form relu(x * sc + sh) of the layer's training-mode BatchNorm + ReLU, refreshed in every forward - what the
//  GEMMs' operand transforms and the fused im2col read instead of a stored activation)
struct FtLayer { FtBn bn1, bn2; long o_w1, o_w3; int K; float *z1; std::string n1, n3; };
struct FtTrans { FtBn bn; long o_w; int Cin, Cout; float *z; std::string nw; };

struct tn_finetune {
  tn_ctx *ctx;
  Pool pool;
  int B, H, W, classes;
  std::string pre, cls;
  long n;                       // trainable parameters (GEMM layouts: conv weights as (Cout, ky*kx*Cin))
  long ns;                      // running statistics
  float *w, *g, *mom, *state;
  long o_w0, o_wd, o_bd;
  FtBn bn0, bnF;
  std::vector<FtLayer> layers[4];
  FtTrans trans[3];
  int Cin[4], Ctot[4], Hb[4];
  // activations kept for backward
  float *x_in, *col7, *z0, *a0, *X[4], *dX[4], *feat, *logits, *loss, *dlog, *dfeat;
  // batch statistics of every channel of a block's concat buffer, computed ONCE when the channel is produced: the BatchNorms in
  // front of the block's 1x1 convolutions, of its transition and of the head all normalise prefixes of the same channels
  float *Xmean[4], *Xvar[4];
  // temporaries
  float *ta, *tb, *col, *dcol, *tg, *tw, *ws;      // ws: split-K partial results / BatchNorm reduction slices
  long ws_floats;
  int32_t *labels;
};

// sc = gamma / sqrt(var + eps), sh = beta - mean * sc
__global__ void ft_bn_fold_kernel(const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, int C, float *__restrict__ sc, float *__restrict__ sh) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] * rsqrtf(var[c] + kEps);
  sc[c] = s;
  sh[c] = fmaf(-mean[c], s, beta[c]);
}
static int ft_slices(long M, int C) {
  const int cb = (C + 63) / 64;
  int RS = (int)((M + 2047) / 2048);                 // >= 2048 rows per slice, <= 512 workgroups
  if (RS > 512 / cb) RS = 512 / cb;
  return RS < 1 ? 1 : RS;
}
static void ft_bn_forward(tn_finetune *f, const FtBn &bn, const float *x, int ld, long M, float *y, hipStream_t s) {
  const int RS = ft_slices(M, bn.C);
  hipLaunchKernelGGL(ft_bn_stats_kernel, dim3((bn.C + 63) / 64, RS), dim3(1024), 0, s, x, ld, M, bn.C, f->ws);
  hipLaunchKernelGGL(ft_bn_stats_finish_kernel, dim3((bn.C + 255) / 256), dim3(256), 0, s, (const float *)f->ws, RS, bn.C, x, M, bn.mean,
                     bn.var);
  hipLaunchKernelGGL(ft_bn_relu_kernel, dim3(nblk(M * bn.C)), dim3(256), 0, s, x, ld, M, bn.C, (const float *)bn.mean,
                     (const float *)bn.var, (const float *)(f->w + bn.o_gamma), (const float *)(f->w + bn.o_beta), y);
}
// batch statistics of columns [0, C) of x into mean / var (no BatchNorm attached: the shared per-channel statistics of a block)
static void ft_stats(tn_finetune *f, const float *x, int ld, long M, int C, float *mean, float *var, hipStream_t s) {
  const int RS = ft_slices(M, C);
  hipLaunchKernelGGL(ft_bn_stats_kernel, dim3((C + 63) / 64, RS), dim3(1024), 0, s, x, ld, M, C, f->ws);
  hipLaunchKernelGGL(ft_bn_stats_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float *)f->ws, RS, C, x, M, mean, var);
}
// relu(x * sc + sh) of a BatchNorm whose batch statistics are known (bn.mean / bn.var)
static void ft_bn_fold(tn_finetune *f, const FtBn &bn, hipStream_t s) {
  hipLaunchKernelGGL(ft_bn_fold_kernel, dim3((bn.C + 255) / 256), dim3(256), 0, s, (const float *)bn.mean, (const float *)bn.var,
                     (const float *)(f->w + bn.o_gamma), (const float *)(f->w + bn.o_beta), bn.C, bn.sc, bn.sh);
}
static void ft_bn_recompute(tn_finetune *f, const FtBn &bn, const float *x, int ld, long M, float *y, hipStream_t s) {
  hipLaunchKernelGGL(ft_bn_relu_kernel, dim3(nblk(M * bn.C)), dim3(256), 0, s, x, ld, M, bn.C, (const float *)bn.mean,
                     (const float *)bn.var, (const float *)(f->w + bn.o_gamma), (const float *)(f->w + bn.o_beta), y);
}
// dy (M,C) contiguous -> gradients of gamma / beta into f->g and dx (stride ldd), assigned or accumulated
static void ft_bn_backward(tn_finetune *f, const FtBn &bn, const float *dy, const float *x, int ld, long M, float *dx, int ldd,
                           int accumulate, hipStream_t s) {
  const int cb = (bn.C + 63) / 64, RS = ft_slices(M, bn.C);
  hipLaunchKernelGGL(ft_bn_bwd_reduce_kernel, dim3(cb, RS), dim3(1024), 0, s, dy, x, ld, M, bn.C, (const float *)bn.mean,
                     (const float *)bn.var, (const float *)(f->w + bn.o_gamma), (const float *)(f->w + bn.o_beta), f->ws);
  hipLaunchKernelGGL(ft_bn_bwd_finish_kernel, dim3((bn.C + 255) / 256), dim3(256), 0, s, (const float *)f->ws, RS, bn.C,
                     f->g + bn.o_gamma, f->g + bn.o_beta);
  hipLaunchKernelGGL(ft_bn_bwd_apply_kernel, dim3(nblk(M * bn.C)), dim3(256), 0, s, dy, x, ld, M, bn.C, (const float *)bn.mean,
                     (const float *)bn.var, (const float *)(f->w + bn.o_gamma), (const float *)(f->w + bn.o_beta),
                     (const float *)(f->g + bn.o_gamma), (const float *)(f->g + bn.o_beta), dx, ldd, accumulate);
}

// conv weight (O, I, kh, kw) as Gluon stores it <-> the GEMM layout (O, kh*kw*I) used here
static void ft_reorder_in(const float *src, float *dst, int O, int I, int kh, int kw) {
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int y = 0; y < kh; ++y)
        for (int x = 0; x < kw; ++x) dst[((long)o * kh * kw + y * kw + x) * I + i] = src[(((long)o * I + i) * kh + y) * kw + x];
}
static void ft_reorder_out(const float *src, float *dst, int O, int I, int kh, int kw) {
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int y = 0; y < kh; ++y)
        for (int x = 0; x < kw; ++x) dst[(((long)o * I + i) * kh + y) * kw + x] = src[((long)o * kh * kw + y * kw + x) * I + i];
}

extern "C" int tn_finetune_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *backbone_prefix,
                                  const char *dense_prefix, int height, int width, int classes, int batch, tn_finetune **out) {
  TN_REQUIRE(ctx && params && backbone_prefix && dense_prefix && out, "tn_finetune_create: null argument");
  TN_REQUIRE(height > 0 && width > 0 && height % 32 == 0 && width % 32 == 0 && height == width && classes > 0 && batch > 0,
             "tn_finetune_create: frames must be square with a side divisible by 32");
  TN_ON_DEVICE(ctx->device);
  std::map<std::string, const tn_param *> pm;
  for (int i = 0; i < n_params; ++i) pm[params[i].name] = &params[i];
  tn_finetune *f = new tn_finetune();
  f->ctx = ctx; f->B = batch; f->H = height; f->W = width; f->classes = classes; f->pre = backbone_prefix; f->cls = dense_prefix;
  const std::string pre = f->pre;
  static const int kCfg[4] = {6, 12, 24, 16};
  long o = 0, os = 0;
  auto take = [&](long c) { const long r = o; o += c; return r; };
  auto mkbn = [&](const std::string &name, int C) {
    FtBn b; b.name = name; b.C = C; b.o_gamma = take(C); b.o_beta = take(C); b.o_rm = os; os += C; b.o_rv = os; os += C;
    b.mean = nullptr; b.var = nullptr;
    return b;
  };
  f->o_w0 = take(64 * 147);
  f->bn0 = mkbn(pre + "batchnorm0", 64);
  int c = 64, outer = 1;
  for (int b = 0; b < 4; ++b) {
    const std::string sp = pre + "stage" + std::to_string(b + 1) + "_";
    f->Cin[b] = c;
    for (int l = 0; l < kCfg[b]; ++l) {
      FtLayer L;
      L.K = c + 32 * l;
      L.bn1 = mkbn(sp + "batchnorm" + std::to_string(2 * l), L.K);
      L.n1 = sp + "conv" + std::to_string(2 * l) + "_weight"; L.o_w1 = take(128L * L.K);
      L.bn2 = mkbn(sp + "batchnorm" + std::to_string(2 * l + 1), 128);
      L.n3 = sp + "conv" + std::to_string(2 * l + 1) + "_weight"; L.o_w3 = take(32L * 1152);
      L.z1 = nullptr;
      f->layers[b].push_back(L);
    }
    c += 32 * kCfg[b];
    f->Ctot[b] = c;
    if (b < 3) {
      FtTrans &T = f->trans[b];
      T.Cin = c; T.Cout = c / 2;
      T.bn = mkbn(pre + "batchnorm" + std::to_string(outer), c);
      T.nw = pre + "conv" + std::to_string(outer) + "_weight"; T.o_w = take((long)T.Cout * T.Cin);
      T.z = nullptr;
      c /= 2;
      ++outer;
    }
  }
  f->bnF = mkbn(pre + "batchnorm" + std::to_string(outer), c);
  f->o_wd = take((long)classes * c); f->o_bd = take(classes);
  f->n = o; f->ns = os;
  std::vector<float> w(f->n), st(f->ns);
  auto fail = [&](int code) { f->pool.release(); delete f; return code; };
  bool ok = true;
  auto get = [&](const std::string &name, long cnt) -> const float * {
    auto it = pm.find(name);
    if (it == pm.end()) { tn_set_error("missing parameter: " + name); ok = false; return nullptr; }
    if (it->second->numel != cnt) { tn_set_error("parameter " + name + " has the wrong size"); ok = false; return nullptr; }
    return it->second->data_host;
  };
  auto loadbn = [&](const FtBn &b) {
    const float *ga = get(b.name + "_gamma", b.C), *be = get(b.name + "_beta", b.C), *rm = get(b.name + "_running_mean", b.C),
                *rv = get(b.name + "_running_var", b.C);
    if (!ga || !be || !rm || !rv) return;
    memcpy(&w[b.o_gamma], ga, sizeof(float) * b.C); memcpy(&w[b.o_beta], be, sizeof(float) * b.C);
    memcpy(&st[b.o_rm], rm, sizeof(float) * b.C); memcpy(&st[b.o_rv], rv, sizeof(float) * b.C);
  };
  if (const float *p0 = get(pre + "conv0_weight", 64 * 147)) ft_reorder_in(p0, &w[f->o_w0], 64, 3, 7, 7);
  loadbn(f->bn0);
  for (int b = 0; b < 4 && ok; ++b) {
    for (auto &L : f->layers[b]) {
      loadbn(L.bn1); loadbn(L.bn2);
      if (const float *p1 = get(L.n1, 128L * L.K)) memcpy(&w[L.o_w1], p1, sizeof(float) * 128 * L.K);
      if (const float *p3 = get(L.n3, 32L * 1152)) ft_reorder_in(p3, &w[L.o_w3], 32, 128, 3, 3);
    }
    if (b < 3) {
      loadbn(f->trans[b].bn);
      if (const float *pt = get(f->trans[b].nw, (long)f->trans[b].Cout * f->trans[b].Cin))
        memcpy(&w[f->trans[b].o_w], pt, sizeof(float) * f->trans[b].Cout * f->trans[b].Cin);
    }
  }
  loadbn(f->bnF);
  if (const float *pd = get(f->cls + "weight", (long)classes * c)) memcpy(&w[f->o_wd], pd, sizeof(float) * classes * c);
  if (const float *pb = get(f->cls + "bias", classes)) memcpy(&w[f->o_bd], pb, sizeof(float) * classes);
  if (!ok) return fail(TN_ERR_MISSING);
  // device buffers
  auto &P = f->pool;
  f->w = P.fl(f->n); f->g = P.fl(f->n); f->mom = P.fl(f->n); f->state = P.fl(f->ns);
  const long B = batch, M0 = B * (height / 2) * (width / 2);
  f->x_in = P.fl(B * height * width * 3); f->col7 = P.fl(M0 * 147); f->z0 = P.fl(M0 * 64); f->a0 = P.fl(M0 * 64);
  long maxMK = M0 * 64, maxM128 = 0;
  for (int b = 0; b < 4; ++b) {
    f->Hb[b] = height / (4 << b);
    const long M = B * f->Hb[b] * f->Hb[b];
    f->X[b] = P.fl(M * f->Ctot[b]); f->dX[b] = P.fl(M * f->Ctot[b]);
    for (auto &L : f->layers[b]) L.z1 = P.fl(M * 128);
    if (b < 3) f->trans[b].z = P.fl(M * f->trans[b].Cout);
    if (M * f->Ctot[b] > maxMK) maxMK = M * f->Ctot[b];
    if (M * 128 > maxM128) maxM128 = M * 128;
  }
  const long Mb0 = B * f->Hb[0] * f->Hb[0];
  f->ta = P.fl(maxMK); f->tb = P.fl(maxMK > maxM128 ? maxMK : maxM128); f->col = P.fl(Mb0 * 1152); f->dcol = P.fl(Mb0 * 1152);
  f->tg = P.fl(maxMK); f->tw = P.fl(1024L * 1024);
  f->ws_floats = 16L << 20; f->ws = P.fl(f->ws_floats);
  f->feat = P.fl(B * c); f->dfeat = P.fl(B * c); f->logits = P.fl(B * classes); f->loss = P.fl(B); f->dlog = P.fl(B * classes);
  {
    void *lp = nullptr;
    if (hipMalloc(&lp, sizeof(int32_t) * B) != hipSuccess) P.failed = true; else P.ptrs.push_back(lp);
    f->labels = (int32_t *)lp;
  }
  auto bnbuf = [&](FtBn &b) { b.mean = P.fl(b.C); b.var = P.fl(b.C); b.sc = P.fl(b.C); b.sh = P.fl(b.C); };
  // a BatchNorm over a prefix of a block's concat buffer reads the block's shared statistics
  auto bnshared = [&](FtBn &b, int blk) { b.mean = f->Xmean[blk]; b.var = f->Xvar[blk]; b.sc = P.fl(b.C); b.sh = P.fl(b.C); };
  for (int b = 0; b < 4; ++b) { f->Xmean[b] = P.fl(f->Ctot[b]); f->Xvar[b] = P.fl(f->Ctot[b]); }
  bnbuf(f->bn0); bnshared(f->bnF, 3);
  for (int b = 0; b < 4; ++b) {
    for (auto &L : f->layers[b]) { bnshared(L.bn1, b); bnbuf(L.bn2); }
    if (b < 3) bnshared(f->trans[b].bn, b);
  }
  if (P.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  TN_HIP_CHECK(hipMemcpy(f->w, w.data(), sizeof(float) * f->n, hipMemcpyHostToDevice));
  TN_HIP_CHECK(hipMemcpy(f->state, st.data(), sizeof(float) * f->ns, hipMemcpyHostToDevice));
  TN_HIP_CHECK(hipMemset(f->g, 0, sizeof(float) * f->n));
  TN_HIP_CHECK(hipMemset(f->mom, 0, sizeof(float) * f->n));
  *out = f;
  return TN_OK;
}

// x (batch, H, W, 3) fp32 normalised frames (NHWC), labels (batch,) int32, both DEVICE.  Runs the training-mode forward,
// the per-sample softmax cross-entropy and the backward of their SUM; loss (batch,) / logits (batch, classes) optional
// device outputs.  Gradients land in the flat buffer (tn_finetune_buffers); BatchNorm running statistics are updated.
extern "C" int tn_finetune_forward_backward(tn_finetune *f, const float *x, const int32_t *labels, int batch, int height, int width,
                                            float *loss, float *logits) {
  TN_REQUIRE(f && x && labels, "tn_finetune_forward_backward: null argument");
  TN_REQUIRE(height == f->H && width == f->W, "tn_finetune_forward_backward: the frame size must equal the handle's");
  TN_REQUIRE(batch == f->B, "tn_finetune_forward_backward: the batch must equal the handle's (BatchNorm statistics are per batch)");
  TN_ON_DEVICE(f->ctx->device);
  hipStream_t s = f->ctx->stream;
  const int B = f->B, H = f->H, W = f->W, NC = f->classes;
  const long M0 = (long)B * (H / 2) * (W / 2);
  float *w = f->w, *g = f->g;
  int rc;
#define TN_TRY(e) do { rc = (e); if (rc) return rc; } while (0)
  // ---------------- forward ----------------
  hipLaunchKernelGGL(ft_im2col7_kernel, dim3(nblk(M0 * 49)), dim3(256), 0, s, x, B, H, W, f->col7);
  TN_TRY(launch_linear_f32(f->col7, 147, w + f->o_w0, 147, nullptr, f->z0, 64, (int)M0, 64, 147, 0, s));
  ft_bn_forward(f, f->bn0, f->z0, 64, M0, f->a0, s);
  hipLaunchKernelGGL(ft_maxpool_kernel, dim3(nblk((long)B * f->Hb[0] * f->Hb[0] * 64)), dim3(256), 0, s, (const float *)f->a0, B, H / 2, W / 2, 64,
                     f->X[0], f->Ctot[0]);
  // Round 4: (a) a channel's batch statistics are computed once, when it is produced (64 / 32 / Cout new columns at a time) - every
  // BatchNorm of the block that normalises it reads them; (b) BatchNorm + ReLU in front of a convolution is never stored: the 1x1
  // GEMMs transform their X operand while staging it (launch_linear_f32_bnrelu), im2col transforms the bottleneck on the way
  ft_stats(f, f->X[0], f->Ctot[0], (long)B * f->Hb[0] * f->Hb[0], f->Cin[0], f->Xmean[0], f->Xvar[0], s);
  for (int b = 0; b < 4; ++b) {
    const int Hh = f->Hb[b], Ct = f->Ctot[b];
    const long M = (long)B * Hh * Hh;
    for (auto &L : f->layers[b]) {
      ft_bn_fold(f, L.bn1, s);
      TN_TRY(launch_linear_f32_bnrelu(f->X[b], Ct, L.bn1.sc, L.bn1.sh, w + L.o_w1, L.K, nullptr, L.z1, 128, (int)M, 128, L.K, 0, s));
      ft_stats(f, L.z1, 128, M, 128, L.bn2.mean, L.bn2.var, s);
      ft_bn_fold(f, L.bn2, s);
      hipLaunchKernelGGL(ft_im2col3_kernel, dim3(nblk(M * 9 * 32)), dim3(256), 0, s, (const float *)L.z1, B, Hh, Hh, 128, f->col,
                         (const float *)L.bn2.sc, (const float *)L.bn2.sh);
      TN_TRY(launch_linear_f32(f->col, 1152, w + L.o_w3, 1152, nullptr, f->X[b] + L.K, Ct, (int)M, 32, 1152, 0, s));
      ft_stats(f, f->X[b] + L.K, Ct, M, 32, f->Xmean[b] + L.K, f->Xvar[b] + L.K, s);
    }
    if (b < 3) {
      FtTrans &T = f->trans[b];
      ft_bn_fold(f, T.bn, s);
      TN_TRY(launch_linear_f32_bnrelu(f->X[b], Ct, T.bn.sc, T.bn.sh, w + T.o_w, T.Cin, nullptr, T.z, T.Cout, (int)M, T.Cout, T.Cin, 0, s));
      hipLaunchKernelGGL(ft_avgpool2_kernel, dim3(nblk(M / 4 * T.Cout)), dim3(256), 0, s, (const float *)T.z, B, Hh, Hh, T.Cout, f->X[b + 1],
                         f->Ctot[b + 1]);
      ft_stats(f, f->X[b + 1], f->Ctot[b + 1], M / 4, T.Cout, f->Xmean[b + 1], f->Xvar[b + 1], s);
    }
  }
  const int CF = f->Ctot[3], P3 = f->Hb[3] * f->Hb[3];
  const long M3 = (long)B * P3;
  ft_bn_recompute(f, f->bnF, f->X[3], CF, M3, f->ta, s);
  hipLaunchKernelGGL(ft_gap_kernel, dim3(nblk((long)B * CF)), dim3(256), 0, s, (const float *)f->ta, B, P3, CF, f->feat);
  TN_TRY(launch_linear_f32(f->feat, CF, w + f->o_wd, CF, w + f->o_bd, f->logits, NC, B, NC, CF, 0, s));
  TN_HIP_CHECK(hipMemcpyAsync(f->labels, labels, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, s));
  TN_TRY(launch_softmax_ce(f->logits, f->labels, B, NC, f->loss, f->dlog, s));
  if (loss) TN_HIP_CHECK(hipMemcpyAsync(loss, f->loss, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
  if (logits) TN_HIP_CHECK(hipMemcpyAsync(logits, f->logits, sizeof(float) * B * NC, hipMemcpyDeviceToDevice, s));
  // ---------------- backward ----------------
  TN_TRY(launch_dense_bwd(f->dlog, f->feat, w + f->o_wd, B, NC, CF, g + f->o_wd, g + f->o_bd, f->dfeat, s));
  hipLaunchKernelGGL(ft_gap_bwd_kernel, dim3(nblk(M3 * CF)), dim3(256), 0, s, (const float *)f->dfeat, B, P3, CF, f->tg);
  ft_bn_backward(f, f->bnF, f->tg, f->X[3], CF, M3, f->dX[3], CF, 0, s);
  for (int b = 3; b >= 0; --b) {
    const int Hh = f->Hb[b], Ct = f->Ctot[b];
    const long M = (long)B * Hh * Hh;
    for (int l = (int)f->layers[b].size() - 1; l >= 0; --l) {
      FtLayer &L = f->layers[b][l];
      const float *dy = f->dX[b] + L.K;                      // (M, 32) view, row stride Ct
      // 3x3: dW3 = dy^T col ; dcol = dy W3 ; col2im
      hipLaunchKernelGGL(ft_im2col3_kernel, dim3(nblk(M * 9 * 32)), dim3(256), 0, s, (const float *)L.z1, B, Hh, Hh, 128, f->col,
                         (const float *)L.bn2.sc, (const float *)L.bn2.sh);
      TN_TRY(launch_gemm_tn_f32(dy, Ct, f->col, 1152, g + L.o_w3, 1152, 32, 1152, (int)M, s, f->ws, f->ws_floats));
      TN_TRY(launch_transpose_f32(w + L.o_w3, 32, 1152, f->tw, s));                      // (1152, 32)
      TN_TRY(launch_linear_f32(dy, Ct, f->tw, 32, nullptr, f->dcol, 1152, (int)M, 1152, 32, 0, s));
      hipLaunchKernelGGL(ft_col2im3_kernel, dim3(nblk(M * 32)), dim3(256), 0, s, (const float *)f->dcol, B, Hh, Hh, 128, f->tg);
      ft_bn_backward(f, L.bn2, f->tg, L.z1, 128, M, f->tb, 128, 0, s);                  // tb = d z1
      // 1x1: dW1 = dz1^T a ; da = dz1 W1
      TN_TRY(launch_gemm_tn_f32_bnrelu(f->tb, 128, f->X[b], Ct, L.bn1.sc, L.bn1.sh, g + L.o_w1, L.K, 128, L.K, (int)M, s, f->ws, f->ws_floats));
      TN_TRY(launch_transpose_f32(w + L.o_w1, 128, L.K, f->tw, s));                     // (K, 128)
      TN_TRY(launch_linear_f32(f->tb, 128, f->tw, 128, nullptr, f->tg, L.K, (int)M, L.K, 128, 0, s));
      ft_bn_backward(f, L.bn1, f->tg, f->X[b], Ct, M, f->dX[b], Ct, 1, s);               // accumulate into channels [0, K)
    }
    if (b > 0) {
      FtTrans &T = f->trans[b - 1];
      const int Hp = f->Hb[b - 1], Cp = f->Ctot[b - 1];
      const long Mp = (long)B * Hp * Hp;
      hipLaunchKernelGGL(ft_avgpool2_bwd_kernel, dim3(nblk(Mp * T.Cout)), dim3(256), 0, s, (const float *)f->dX[b], Ct, B, Hp, Hp, T.Cout, f->tb);
      TN_TRY(launch_gemm_tn_f32_bnrelu(f->tb, T.Cout, f->X[b - 1], Cp, T.bn.sc, T.bn.sh, g + T.o_w, T.Cin, T.Cout, T.Cin, (int)Mp, s, f->ws,
                                       f->ws_floats));
      TN_TRY(launch_transpose_f32(w + T.o_w, T.Cout, T.Cin, f->tw, s));                  // (Cin, Cout)
      TN_TRY(launch_linear_f32(f->tb, T.Cout, f->tw, T.Cout, nullptr, f->tg, T.Cin, (int)Mp, T.Cin, T.Cout, 0, s));
      ft_bn_backward(f, T.bn, f->tg, f->X[b - 1], Cp, Mp, f->dX[b - 1], Cp, 0, s);
    } else {
      // stem: maxpool -> BN+ReLU -> conv 7x7 (its input gradient is not needed)
      ft_bn_recompute(f, f->bn0, f->z0, 64, M0, f->a0, s);
      hipLaunchKernelGGL(ft_maxpool_bwd_kernel, dim3(nblk(M0 * 64)), dim3(256), 0, s, (const float *)f->a0, (const float *)f->dX[0], Ct, B, H / 2,
                         W / 2, 64, f->tg);
      ft_bn_backward(f, f->bn0, f->tg, f->z0, 64, M0, f->tb, 64, 0, s);
      TN_TRY(launch_gemm_tn_f32(f->tb, 64, f->col7, 147, g + f->o_w0, 147, 64, 147, (int)M0, s, f->ws, f->ws_floats));
    }
  }
  // ---------------- BatchNorm running statistics ----------------
  auto upd = [&](const FtBn &b) {
    hipLaunchKernelGGL(ft_bn_running_kernel, dim3((b.C + 255) / 256), dim3(256), 0, s, f->state + b.o_rm, f->state + b.o_rv,
                       (const float *)b.mean, (const float *)b.var, b.C);
  };
  upd(f->bn0); upd(f->bnF);
  for (int b = 0; b < 4; ++b) {
    for (auto &L : f->layers[b]) { upd(L.bn1); upd(L.bn2); }
    if (b < 3) upd(f->trans[b].bn);
  }
#undef TN_TRY
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

extern "C" int tn_finetune_buffers(tn_finetune *f, float **params_dev, float **grads_dev, int64_t *numel) {
  TN_REQUIRE(f, "tn_finetune_buffers: null handle");
  if (params_dev) *params_dev = f->w;
  if (grads_dev) *grads_dev = f->g;
  if (numel) *numel = f->n;
  return TN_OK;
}

extern "C" int tn_finetune_sgd_step(tn_finetune *f, float lr, float momentum, float wd, float rescale_grad) {
  TN_REQUIRE(f, "tn_finetune_sgd_step: null handle");
  TN_ON_DEVICE(f->ctx->device);
  return launch_sgd_momentum(f->w, f->g, f->mom, f->n, lr, momentum, wd, rescale_grad, f->ctx->stream);
}

// Gluon-named parameter (conv weights back in (O, I, kh, kw) order), its gradient (gradient = 1), a running statistic, or
// a BatchNorm's batch statistic of the last step ("<bn>_batch_mean" / "<bn>_batch_var", test hook)
extern "C" int tn_finetune_read_param(tn_finetune *f, const char *name_c, int gradient, float *out_host, int64_t capacity,
                                      int64_t *numel) {
  TN_REQUIRE(f && name_c && out_host && numel, "tn_finetune_read_param: null argument");
  const std::string name(name_c);
  TN_ON_DEVICE(f->ctx->device);
  TN_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
  const float *base = gradient ? f->g : f->w;
  auto copy = [&](const float *dev, long cnt) -> int {
    TN_REQUIRE(capacity >= cnt, "tn_finetune_read_param: host buffer too small");
    TN_HIP_CHECK(hipMemcpy(out_host, dev, sizeof(float) * cnt, hipMemcpyDeviceToHost));
    *numel = cnt;
    return TN_OK;
  };
  auto conv = [&](long off, int O, int I, int kh, int kw) -> int {
    const long cnt = (long)O * I * kh * kw;
    TN_REQUIRE(capacity >= cnt, "tn_finetune_read_param: host buffer too small");
    std::vector<float> tmp(cnt);
    TN_HIP_CHECK(hipMemcpy(tmp.data(), base + off, sizeof(float) * cnt, hipMemcpyDeviceToHost));
    ft_reorder_out(tmp.data(), out_host, O, I, kh, kw);
    *numel = cnt;
    return TN_OK;
  };
  auto bn = [&](const FtBn &b, int &rcode) -> bool {
    if (name == b.name + "_gamma") { rcode = copy(base + b.o_gamma, b.C); return true; }
    if (name == b.name + "_beta") { rcode = copy(base + b.o_beta, b.C); return true; }
    if (name == b.name + "_running_mean") { rcode = copy(f->state + b.o_rm, b.C); return true; }
    if (name == b.name + "_running_var") { rcode = copy(f->state + b.o_rv, b.C); return true; }
    if (name == b.name + "_batch_mean") { rcode = copy(b.mean, b.C); return true; }
    if (name == b.name + "_batch_var") { rcode = copy(b.var, b.C); return true; }
    return false;
  };
  int rcode = TN_OK;
  if (name == f->pre + "conv0_weight") return conv(f->o_w0, 64, 3, 7, 7);
  if (bn(f->bn0, rcode) || bn(f->bnF, rcode)) return rcode;
  for (int b = 0; b < 4; ++b) {
    for (auto &L : f->layers[b]) {
      if (bn(L.bn1, rcode) || bn(L.bn2, rcode)) return rcode;
      if (name == L.n1) return copy(base + L.o_w1, 128L * L.K);
      if (name == L.n3) return conv(L.o_w3, 32, 128, 3, 3);
    }
    if (b < 3) {
      if (bn(f->trans[b].bn, rcode)) return rcode;
      if (name == f->trans[b].nw) return copy(base + f->trans[b].o_w, (long)f->trans[b].Cout * f->trans[b].Cin);
    }
  }
  if (name == f->cls + "weight") return copy(base + f->o_wd, (long)f->classes * f->Ctot[3]);
  if (name == f->cls + "bias") return copy(base + f->o_bd, f->classes);
  TN_REQUIRE(false, "tn_finetune_read_param: unknown parameter name");
}

extern "C" int tn_finetune_destroy(tn_finetune *f) {
  if (!f) return TN_OK;
  TnDeviceGuard tn_dg_(f->ctx->device);
  (void)hipStreamSynchronize(f->ctx->stream);
  f->pool.release();
  delete f;
  return TN_OK;
}

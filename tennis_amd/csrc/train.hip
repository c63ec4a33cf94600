// Training step of the temporal head: bi-GRU(hidden) over (B,T,F) features -> max over T ->
// Dense(classes) -> SoftmaxCrossEntropyLoss, backward and SGD(momentum, wd) — the frozen-backbone recipe of
// reference train.py:298-299 (gluon.Trainer 'sgd'), :324 (SoftmaxCrossEntropyLoss), :410-424 (record / backward /
// trainer.step(batch_size)) with models/vision/definitions.py:94-110 as the model.  fp32 throughout.
//
// Forward keeps what BPTT needs (r, z, n and the h2h candidate term per step); backward walks the steps in the
// reverse of each direction's own order inside one persistent workgroup per (direction, 4 batch rows), the
// weight gradients are three transposed GEMMs over all B*T rows afterwards.
#include "common.h"
#include "train.h"

namespace {

constexpr int NB = 4;   // batch rows per workgroup of the recurrent kernels

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates buffer: [dir][B*T][4H] = r | z | n | (W_hn h + b_hn)
__global__ void gru_train_fwd_kernel(const float *__restrict__ gi,    // [B*T][2*3H]  x W_ih^T + b_ih, both directions
                                     const float *__restrict__ whT,   // [2][H][3H]
                                     const float *__restrict__ bh,    // [2][3H]
                                     float *__restrict__ seq,         // [B*T][2H]
                                     float *__restrict__ gates, int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 3 * H;
  float *hs = lds;             // [NB][H]
  float *gh = hs + NB * H;     // [NB][3H]
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const float *wcol = whT + (long)dir * H * GH + j;
  const float bj = bh[dir * GH + j];
  for (int i = j; i < NB * H; i += GH) hs[i] = 0.f;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = bj;
    for (int k = 0; k < H; k += 4) {
      const float w0 = wcol[(long)(k + 0) * GH], w1 = wcol[(long)(k + 1) * GH];
      const float w2 = wcol[(long)(k + 2) * GH], w3 = wcol[(long)(k + 3) * GH];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 hv = *(const float4 *)(hs + b * H + k);
        acc[b] = fmaf(w0, hv.x, acc[b]);
        acc[b] = fmaf(w1, hv.y, acc[b]);
        acc[b] = fmaf(w2, hv.z, acc[b]);
        acc[b] = fmaf(w3, hv.w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) gh[b * GH + j] = acc[b];
    __syncthreads();
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, u = idx - b * H, bg = b0 + b;
      if (bg >= B) continue;
      const long row = (long)bg * T + t;
      const float *g = gi + row * (2 * GH) + dir * GH;
      const float *q = gh + b * GH;
      const float r = sigm(g[u] + q[u]);
      const float z = sigm(g[H + u] + q[H + u]);
      const float n = tanhf(g[2 * H + u] + r * q[2 * H + u]);
      const float hn = (1.f - z) * n + z * hs[idx];
      float *sv = gates + ((long)dir * B * T + row) * (4 * H);
      sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = q[2 * H + u];
      hs[idx] = hn;
      seq[row * (2 * H) + dir * H + u] = hn;
    }
    __syncthreads();
  }
}

__global__ void pool_max_arg_kernel(const float *__restrict__ x, int B, int T, int F, float *__restrict__ y,
                                    int32_t *__restrict__ arg) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * F) return;
  const int b = (int)(id / F), f = (int)(id % F);
  const float *p = x + (long)b * T * F + f;
  float best = p[0];
  int at = 0;
  for (int t = 1; t < T; ++t) {
    const float v = p[(long)t * F];
    if (v > best) { best = v; at = t; }      // first maximum takes the gradient
  }
  y[id] = best;
  arg[id] = at;
}

// per-sample loss -log softmax(logits)[label] (gluon SoftmaxCrossEntropyLoss, sparse labels) and d(sum loss)/dlogits
__global__ void softmax_ce_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels, int B, int C,
                                  float *__restrict__ loss, float *__restrict__ dlogits) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *p = logits + (long)b * C;
  float m = p[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(p[c] - m);
  const float lse = m + logf(sum);
  const int lab = labels[b];
  loss[b] = lse - p[lab];
  for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = expf(p[c] - lse) - (c == lab ? 1.f : 0.f);
}

// Dense backward (tiny: B x C x K = 32 x 11 x 256): one thread per weight / per pooled element
__global__ void dense_bwd_kernel(const float *__restrict__ dlogits, const float *__restrict__ pooled,
                                 const float *__restrict__ wd, int B, int C, int K, float *__restrict__ dwd,
                                 float *__restrict__ dbd, float *__restrict__ dpooled) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id < C * K) {
    const int c = id / K, k = id - c * K;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a = fmaf(dlogits[(long)b * C + c], pooled[(long)b * K + k], a);
    dwd[id] = a;
  }
  if (id < C) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlogits[(long)b * C + id];
    dbd[id] = a;
  }
  if (id < B * K) {
    const int b = id / K, k = id - b * K;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(dlogits[(long)b * C + c], wd[(long)c * K + k], a);
    dpooled[id] = a;
  }
}

__global__ void scatter_pool_grad_kernel(const float *__restrict__ dpooled, const int32_t *__restrict__ arg, int B,
                                         int T, int F, float *__restrict__ dseq) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * T * F) return;
  const int f = (int)(id % F);
  const long bt = id / F;
  const int t = (int)(bt % T), b = (int)(bt / T);
  dseq[id] = arg[(long)b * F + f] == t ? dpooled[(long)b * F + f] : 0.f;
}

// BPTT of one direction for NB batch rows: thread j = (gate block g, unit u)
__global__ void gru_train_bwd_kernel(const float *__restrict__ seq, const float *__restrict__ gates,
                                     const float *__restrict__ dseq, const float *__restrict__ wh,   // [2][3H][H]
                                     float *__restrict__ dgi, float *__restrict__ dgh,               // [B*T][2*3H]
                                     float *__restrict__ hprev,                                      // [2][B*T][H]
                                     int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 3 * H;
  float *dh = lds;                  // [NB][H]   gradient flowing into h_t from the later step
  float *dgs = dh + NB * H;         // [NB][3H]  this step's h2h pre-activation gradients
  float *part = dgs + NB * GH;      // [NB][3][H]
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const int g = j / H, u = j - g * H;
  const float *wrow = wh + (long)dir * GH * H + (long)g * H * H + u;   // W_hh[g*H + jj][u], jj = 0..H-1
  for (int i = j; i < NB * H; i += GH) dh[i] = 0.f;
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {          // reverse of the direction's own walking order
    const int t = dir ? T - 1 - s : s;
    const int tp = dir ? t + 1 : t - 1;       // where h_prev of this step was emitted
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H, bg = b0 + b;
      float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_nr = 0.f, dhp = 0.f, hp = 0.f;
      if (bg < B) {
        const long row = (long)bg * T + t;
        const float *sv = gates + ((long)dir * B * T + row) * (4 * H);
        const float r = sv[uu], z = sv[H + uu], n = sv[2 * H + uu], ghn = sv[3 * H + uu];
        hp = s > 0 ? seq[((long)bg * T + tp) * (2 * H) + dir * H + uu] : 0.f;
        const float dht = dh[idx] + dseq[row * (2 * H) + dir * H + uu];
        const float dn = dht * (1.f - z), dz = dht * (hp - n);
        dhp = dht * z;
        d_n = dn * (1.f - n * n);
        d_z = dz * z * (1.f - z);
        d_r = d_n * ghn * r * (1.f - r);
        d_nr = d_n * r;
        float *o1 = dgi + row * (2 * GH) + dir * GH, *o2 = dgh + row * (2 * GH) + dir * GH;
        o1[uu] = d_r; o1[H + uu] = d_z; o1[2 * H + uu] = d_n;
        o2[uu] = d_r; o2[H + uu] = d_z; o2[2 * H + uu] = d_nr;
        hprev[((long)dir * B * T + row) * H + uu] = hp;
      }
      dgs[b * GH + uu] = d_r; dgs[b * GH + H + uu] = d_z; dgs[b * GH + 2 * H + uu] = d_nr;
      dh[idx] = dhp;
    }
    __syncthreads();
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    for (int jj = 0; jj < H; jj += 4) {
      const float w0 = wrow[(long)(jj + 0) * H], w1 = wrow[(long)(jj + 1) * H];
      const float w2 = wrow[(long)(jj + 2) * H], w3 = wrow[(long)(jj + 3) * H];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj);
        acc[b] = fmaf(w0, dv.x, acc[b]);
        acc[b] = fmaf(w1, dv.y, acc[b]);
        acc[b] = fmaf(w2, dv.z, acc[b]);
        acc[b] = fmaf(w3, dv.w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) part[(b * 3 + g) * H + u] = acc[b];
    __syncthreads();
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H;
      dh[idx] += part[(b * 3 + 0) * H + uu] + part[(b * 3 + 1) * H + uu] + part[(b * 3 + 2) * H + uu];
    }
    __syncthreads();
  }
}

// ---- LSTM (gate order [i, f, g, o], mx.gluon.rnn.LSTM) ----------------------------------------------------------
// gates buffer: [dir][B*T][5H] = i | f | g | o | c_t
__global__ void lstm_train_fwd_kernel(const float *__restrict__ gi,    // [B*T][2*4H]  x W_ih^T + b_ih, both directions
                                      const float *__restrict__ whT,   // [2][H][4H]
                                      const float *__restrict__ bh,    // [2][4H]
                                      float *__restrict__ seq,         // [B*T][2H]
                                      float *__restrict__ gates, int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 4 * H;
  float *hs = lds;             // [NB][H]
  float *cs = hs + NB * H;     // [NB][H]
  float *gh = cs + NB * H;     // [NB][4H]
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const float *wcol = whT + (long)dir * H * GH + j;
  const float bj = bh[dir * GH + j];
  for (int i = j; i < NB * H; i += GH) { hs[i] = 0.f; cs[i] = 0.f; }
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = bj;
    for (int k = 0; k < H; k += 4) {
      const float w0 = wcol[(long)(k + 0) * GH], w1 = wcol[(long)(k + 1) * GH];
      const float w2 = wcol[(long)(k + 2) * GH], w3 = wcol[(long)(k + 3) * GH];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 hv = *(const float4 *)(hs + b * H + k);
        acc[b] = fmaf(w0, hv.x, acc[b]);
        acc[b] = fmaf(w1, hv.y, acc[b]);
        acc[b] = fmaf(w2, hv.z, acc[b]);
        acc[b] = fmaf(w3, hv.w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) gh[b * GH + j] = acc[b];
    __syncthreads();
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, u = idx - b * H, bg = b0 + b;
      if (bg >= B) continue;
      const long row = (long)bg * T + t;
      const float *g = gi + row * (2 * GH) + dir * GH;
      const float *q = gh + b * GH;
      const float ig = sigm(g[u] + q[u]);
      const float fg = sigm(g[H + u] + q[H + u]);
      const float gg = tanhf(g[2 * H + u] + q[2 * H + u]);
      const float og = sigm(g[3 * H + u] + q[3 * H + u]);
      const float c2 = fg * cs[idx] + ig * gg;
      const float hn = og * tanhf(c2);
      float *sv = gates + ((long)dir * B * T + row) * (5 * H);
      sv[u] = ig; sv[H + u] = fg; sv[2 * H + u] = gg; sv[3 * H + u] = og; sv[4 * H + u] = c2;
      cs[idx] = c2;
      hs[idx] = hn;
      seq[row * (2 * H) + dir * H + u] = hn;
    }
    __syncthreads();
  }
}

// BPTT of one LSTM direction for NB batch rows: thread j = (gate block g of 4, unit u).  The pre-activation
// gradient is the same for the i2h and the h2h branch, so only dgi is written (the caller uses it for both).
__global__ void lstm_train_bwd_kernel(const float *__restrict__ seq, const float *__restrict__ gates,
                                      const float *__restrict__ dseq, const float *__restrict__ wh,   // [2][4H][H]
                                      float *__restrict__ dgi,                                        // [B*T][2*4H]
                                      float *__restrict__ hprev,                                      // [2][B*T][H]
                                      int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 4 * H;
  float *dh = lds;                  // [NB][H]   gradient flowing into h_t from the later step
  float *dc = dh + NB * H;          // [NB][H]   ... into c_t
  float *dgs = dc + NB * H;         // [NB][4H]  this step's pre-activation gradients
  float *part = dgs + NB * GH;      // [NB][4][H]
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const int g = j / H, u = j - g * H;
  const float *wrow = wh + (long)dir * GH * H + (long)g * H * H + u;   // W_hh[g*H + jj][u], jj = 0..H-1
  for (int i = j; i < NB * H; i += GH) { dh[i] = 0.f; dc[i] = 0.f; }
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {
    const int t = dir ? T - 1 - s : s;
    const int tp = dir ? t + 1 : t - 1;
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H, bg = b0 + b;
      float d_i = 0.f, d_f = 0.f, d_g = 0.f, d_o = 0.f, dcp = 0.f;
      if (bg < B) {
        const long row = (long)bg * T + t;
        const float *sv = gates + ((long)dir * B * T + row) * (5 * H);
        const float ig = sv[uu], fg = sv[H + uu], gg = sv[2 * H + uu], og = sv[3 * H + uu], c2 = sv[4 * H + uu];
        float hp = 0.f, cp = 0.f;
        if (s > 0) {
          const long rp = (long)bg * T + tp;
          hp = seq[rp * (2 * H) + dir * H + uu];
          cp = gates[((long)dir * B * T + rp) * (5 * H) + 4 * H + uu];
        }
        const float tc = tanhf(c2);
        const float dht = dh[idx] + dseq[row * (2 * H) + dir * H + uu];
        const float dct = dc[idx] + dht * og * (1.f - tc * tc);
        d_o = dht * tc * og * (1.f - og);
        d_i = dct * gg * ig * (1.f - ig);
        d_f = dct * cp * fg * (1.f - fg);
        d_g = dct * ig * (1.f - gg * gg);
        dcp = dct * fg;
        float *o1 = dgi + row * (2 * GH) + dir * GH;
        o1[uu] = d_i; o1[H + uu] = d_f; o1[2 * H + uu] = d_g; o1[3 * H + uu] = d_o;
        hprev[((long)dir * B * T + row) * H + uu] = hp;
      }
      dgs[b * GH + uu] = d_i; dgs[b * GH + H + uu] = d_f; dgs[b * GH + 2 * H + uu] = d_g; dgs[b * GH + 3 * H + uu] = d_o;
      dc[idx] = dcp;
    }
    __syncthreads();
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    for (int jj = 0; jj < H; jj += 4) {
      const float w0 = wrow[(long)(jj + 0) * H], w1 = wrow[(long)(jj + 1) * H];
      const float w2 = wrow[(long)(jj + 2) * H], w3 = wrow[(long)(jj + 3) * H];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj);
        acc[b] = fmaf(w0, dv.x, acc[b]);
        acc[b] = fmaf(w1, dv.y, acc[b]);
        acc[b] = fmaf(w2, dv.z, acc[b]);
        acc[b] = fmaf(w3, dv.w, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) part[(b * 4 + g) * H + u] = acc[b];
    __syncthreads();
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H;
      dh[idx] = part[(b * 4 + 0) * H + uu] + part[(b * 4 + 1) * H + uu] + part[(b * 4 + 2) * H + uu] +
                part[(b * 4 + 3) * H + uu];
    }
    __syncthreads();
  }
}

// C[M][N] = A^T B, A [K][lda] (M columns), B [K][ldb] (N columns); 64x64 tile, 256 threads x 4x4 outputs
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float *__restrict__ A, int lda,
                                                          const float *__restrict__ Bm, int ldb,
                                                          float *__restrict__ Cm, int ldc, int M, int N, int K) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = t; i < 16 * 64; i += 256) {
      const int kk = i >> 6, c = i & 63;
      As[kk][c] = (k0 + kk < K && m0 + c < M) ? A[(long)(k0 + kk) * lda + m0 + c] : 0.f;
      Bs[kk][c] = (k0 + kk < K && n0 + c < N) ? Bm[(long)(k0 + kk) * ldb + n0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = fmaf(a[i], b[q], acc[i][q]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + q;
      if (m < M && n < N) Cm[(long)m * ldc + n] = acc[i][q];
    }
}

__global__ void colsum_f32_kernel(const float *__restrict__ A, int lda, int rows, int cols, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += A[(long)r * lda + c];
  out[c] = a;
}

// MXNet sgd_mom_update [EXT]: mom = momentum*mom - lr*(rescale*grad + wd*w); w += mom
__global__ void sgd_momentum_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ mom,
                                    long n, float lr, float momentum, float wd, float rescale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = momentum * mom[i] - lr * (rescale * g[i] + wd * w[i]);
  mom[i] = m;
  w[i] += m;
}

__global__ void transpose_f32_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(long)c * rows + r] = src[i];
}

}  // namespace

#define TN_LAUNCH_CHECK() do { TN_HIP_CHECK(hipGetLastError()); return TN_OK; } while (0)

int launch_gru_train_fwd(const float *gi, const float *whT, const float *bh, float *seq, float *gates, int B, int T,
                         int H, hipStream_t s) {
  TN_REQUIRE(3 * H <= 1024 && H % 4 == 0, "gru_train: 3*hidden must be <= 1024 and hidden % 4 == 0");
  hipLaunchKernelGGL(gru_train_fwd_kernel, dim3((B + NB - 1) / NB, 2), dim3(3 * H), (size_t)(NB * H + NB * 3 * H) * 4, s,
                     gi, whT, bh, seq, gates, B, T, H);
  TN_LAUNCH_CHECK();
}
int launch_pool_max_arg(const float *x, int B, int T, int F, float *y, int32_t *arg, hipStream_t s) {
  hipLaunchKernelGGL(pool_max_arg_kernel, dim3(((long)B * F + 255) / 256), dim3(256), 0, s, x, B, T, F, y, arg);
  TN_LAUNCH_CHECK();
}
int launch_softmax_ce(const float *logits, const int32_t *labels, int B, int C, float *loss, float *dlogits,
                      hipStream_t s) {
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((B + 63) / 64), dim3(64), 0, s, logits, labels, B, C, loss, dlogits);
  TN_LAUNCH_CHECK();
}
int launch_dense_bwd(const float *dlogits, const float *pooled, const float *wd, int B, int C, int K, float *dwd,
                     float *dbd, float *dpooled, hipStream_t s) {
  const int n = (C * K > B * K ? C * K : B * K);
  hipLaunchKernelGGL(dense_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dlogits, pooled, wd, B, C, K, dwd, dbd,
                     dpooled);
  TN_LAUNCH_CHECK();
}
int launch_scatter_pool_grad(const float *dpooled, const int32_t *arg, int B, int T, int F, float *dseq, hipStream_t s) {
  hipLaunchKernelGGL(scatter_pool_grad_kernel, dim3(((long)B * T * F + 255) / 256), dim3(256), 0, s, dpooled, arg, B, T,
                     F, dseq);
  TN_LAUNCH_CHECK();
}
int launch_gru_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                         float *dgh, float *hprev, int B, int T, int H, hipStream_t s) {
  hipLaunchKernelGGL(gru_train_bwd_kernel, dim3((B + NB - 1) / NB, 2), dim3(3 * H),
                     (size_t)(NB * H + NB * 3 * H + NB * 3 * H) * 4, s, seq, gates, dseq, wh, dgi, dgh, hprev, B, T, H);
  TN_LAUNCH_CHECK();
}
int launch_lstm_train_fwd(const float *gi, const float *whT, const float *bh, float *seq, float *gates, int B, int T,
                          int H, hipStream_t s) {
  TN_REQUIRE(4 * H <= 1024 && H % 4 == 0, "lstm_train: 4*hidden must be <= 1024 and hidden % 4 == 0");
  hipLaunchKernelGGL(lstm_train_fwd_kernel, dim3((B + NB - 1) / NB, 2), dim3(4 * H), (size_t)(2 * NB * H + NB * 4 * H) * 4,
                     s, gi, whT, bh, seq, gates, B, T, H);
  TN_LAUNCH_CHECK();
}
int launch_lstm_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                          float *hprev, int B, int T, int H, hipStream_t s) {
  hipLaunchKernelGGL(lstm_train_bwd_kernel, dim3((B + NB - 1) / NB, 2), dim3(4 * H),
                     (size_t)(2 * NB * H + NB * 4 * H + NB * 4 * H) * 4, s, seq, gates, dseq, wh, dgi, hprev, B, T, H);
  TN_LAUNCH_CHECK();
}
int launch_gemm_tn_f32(const float *A, int lda, const float *Bm, int ldb, float *Cm, int ldc, int M, int N, int K,
                       hipStream_t s) {
  hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, A, lda, Bm, ldb, Cm, ldc,
                     M, N, K);
  TN_LAUNCH_CHECK();
}
int launch_colsum_f32(const float *A, int lda, int rows, int cols, float *out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 127) / 128), dim3(128), 0, s, A, lda, rows, cols, out);
  TN_LAUNCH_CHECK();
}
int launch_sgd_momentum(float *w, const float *g, float *mom, long n, float lr, float momentum, float wd,
                        float rescale, hipStream_t s) {
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, g, mom, n, lr, momentum, wd, rescale);
  TN_LAUNCH_CHECK();
}
int launch_transpose_f32(const float *src, int rows, int cols, float *dst, hipStream_t s) {
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(((long)rows * cols + 255) / 256), dim3(256), 0, s, src, rows, cols, dst);
  TN_LAUNCH_CHECK();
}

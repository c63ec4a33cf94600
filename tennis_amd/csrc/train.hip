// Training step of the temporal head: bi-GRU(hidden) over (B,T,F) features -> max over T ->
// Dense(classes) -> SoftmaxCrossEntropyLoss, backward and SGD(momentum, wd) — the frozen-backbone recipe of
// reference train.py:298-299 (gluon.Trainer 'sgd'), :324 (SoftmaxCrossEntropyLoss), :410-424 (record / backward /
// trainer.step(batch_size)) with models/vision/definitions.py:94-110 as the model.  fp32 throughout.
//
// The forward recurrence is rnn.hip's persistent kernel with its `save` output (r, z, n and the h2h candidate term per
// step; LSTM: i, f, g, o, c); backward walks the steps in the
// reverse of each direction's own order inside one persistent workgroup per (direction, 4 batch rows), the
// weight gradients are three transposed GEMMs over all B*T rows afterwards.
#include "common.h"
#include "train.h"
#include "rnn_dot.h"

namespace {


__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

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

// BPTT of one direction for NB batch rows: thread j = (gate block g, unit u).  As in rnn.hip's forward kernel the
// first KR values of the thread's W_hh column (loop-invariant over the steps) stay in registers, the rest is streamed
// 16 loads at a time, and small batches run one row per workgroup.
template <int NB, int KR, int MAXT, int KL = 0>   // KL > 0 (NB = 1, H % 16 == 0): KL more weights in LDS, x through DPP (rnn_dot.h)
__global__ __launch_bounds__(MAXT) void gru_train_bwd_kernel(const float *__restrict__ seq, const float *__restrict__ gates,
                                     const float *__restrict__ dseq, const float *__restrict__ wh,   // [dirs][3H][H]
                                     float *__restrict__ dgi, float *__restrict__ dgh,               // [B*T][dirs*3H]
                                     float *__restrict__ hprev,                                      // [dirs][B*T][H]
                                     int B, int T, int H, int dirs,
                                     const int32_t *__restrict__ valid_len,    // [B] or null: steps >= valid_len never ran
                                     const float *__restrict__ dh_last) {      // [dirs][B][H] or null: d loss / d final state
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 3 * H;
  float *dh = lds;                  // [NB][H]   gradient flowing into h_t from the later step
  float *dgs = dh + NB * H;         // [NB][3H]  this step's h2h pre-activation gradients
  float *part = dgs + NB * GH;      // [NB][3][H]
  float *wl = part + NB * GH;       // [KL/4][3H][4] (KL > 0)
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const int g = j / H, u = j - g * H;
  const float *wrow = wh + (long)dir * GH * H + (long)g * H * H + u;   // W_hh[g*H + jj][u], jj = 0..H-1
  float wr[KR > 0 ? KR : 1];
#pragma unroll
  for (int k = 0; k < KR; ++k) wr[k] = wrow[(long)k * H];
  if constexpr (KL > 0) rnn_dot_fill_lds<KR, KL>(wl, GH, j, wrow, H);
  for (int i = j; i < NB * H; i += GH) dh[i] = 0.f;
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {          // reverse of the direction's own walking order
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H, bg = b0 + b;
      float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_nr = 0.f, dhp = 0.f, hp = 0.f;
      const int vlen = bg < B ? (valid_len ? valid_len[bg] : T) : 0;
      if (s < vlen) {
        const int t = dir ? vlen - 1 - s : s;       // the reverse direction starts at the row's last valid step
        const int tp = dir ? t + 1 : t - 1;         // where h_prev of this step was emitted
        const long row = (long)bg * T + t;
        const float *sv = gates + ((long)dir * B * T + row) * (4 * H);
        const float r = sv[uu], z = sv[H + uu], n = sv[2 * H + uu], ghn = sv[3 * H + uu];
        hp = s > 0 ? seq[((long)bg * T + tp) * (dirs * H) + dir * H + uu] : 0.f;
        const float carry = (s == vlen - 1 && dh_last) ? dh_last[((long)dir * B + bg) * H + uu] : dh[idx];
        const float dht = carry + dseq[row * (dirs * H) + dir * H + uu];
        const float dn = dht * (1.f - z), dz = dht * (hp - n);
        dhp = dht * z;
        d_n = dn * (1.f - n * n);
        d_z = dz * z * (1.f - z);
        d_r = d_n * ghn * r * (1.f - r);
        d_nr = d_n * r;
        float *o1 = dgi + row * (dirs * GH) + dir * GH, *o2 = dgh + row * (dirs * GH) + dir * GH;
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
    if constexpr (KL > 0) {
      static_assert(KL == 0 || NB == 1, "the LDS share is built for one row per workgroup");
      acc[0] = rnn_dot_big<KR, KL>(0.f, wr, wl, GH, j, wrow, H, dgs + g * H, H, (j & 3) * 4);
    } else {
#pragma unroll
    for (int jj = 0; jj < KR; jj += 4) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj);
        acc[b] = fmaf(wr[jj + 0], dv.x, acc[b]);
        acc[b] = fmaf(wr[jj + 1], dv.y, acc[b]);
        acc[b] = fmaf(wr[jj + 2], dv.z, acc[b]);
        acc[b] = fmaf(wr[jj + 3], dv.w, acc[b]);
      }
    }
    int jj = KR;
    for (; jj + 16 <= H; jj += 16) {
      float w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = wrow[(long)(jj + i) * H];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj + i);
          acc[b] = fmaf(w[i + 0], dv.x, acc[b]);
          acc[b] = fmaf(w[i + 1], dv.y, acc[b]);
          acc[b] = fmaf(w[i + 2], dv.z, acc[b]);
          acc[b] = fmaf(w[i + 3], dv.w, acc[b]);
        }
      }
    }
    for (; jj < H; jj += 4) {
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

// ---- LSTM (gate order [i, f, g, o], mx.gluon.rnn.LSTM); saved per step: i | f | g | o | c_t ----
// BPTT of one LSTM direction for NB batch rows: thread j = (gate block g of 4, unit u).  The pre-activation
// gradient is the same for the i2h and the h2h branch, so only dgi is written (the caller uses it for both).
template <int NB, int KR, int MAXT, int KL = 0>   // KL > 0 (NB = 1, H % 16 == 0): KL more weights in LDS, x through DPP (rnn_dot.h)
__global__ __launch_bounds__(MAXT) void lstm_train_bwd_kernel(const float *__restrict__ seq, const float *__restrict__ gates,
                                      const float *__restrict__ dseq, const float *__restrict__ wh,   // [dirs][4H][H]
                                      float *__restrict__ dgi,                                        // [B*T][dirs*4H]
                                      float *__restrict__ hprev,                                      // [dirs][B*T][H]
                                      int B, int T, int H, int dirs,
                                      const int32_t *__restrict__ valid_len,     // [B] or null: steps >= valid_len never ran
                                      const float *__restrict__ dh_last,         // [dirs][B][H] or null: d loss / d final h
                                      const float *__restrict__ dc_last) {       // ... / d final c
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = 4 * H;
  float *dh = lds;                  // [NB][H]   gradient flowing into h_t from the later step
  float *dc = dh + NB * H;          // [NB][H]   ... into c_t
  float *dgs = dc + NB * H;         // [NB][4H]  this step's pre-activation gradients
  float *part = dgs + NB * GH;      // [NB][4][H]
  float *wl = part + NB * GH;       // [KL/4][4H][4] (KL > 0)
  const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;
  const int g = j / H, u = j - g * H;
  const float *wrow = wh + (long)dir * GH * H + (long)g * H * H + u;   // W_hh[g*H + jj][u], jj = 0..H-1
  float wr[KR > 0 ? KR : 1];
#pragma unroll
  for (int k = 0; k < KR; ++k) wr[k] = wrow[(long)k * H];
  if constexpr (KL > 0) rnn_dot_fill_lds<KR, KL>(wl, GH, j, wrow, H);
  for (int i = j; i < NB * H; i += GH) { dh[i] = 0.f; dc[i] = 0.f; }
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {
    for (int idx = j; idx < NB * H; idx += GH) {
      const int b = idx / H, uu = idx - b * H, bg = b0 + b;
      float d_i = 0.f, d_f = 0.f, d_g = 0.f, d_o = 0.f, dcp = 0.f;
      const int vlen = bg < B ? (valid_len ? valid_len[bg] : T) : 0;
      if (s < vlen) {
        const int t = dir ? vlen - 1 - s : s;
        const int tp = dir ? t + 1 : t - 1;
        const long row = (long)bg * T + t;
        const float *sv = gates + ((long)dir * B * T + row) * (5 * H);
        const float ig = sv[uu], fg = sv[H + uu], gg = sv[2 * H + uu], og = sv[3 * H + uu], c2 = sv[4 * H + uu];
        float hp = 0.f, cp = 0.f;
        if (s > 0) {
          const long rp = (long)bg * T + tp;
          hp = seq[rp * (dirs * H) + dir * H + uu];
          cp = gates[((long)dir * B * T + rp) * (5 * H) + 4 * H + uu];
        }
        const bool lastp = s == vlen - 1;
        const float tc = tanhf(c2);
        const float dht = ((lastp && dh_last) ? dh_last[((long)dir * B + bg) * H + uu] : dh[idx]) + dseq[row * (dirs * H) + dir * H + uu];
        const float dct = ((lastp && dc_last) ? dc_last[((long)dir * B + bg) * H + uu] : dc[idx]) + dht * og * (1.f - tc * tc);
        d_o = dht * tc * og * (1.f - og);
        d_i = dct * gg * ig * (1.f - ig);
        d_f = dct * cp * fg * (1.f - fg);
        d_g = dct * ig * (1.f - gg * gg);
        dcp = dct * fg;
        float *o1 = dgi + row * (dirs * GH) + dir * GH;
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
    if constexpr (KL > 0) {
      static_assert(KL == 0 || NB == 1, "the LDS share is built for one row per workgroup");
      acc[0] = rnn_dot_big<KR, KL>(0.f, wr, wl, GH, j, wrow, H, dgs + g * H, H, (j & 3) * 4);
    } else {
#pragma unroll
    for (int jj = 0; jj < KR; jj += 4) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj);
        acc[b] = fmaf(wr[jj + 0], dv.x, acc[b]);
        acc[b] = fmaf(wr[jj + 1], dv.y, acc[b]);
        acc[b] = fmaf(wr[jj + 2], dv.z, acc[b]);
        acc[b] = fmaf(wr[jj + 3], dv.w, acc[b]);
      }
    }
    int jj = KR;
    for (; jj + 16 <= H; jj += 16) {
      float w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = wrow[(long)(jj + i) * H];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 dv = *(const float4 *)(dgs + b * GH + g * H + jj + i);
          acc[b] = fmaf(w[i + 0], dv.x, acc[b]);
          acc[b] = fmaf(w[i + 1], dv.y, acc[b]);
          acc[b] = fmaf(w[i + 2], dv.z, acc[b]);
          acc[b] = fmaf(w[i + 3], dv.w, acc[b]);
        }
      }
    }
    for (; jj < H; jj += 4) {
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

// C[M][N] = A^T B, A [K][lda] (M columns), B [K][ldb] (N columns): the weight gradients dW = dG^T X over all B*T rows.
// Exact-f32 MFMA 16x16x4, 64x64 tile, 4 waves each 32x32.  Both operands are k-major in memory, which is what the
// fragments want (lane = (column, k)): rows of 64 columns are staged as they lie, no transposes.  kTnPD k-tiles of 16
// stay in flight in registers (the compiler does not overlap a plain load -> LDS -> MFMA loop by itself).
constexpr int kTnPD = 4;
template <bool BNB = false>       // BNB: the B operand goes through relu(b * bsc[n] + bsh[n]) while it is staged (finetune.hip)
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float *__restrict__ A, int lda,
                                                          const float *__restrict__ Bm, int ldb,
                                                          float *__restrict__ Cm, int ldc, int M, int N, int K, int kchunk,
                                                          const float *__restrict__ bsc = nullptr, const float *__restrict__ bsh = nullptr) {
  // split-K: slice blockIdx.z covers rows [z*kchunk, (z+1)*kchunk) and writes its own (M, N) partial result
  A += (long)blockIdx.z * kchunk * lda;
  Bm += (long)blockIdx.z * kchunk * ldb;
  Cm += (long)blockIdx.z * M * ldc;
  K = min(kchunk, K - (int)blockIdx.z * kchunk);
  __shared__ float As[2][16][64 + 4], Bs[2][16][64 + 4];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int sk = t >> 4, sc = (t & 15) * 4;           // staging: k row, 4 columns
  const bool vec = ((lda | ldb) & 3) == 0 && (((uintptr_t)A | (uintptr_t)Bm) & 15) == 0 && m0 + 64 <= M && n0 + 64 <= N;
  const int nk = (K + 15) / 16;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av[kTnPD][4], bv[kTnPD][4];
  auto fetch = [&](int it, float *a4, float *b4) {
    const int k = it * 16 + sk;
#pragma unroll
    for (int q = 0; q < 4; ++q) { a4[q] = 0.f; b4[q] = 0.f; }
    if (it >= nk || k >= K) return;
    if (vec) {
      const float4 va = *(const float4 *)(A + (long)k * lda + m0 + sc), vb = *(const float4 *)(Bm + (long)k * ldb + n0 + sc);
      a4[0] = va.x; a4[1] = va.y; a4[2] = va.z; a4[3] = va.w;
      b4[0] = vb.x; b4[1] = vb.y; b4[2] = vb.z; b4[3] = vb.w;
      if constexpr (BNB) {
        const float4 s4 = *(const float4 *)(bsc + n0 + sc), h4 = *(const float4 *)(bsh + n0 + sc);
        b4[0] = fmaxf(fmaf(b4[0], s4.x, h4.x), 0.f); b4[1] = fmaxf(fmaf(b4[1], s4.y, h4.y), 0.f);
        b4[2] = fmaxf(fmaf(b4[2], s4.z, h4.z), 0.f); b4[3] = fmaxf(fmaf(b4[3], s4.w, h4.w), 0.f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (m0 + sc + q < M) a4[q] = A[(long)k * lda + m0 + sc + q];
        if (n0 + sc + q < N) {
          b4[q] = Bm[(long)k * ldb + n0 + sc + q];
          if constexpr (BNB) b4[q] = fmaxf(fmaf(b4[q], bsc[n0 + sc + q], bsh[n0 + sc + q]), 0.f);
        }
      }
    }
  };
#pragma unroll
  for (int p = 0; p < kTnPD; ++p) fetch(p, av[p], bv[p]);
  for (int it0 = 0; it0 < nk; it0 += kTnPD) {
#pragma unroll
    for (int p = 0; p < kTnPD; ++p) {
      const int it = it0 + p;
      if (it < nk) {              // block-uniform
        const int buf = it & 1;
        *(float4 *)&As[buf][sk][sc] = make_float4(av[p][0], av[p][1], av[p][2], av[p][3]);
        *(float4 *)&Bs[buf][sk][sc] = make_float4(bv[p][0], bv[p][1], bv[p][2], bv[p][3]);
        fetch(it + kTnPD, av[p], bv[p]);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kq = ks * 4 + (lane >> 4);
          float a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            a[i] = As[buf][kq][wm * 32 + i * 16 + (lane & 15)];
            b[i] = Bs[buf][kq][wn * 32 + i * 16 + (lane & 15)];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[q], acc[i][q], 0, 0, 0);
        }
      }
    }
  }
  // D[i=m][j=n]: lane: n = lane&15, m = (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + wn * 32 + q * 16 + (lane & 15);
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (m < M) Cm[(long)m * ldc + n] = acc[i][q][r];
      }
    }
}

// sum of S (M, N) partial results (row stride N) into C (row stride ldc), slices added in order
__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int S, int M, int N, float *__restrict__ Cm, int ldc) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)M * N) return;
  float a = 0.f;
  for (int z = 0; z < S; ++z) a += ws[(long)z * M * N + id];
  Cm[(id / N) * ldc + id % N] = a;
}

// column sums of A [rows][lda] (the bias gradients): 64 columns per workgroup, 16 row groups, 16 loads in flight,
// partials combined in a fixed order
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float *__restrict__ A, int lda, int rows, int cols,
                                                          float *__restrict__ out) {
  __shared__ float part[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float a = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 15 * 16 < rows; r += 16 * 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = A[(long)(r + i * 16) * lda + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) a += v[i];
    }
    for (; r < rows; r += 16) a += A[(long)r * lda + c];
  }
  part[rg][threadIdx.x & 63] = a;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][threadIdx.x];
    out[c] = s;
  }
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
// rows per workgroup / register-resident prefix: the same policy as launch_rnn_recurrent (rnn.hip)
#define TN_BPTT_DISPATCH(KERNEL, GATES, DIRS, ...)                                                                      \
  do {                                                                                                            \
    const int threads = GATES * H;                                                                                \
    TN_REQUIRE(threads <= 1024 && H % 4 == 0, "train: gates*hidden must be <= 1024 and hidden % 4 == 0");        \
    const int nb = ((B + 3) / 4) * (DIRS) >= 256 ? 4 : 1;                                                              \
    const int kr = (threads <= 512 && H >= 128) ? 128 : (threads <= 768 && H >= 96) ? 96 : H >= 64 ? 64 : 0;      \
    const dim3 grid((B + nb - 1) / nb, DIRS), block(threads);                                                     \
    const size_t lds = (size_t)(nb * H * (GATES == 4 ? 2 : 1) + 2 * nb * GATES * H) * sizeof(float);              \
    if (nb == 1 && H == 256) {   /* the column does not fit the registers: registers + LDS + stream (rnn_dot.h) */       \
      constexpr int KRb = GATES == 3 ? 112 : 64, KLb = GATES == 3 ? 48 : 32, MTb = GATES == 3 ? 768 : 1024;              \
      const size_t lds2 = lds + (size_t)KLb * GATES * H * sizeof(float);                                                 \
      TN_SET_ATTR_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)KERNEL<1, KRb, MTb, KLb>,                      \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));    \
      hipLaunchKernelGGL((KERNEL<1, KRb, MTb, KLb>), grid, block, lds2, s, __VA_ARGS__);                                 \
    } else                                                                                                              \
    if (nb == 4) hipLaunchKernelGGL((KERNEL<4, 0, 1024>), grid, block, lds, s, __VA_ARGS__);                      \
    else if (kr == 128) hipLaunchKernelGGL((KERNEL<1, 128, 512>), grid, block, lds, s, __VA_ARGS__);              \
    else if (kr == 96) hipLaunchKernelGGL((KERNEL<1, 96, 768>), grid, block, lds, s, __VA_ARGS__);                \
    else if (kr == 64) hipLaunchKernelGGL((KERNEL<1, 64, 1024>), grid, block, lds, s, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<1, 0, 1024>), grid, block, lds, s, __VA_ARGS__);                              \
  } while (0)
int launch_gru_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                         float *dgh, float *hprev, int B, int T, int H, hipStream_t s, int dirs, const int32_t *valid_len,
                         const float *dh_last) {
  TN_BPTT_DISPATCH(gru_train_bwd_kernel, 3, dirs, seq, gates, dseq, wh, dgi, dgh, hprev, B, T, H, dirs, valid_len, dh_last);
  TN_LAUNCH_CHECK();
}
int launch_lstm_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                          float *hprev, int B, int T, int H, hipStream_t s, int dirs, const int32_t *valid_len,
                          const float *dh_last, const float *dc_last) {
  TN_BPTT_DISPATCH(lstm_train_bwd_kernel, 4, dirs, seq, gates, dseq, wh, dgi, hprev, B, T, H, dirs, valid_len, dh_last, dc_last);
  TN_LAUNCH_CHECK();
}
// C (M,N) = A^T B over K rows; bsc / bsh non-null: B -> relu(B * bsc[n] + bsh[n]) on the way in (a BatchNorm + ReLU that is never stored)
static int gemm_tn_dispatch(const float *A, int lda, const float *Bm, int ldb, const float *bsc, const float *bsh, float *Cm, int ldc, int M,
                            int N, int K, hipStream_t s, float *workspace, long workspace_floats) {
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
  // few output tiles and a long reduction (weight gradients over all pixels): split K over workgroups, partial results
  // in the caller's workspace, summed in slice order (deterministic)
  int S = 1;
  if (workspace && tiles < 256 && K >= 2048) {
    S = (512 + tiles - 1) / tiles;
    if (S > K / 512) S = K / 512;
    while (S > 1 && (long)S * M * N > workspace_floats) --S;
  }
  const bool bn = bsc != nullptr;
  if (S <= 1) {
    const dim3 grid((N + 63) / 64, (M + 63) / 64, 1);
    if (bn) hipLaunchKernelGGL(gemm_tn_f32_kernel<true>, grid, dim3(256), 0, s, A, lda, Bm, ldb, Cm, ldc, M, N, K, K, bsc, bsh);
    else hipLaunchKernelGGL(gemm_tn_f32_kernel<false>, grid, dim3(256), 0, s, A, lda, Bm, ldb, Cm, ldc, M, N, K, K, bsc, bsh);
    TN_LAUNCH_CHECK();
  }
  const int kchunk = (((K + S - 1) / S) + 15) / 16 * 16;
  S = (K + kchunk - 1) / kchunk;
  const dim3 grid((N + 63) / 64, (M + 63) / 64, S);
  if (bn) hipLaunchKernelGGL(gemm_tn_f32_kernel<true>, grid, dim3(256), 0, s, A, lda, Bm, ldb, workspace, N, M, N, K, kchunk, bsc, bsh);
  else hipLaunchKernelGGL(gemm_tn_f32_kernel<false>, grid, dim3(256), 0, s, A, lda, Bm, ldb, workspace, N, M, N, K, kchunk, bsc, bsh);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(((long)M * N + 255) / 256), dim3(256), 0, s, (const float *)workspace, S, M, N, Cm, ldc);
  TN_LAUNCH_CHECK();
}
int launch_gemm_tn_f32(const float *A, int lda, const float *Bm, int ldb, float *Cm, int ldc, int M, int N, int K,
                       hipStream_t s, float *workspace, long workspace_floats) {
  return gemm_tn_dispatch(A, lda, Bm, ldb, nullptr, nullptr, Cm, ldc, M, N, K, s, workspace, workspace_floats);
}
int launch_gemm_tn_f32_bnrelu(const float *A, int lda, const float *Bm, int ldb, const float *bsc, const float *bsh, float *Cm, int ldc,
                              int M, int N, int K, hipStream_t s, float *workspace, long workspace_floats) {
  return gemm_tn_dispatch(A, lda, Bm, ldb, bsc, bsh, Cm, ldc, M, N, K, s, workspace, workspace_floats);
}
int launch_colsum_f32(const float *A, int lda, int rows, int cols, float *out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, A, lda, rows, cols, out);
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

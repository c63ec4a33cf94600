// GNMT captioner, inference, fp32 throughout:
//   encoder      GNMTEncoder.forward, reference models/captioning/gnmt.py:136-160
//                (bi-GRU with valid_length + uni-GRU, reusing the tn_birnn kernels)
//   init state   GNMTDecoder.init_state_from_encoder, gnmt.py:224-252
//   decode step  GNMTDecoder.hybrid_forward, gnmt.py:345-404, behind NMTModel.decode_step
//                [EXT]: tgt_embed -> GRUCell0([emb, att]) -> scaled-Luong attention ->
//                GRUCell1([h0, ctx]) -> tgt_proj -> log_softmax (utils/translation.py:51-53)
//   beam search  gluonnlp BeamSearchSampler/Scorer [EXT] as driven by
//                BeamSearchTranslator.translate, utils/translation.py:55-82
// The whole token loop is enqueued on the stream with no per-step host sync; the host
// looks at a device flag every 16 steps only to stop early once every beam has finished.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "linear.h"

namespace {

constexpr float kNeg = -1e18f;

__global__ void embed_concat_kernel(const float *__restrict__ emb, const int32_t *__restrict__ tok,
                                    const float *__restrict__ att, float *__restrict__ x, int R, int E, int H) {
  const int r = blockIdx.x;
  const float *e = emb + (long)tok[r] * E;
  for (int i = threadIdx.x; i < E + H; i += blockDim.x) x[(long)r * (E + H) + i] = i < E ? e[i] : att[(long)r * H + i - E];
}

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// GRU cell gates [r,z,n]: h' = (1-z)*n + z*h ; optionally also writes h' into the first H
// columns of a concat buffer xcat (row stride ldx)
__global__ void gru_gate_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                const float *__restrict__ h, float *__restrict__ hn, float *__restrict__ xcat, int ldx,
                                int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const int r = (int)(id / H), u = (int)(id % H);
  const float *a = gi + (long)r * 3 * H, *b = gh + (long)r * 3 * H;
  const float rg = sigm(a[u] + b[u]);
  const float zg = sigm(a[H + u] + b[H + u]);
  const float ng = tanhf(a[2 * H + u] + rg * b[2 * H + u]);
  const float v = (1.f - zg) * ng + zg * h[id];
  hn[id] = v;
  if (xcat) xcat[(long)r * ldx + u] = v;
}

// LSTM cell gates [i,f,g,o] (gluon rnn.LSTMCell [EXT], same order as the fused layer): c' = f*c + i*g, h' = o*tanh(c')
__global__ void lstm_gate_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                 const float *__restrict__ c, float *__restrict__ hn, float *__restrict__ cn,
                                 float *__restrict__ xcat, int ldx, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const int r = (int)(id / H), u = (int)(id % H);
  const float *a = gi + (long)r * 4 * H, *b = gh + (long)r * 4 * H;
  const float ig = sigm(a[u] + b[u]);
  const float fg = sigm(a[H + u] + b[H + u]);
  const float gg = tanhf(a[2 * H + u] + b[2 * H + u]);
  const float og = sigm(a[3 * H + u] + b[3 * H + u]);
  const float c2 = fg * c[id] + ig * gg;
  const float v = og * tanhf(c2);
  cn[id] = c2;
  hn[id] = v;
  if (xcat) xcat[(long)r * ldx + u] = v;
}

// scaled-Luong attention for one decoder row per workgroup: scores over the source steps,
// masked softmax (masked -> -1e18, weights * mask), context; writes ctx and xcat[r, H:2H].
// q[H] (already scaled by 1/sqrt(H)) sits at the start of the dynamic LDS: q[H] | w[T] | red[256]
__device__ __forceinline__ void attention_row(float *sm, int r, int b, const float *__restrict__ keyproj,
                                              const float *__restrict__ mem, const int32_t *__restrict__ valid_len,
                                              float *__restrict__ ctx, float *__restrict__ xcat, int ldx, int T, int H) {
  float *q = sm, *w = sm + H, *red = w + T;
  const int t = threadIdx.x;
  const int vl = valid_len[b];
  const float *kp = keyproj + (long)b * T * H;
  float mx = -INFINITY;
  for (int s = t; s < T; s += 256) {
    float a = 0.f;
    const float *row = kp + (long)s * H;
    for (int i = 0; i < H; ++i) a = fmaf(q[i], row[i], a);
    a = s < vl ? a : kNeg;
    w[s] = a;
    mx = fmaxf(mx, a);
  }
  red[t] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmaxf(red[t], red[t + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int s = t; s < T; s += 256) {
    const float e = expf(w[s] - mx);
    w[s] = e;
    sum += e;
  }
  red[t] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float rs = 1.0f / red[0];
  __syncthreads();
  for (int s = t; s < T; s += 256) w[s] = s < vl ? w[s] * rs : 0.f;
  __syncthreads();
  const float *mv = mem + (long)b * T * H;
  for (int i = t; i < H; i += 256) {
    float a = 0.f;
    for (int s = 0; s < T; ++s) a = fmaf(w[s], mv[(long)s * H + i], a);
    ctx[(long)r * H + i] = a;
    xcat[(long)r * ldx + H + i] = a;
  }
}

__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ hq, const float *__restrict__ keyproj,
                                                        const float *__restrict__ mem,
                                                        const int32_t *__restrict__ valid_len, float *__restrict__ ctx,
                                                        float *__restrict__ xcat, int beam, int T, int H) {
  extern __shared__ float sm[];
  const int r = blockIdx.x;
  const float inv = 1.0f / sqrtf((float)H);
  for (int i = threadIdx.x; i < H; i += 256) sm[i] = hq[(long)r * H + i] * inv;
  __syncthreads();
  attention_row(sm, r, r / beam, keyproj, mem, valid_len, ctx, xcat, 2 * H, T, H);
}

// Beam-search step, launch 2 of 4: the first decoder cell's gate arithmetic on the fused pre-activations
// g0 (R,4H) — GRU columns [r, z, n_i2h, n_h2h] (r and z already summed over both branches), LSTM [i, f, g, o] —
// then the attention of that row.  h_prev is read from the step input x0 (its last H columns), the new state
// goes to hn (R,H) (+ cn for LSTM) and to the first H columns of x1 (row stride ldx1).
__global__ __launch_bounds__(256) void dec_attention_kernel(
    const float *__restrict__ g0, const float *__restrict__ hprev, int ldh, const float *__restrict__ cprev, int lstm,
    float *__restrict__ hn, float *__restrict__ cn, float *__restrict__ x1, int ldx1,
    const float *__restrict__ keyproj, const float *__restrict__ mem, const int32_t *__restrict__ valid_len,
    float *__restrict__ ctx, int beam, int T, int H) {
  extern __shared__ float sm[];
  const int r = blockIdx.x;
  const float inv = 1.0f / sqrtf((float)H);
  const float *g = g0 + (long)r * 4 * H;
  for (int u = threadIdx.x; u < H; u += 256) {
    float v;
    if (lstm) {
      const float ig = sigm(g[u]), fg = sigm(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigm(g[3 * H + u]);
      const float c2 = fg * cprev[(long)r * H + u] + ig * gg;
      cn[(long)r * H + u] = c2;
      v = og * tanhf(c2);
    } else {
      const float rg = sigm(g[u]), zg = sigm(g[H + u]);
      const float ng = tanhf(g[2 * H + u] + rg * g[3 * H + u]);
      v = (1.f - zg) * ng + zg * hprev[(long)r * ldh + u];
    }
    hn[(long)r * H + u] = v;
    x1[(long)r * ldx1 + u] = v;
    sm[u] = v * inv;
  }
  __syncthreads();
  attention_row(sm, r, r / beam, keyproj, mem, valid_len, ctx, x1, ldx1, T, H);
}

// ---- block-wide reductions of the beam kernel (1024 threads = 16 waves; red: 16 floats, redi: 16 ints) ----
constexpr int kBeamThreads = 1024;
__device__ __forceinline__ float block_max(float v, float *red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < kBeamThreads / 64; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float *red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < kBeamThreads / 64; ++i) r += red[i];
  return r;
}
// arg-max with ties to the lowest index (the oracle's stable descending argsort)
__device__ __forceinline__ void block_argmax(float &v, int &idx, float *red, int *redi) {
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  v = red[0]; idx = redi[0];
  for (int i = 1; i < kBeamThreads / 64; ++i)
    if (red[i] > v || (red[i] == v && redi[i] < idx)) { v = red[i]; idx = redi[i]; }
}

// Beam-search step, launch 4 of 4, one workgroup per source clip:
//   second decoder cell's gates on g1 (R,4H) (column layout as in dec_attention_kernel; h_prev / c_prev are
//   the clip's rows of x1[:, 2H:3H] / c1cur) -> projection to the vocabulary (Wp^T streamed once for all
//   beams, 4 partial sums over K) -> log_softmax -> length-penalised candidates -> top-`beam` over
//   [beam*V | finished] -> bookkeeping -> the NEXT step's inputs, re-gathered by parent beam:
//   x0 = [embed(word), ctx[parent], h0[parent]], x1[:, 2H:3H] = h1[parent], c0cur / c1cur (LSTM).
template <int NBM>
__global__ __launch_bounds__(kBeamThreads) void dec_beam_kernel(
    const float *__restrict__ g1, float *__restrict__ x1, int lstm, float *__restrict__ c1cur,
    const float *__restrict__ wpT, const float *__restrict__ bp, const float *__restrict__ h0n,
    const float *__restrict__ ctx, const float *__restrict__ c0n, float *__restrict__ x0, float *__restrict__ c0cur,
    const float *__restrict__ emb, int H, int E, int V, int beam, int step, float alpha, float Kp, int eos,
    float *__restrict__ scores, int32_t *__restrict__ alive, int32_t *__restrict__ vlen,
    const int32_t *__restrict__ samples_in, int32_t *__restrict__ samples_out, int L, int32_t *__restrict__ any_alive) {
  extern __shared__ float sm[];   // h1n[NBM*H] | c1n[NBM*H] | logits[beam*V] | part[4*beam*V] (cand aliases part) | lse[16]
  const int b = blockIdx.x, t = threadIdx.x, NC = beam * V + beam, K0 = E + 2 * H, K1 = 3 * H;
  float *h1n = sm, *c1n = h1n + NBM * H, *logits = c1n + NBM * H, *part = logits + beam * V, *cand = part;
  float *lse = part + 4 * beam * V;
  __shared__ float red[16], sel_val[16];
  __shared__ int redi[16], sel_idx[16], o_alive[16], o_vlen[16];
  // ---- cell 1 ----
  for (int idx = t; idx < NBM * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    float v = 0.f, c2 = 0.f;
    if (k < beam) {
      const long r = (long)b * beam + k;
      const float *g = g1 + r * 4 * H;
      if (lstm) {
        const float ig = sigm(g[u]), fg = sigm(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigm(g[3 * H + u]);
        c2 = fg * c1cur[r * H + u] + ig * gg;
        v = og * tanhf(c2);
      } else {
        const float rg = sigm(g[u]), zg = sigm(g[H + u]);
        const float ng = tanhf(g[2 * H + u] + rg * g[3 * H + u]);
        v = (1.f - zg) * ng + zg * x1[r * K1 + 2 * H + u];
      }
    }
    h1n[idx] = v;
    c1n[idx] = c2;
  }
  __syncthreads();
  // ---- projection: thread = (quarter of K, vocabulary column) ----
  {
    const int kq = t >> 8, tv = t & 255, kn = H / 4, k0 = kq * kn;
    for (int v = tv; v < V; v += 256) {
      float acc[NBM];
#pragma unroll
      for (int q = 0; q < NBM; ++q) acc[q] = 0.f;
#pragma unroll 8
      for (int k = k0; k < k0 + kn; ++k) {
        const float w = wpT[(long)k * V + v];
#pragma unroll
        for (int q = 0; q < NBM; ++q) acc[q] = fmaf(w, h1n[q * H + k], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < NBM; ++q)
        if (q < beam) part[(kq * beam + q) * V + v] = acc[q];
    }
  }
  __syncthreads();
  for (int c = t; c < beam * V; c += kBeamThreads) {
    const int v = c % V;
    logits[c] = bp[v] + ((part[c] + part[beam * V + c]) + (part[2 * beam * V + c] + part[3 * beam * V + c]));
  }
  __syncthreads();
  // ---- log-sum-exp per beam row ----
  for (int k = 0; k < beam; ++k) {
    const float *z = logits + k * V;
    float mx = -INFINITY;
    for (int v = t; v < V; v += kBeamThreads) mx = fmaxf(mx, z[v]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int v = t; v < V; v += kBeamThreads) sum += expf(z[v] - mx);
    sum = block_sum(sum, red);
    if (t == 0) lse[k] = mx + logf(sum);
  }
  __syncthreads();
  // ---- candidates (part is dead: cand aliases it) ----
  const float lp = powf(Kp + (float)step, alpha) / powf(Kp + 1.f, alpha);
  const float prev_lp = step == 1 ? 1.f : powf(Kp + (float)(step - 1), alpha) / powf(Kp + 1.f, alpha);
  for (int c = t; c < NC; c += kBeamThreads) {
    float v;
    if (c < beam * V) {
      const int k = c / V;
      const float logp = logits[c] - lse[k];
      v = alive[b * beam + k] ? (scores[b * beam + k] * prev_lp + logp) / lp : kNeg;
    } else {
      const int k = c - beam * V;
      v = alive[b * beam + k] ? kNeg : scores[b * beam + k];
    }
    cand[c] = v;
  }
  __syncthreads();
  // ---- top-`beam`, descending, ties -> lowest index ----
  for (int k = 0; k < beam; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = t; c < NC; c += kBeamThreads) {
      const float v = cand[c];
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    block_argmax(bv, bi, red, redi);
    if (t == 0) { sel_idx[k] = bi; sel_val[k] = bv; cand[bi] = -INFINITY; }
    __syncthreads();
  }
  // ---- bookkeeping (old values are read before any thread overwrites them) ----
  if (t < beam) { o_alive[t] = alive[b * beam + t]; o_vlen[t] = vlen[b * beam + t]; }
  __syncthreads();
  if (t < beam) {
    const int idx = sel_idx[t];
    const bool use_prev = idx >= beam * V;
    const int word = use_prev ? -1 : idx % V;
    const int bid = use_prev ? idx - beam * V : idx / V;
    scores[b * beam + t] = sel_val[t];
    vlen[b * beam + t] = o_vlen[bid] + 1 - (use_prev ? 1 : 0);
    const int al = o_alive[bid] && word != eos;
    alive[b * beam + t] = al;
    sel_idx[t] = bid;      // reuse: source beam
    sel_val[t] = (float)word;
    if (al) atomicOr(any_alive, 1);
  }
  __syncthreads();
  // ---- samples: copy the chosen parent's prefix (step entries: BOS + step-1 words), append the word ----
  for (int k = 0; k < beam; ++k) {
    const int32_t *src = samples_in + ((long)b * beam + sel_idx[k]) * L;
    int32_t *dst = samples_out + ((long)b * beam + k) * L;
    for (int i = t; i < step; i += kBeamThreads) dst[i] = src[i];
    if (t == 0) dst[step] = (int32_t)sel_val[k];
  }
  // ---- next step's inputs, states re-gathered by parent beam ----
  for (int idx = t; idx < beam * K0; idx += kBeamThreads) {
    const int k = idx / K0, i = idx - k * K0;
    const long r = (long)b * beam + k, pr = (long)b * beam + sel_idx[k];
    const int word = (int)sel_val[k];
    float v;
    if (i < E) v = emb[(long)(word > 0 ? word : 0) * E + i];
    else if (i < E + H) v = ctx[pr * H + i - E];
    else v = h0n[pr * H + i - E - H];
    x0[r * K0 + i] = v;
  }
  for (int idx = t; idx < beam * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    const long r = (long)b * beam + k, pr = (long)b * beam + sel_idx[k];
    x1[r * K1 + 2 * H + u] = h1n[sel_idx[k] * H + u];
    if (lstm) {
      c1cur[r * H + u] = c1n[sel_idx[k] * H + u];
      c0cur[r * H + u] = c0n[pr * H + u];
    }
  }
}

// first step's inputs: x0 = [embed(bos), 0, h0 of the clip], x1[:, 2H:3H] = h1 of the clip, cell states (LSTM)
__global__ void dec_init_kernel(const float *__restrict__ emb, int bos, const float *__restrict__ h0c,
                                const float *__restrict__ h1c, const float *__restrict__ c0c,
                                const float *__restrict__ c1c, float *__restrict__ x0, float *__restrict__ x1,
                                float *__restrict__ c0cur, float *__restrict__ c1cur, int beam, int H, int E) {
  const int r = blockIdx.x, b = r / beam, K0 = E + 2 * H, K1 = 3 * H;
  for (int i = threadIdx.x; i < K0; i += blockDim.x)
    x0[(long)r * K0 + i] = i < E ? emb[(long)bos * E + i] : i < E + H ? 0.f : h0c[(long)b * H + i - E - H];
  for (int u = threadIdx.x; u < H; u += blockDim.x) {
    x1[(long)r * K1 + 2 * H + u] = h1c[(long)b * H + u];
    if (c0c) { c0cur[(long)r * H + u] = c0c[(long)b * H + u]; c1cur[(long)r * H + u] = c1c[(long)b * H + u]; }
  }
}

__global__ void expand_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int B, int beam, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * beam * H) return;
  const int r = (int)(id / H), u = (int)(id % H);
  dst[id] = src ? src[(long)(r / beam) * H + u] : 0.f;
}

__global__ void beam_init_kernel(float *scores, int32_t *alive, int32_t *vlen, int32_t *tok, int32_t *samples, int L,
                                 int B, int beam, int bos) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * beam) return;
  scores[id] = (id % beam) == 0 ? 0.f : kNeg;
  alive[id] = 1;
  vlen[id] = 1;
  tok[id] = bos;
  samples[(long)id * L] = bos;
}

__global__ void beam_finalize_kernel(const int32_t *alive, int32_t *vlen, int32_t *samples, int L, int last, int R,
                                     int eos) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= R) return;
  samples[(long)id * L + last] = alive[id] ? eos : -1;
  vlen[id] += alive[id] ? 1 : 0;
}

__global__ void take_column_kernel(const int32_t *__restrict__ tgt, int ld, int col, int32_t *__restrict__ tok, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { const int v = tgt[(long)b * ld + col]; tok[b] = v > 0 ? v : 0; }
}

// MaskedSoftmaxCELoss [EXT gluonnlp]: per sample, mean over the L time steps of
// -log_softmax(logits[b,t])[label[b,t]] * (t < valid_len[b]).  One workgroup per sample.
__global__ __launch_bounds__(256) void masked_ce_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels,
                                                        int ldl, const int32_t *__restrict__ valid_len, float *__restrict__ loss,
                                                        int L, int V) {
  __shared__ float red[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int vl = valid_len[b];
  float total = 0.f;
  for (int s = 0; s < L && s < vl; ++s) {
    const float *z = logits + ((long)b * L + s) * V;
    float mx = -INFINITY;
    for (int v = t; v < V; v += 256) mx = fmaxf(mx, z[v]);
    red[t] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmaxf(red[t], red[t + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int v = t; v < V; v += 256) sum += expf(z[v] - mx);
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    if (t == 0) total += -(z[labels[(long)b * ldl + s]] - mx - logf(red[0]));
    __syncthreads();
  }
  if (t == 0) loss[b] = total / (float)L;
}

struct DevBuf {
  std::vector<void *> ptrs;
  bool failed = false;
  template <typename T>
  T *alloc(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) { failed = true; return nullptr; }
    ptrs.push_back(p);
    return (T *)p;
  }
  float *upload(const float *h, size_t n) {
    float *d = alloc<float>(n);
    if (d && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) failed = true;
    return d;
  }
  void release() { for (void *p : ptrs) (void)hipFree(p); ptrs.clear(); }
};

}  // namespace

struct tn_gnmt {
  tn_ctx *ctx;
  DevBuf pool;
  tn_birnn *enc0, *enc1;   // bi layer (F -> 2H), uni layer (2H -> H)
  int F, H, E, V, maxB, maxT, beam, maxL;
  float *wi0, *wh0, *bi0, *bh0, *wi1, *wh1, *bi1, *bh1, *wk, *wp, *bp, *emb;
  // per-call workspace
  int G;                   // gates per cell: 3 GRU, 4 LSTM
  float *seq0, *mem, *keyproj, *hl0, *hl1, *cl0, *cl1;
  float *c0[2], *c1[2];    // LSTM cell states of the two decoder layers
  int32_t *vl;
  float *h0[2], *h1[2], *att[2], *x0, *x1, *gi, *gh, *logits, *scores;
  int32_t *alive, *vlen, *tok, *gather, *samples[2], *flag;
  // fused beam-search step: stacked [i2h | h2h] weights (4H rows each), transposed projection, step buffers
  float *w0c, *b0c, *w1c, *b1c, *wpT;
  float *sx0, *sx1, *g0, *g1, *h0n, *ctxn, *c0n, *c0cur, *c1cur;
  int B, T;
};

extern "C" int tn_gnmt_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, int cell_kind,
                              int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                              int max_batch, int max_src_len, int beam, int max_length, tn_gnmt **out) {
  TN_REQUIRE(ctx && params && prefix_c && out, "tn_gnmt_create: null argument");
  TN_REQUIRE(cell_kind == TN_RNN_GRU || cell_kind == TN_RNN_LSTM, "tn_gnmt_create: cell_type must be 'gru' or 'lstm'");
  TN_REQUIRE(num_layers == 2 && num_bi_layers == 1, "tn_gnmt_create: only num_layers=2, num_bi_layers=1 (reference defaults)");
  TN_REQUIRE(beam >= 1 && beam <= 16 && vocab >= beam && max_length >= 1, "tn_gnmt_create: bad beam/vocab/max_length");
  TN_REQUIRE(input_size > 0 && hidden > 0 && hidden % 4 == 0 && embed > 0 && max_batch > 0 && max_src_len > 0,
             "tn_gnmt_create: bad shape");
  TN_HIP_CHECK(hipSetDevice(ctx->device));
  const std::string pre(prefix_c);
  std::map<std::string, const tn_param *> pm;
  for (int i = 0; i < n_params; ++i) pm[params[i].name] = &params[i];
  auto get = [&](const std::string &name, int64_t numel) -> const float * {
    auto it = pm.find(name);
    if (it == pm.end()) { tn_set_error("missing parameter: " + name); return nullptr; }
    if (it->second->numel != numel) { tn_set_error("parameter " + name + " has the wrong size"); return nullptr; }
    return it->second->data_host;
  };
  tn_gnmt *g = new tn_gnmt();   // value-initialised: every pointer member starts null
  g->ctx = ctx; g->F = input_size; g->H = hidden; g->E = embed; g->V = vocab; g->maxB = max_batch; g->maxT = max_src_len;
  g->beam = beam; g->maxL = max_length + 2; g->G = cell_kind == TN_RNN_GRU ? 3 : 4;
  auto fail = [&](int code) { g->pool.release(); if (g->enc0) tn_birnn_destroy(g->enc0); if (g->enc1) tn_birnn_destroy(g->enc1); delete g; return code; };
  // encoder layers: rename "<pre>enc_rnn0_{l,r}_*" -> "{l,r}0_*" for tn_birnn
  std::vector<tn_param> p0, p1;
  std::vector<std::string> names;
  names.reserve(16);
  const char *sfx[4] = {"i2h_weight", "h2h_weight", "i2h_bias", "h2h_bias"};
  const int H = hidden, G3 = g->G * hidden;   // (G3: gates * hidden, 3H or 4H)
  for (int d = 0; d < 2; ++d)
    for (int k = 0; k < 4; ++k) {
      const std::string src = pre + "enc_rnn0_" + (d ? "r_" : "l_") + sfx[k];
      const int64_t n = k == 0 ? (int64_t)G3 * input_size : k == 1 ? (int64_t)G3 * H : G3;
      const float *v = get(src, n);
      if (!v) return fail(TN_ERR_MISSING);
      names.push_back(std::string(d ? "r0_" : "l0_") + sfx[k]);
      p0.push_back(tn_param{nullptr, v, n});
    }
  for (size_t i = 0; i < p0.size(); ++i) p0[i].name = names[i].c_str();
  std::vector<std::string> names1;
  names1.reserve(8);
  for (int k = 0; k < 4; ++k) {
    const int64_t n = k == 0 ? (int64_t)G3 * 2 * H : k == 1 ? (int64_t)G3 * H : G3;
    const float *v = get(pre + "enc_rnn1_" + sfx[k], n);
    if (!v) return fail(TN_ERR_MISSING);
    names1.push_back(std::string("l0_") + sfx[k]);
    p1.push_back(tn_param{nullptr, v, n});
  }
  for (size_t i = 0; i < p1.size(); ++i) p1[i].name = names1[i].c_str();
  int rc = tn_birnn_create(ctx, (tn_rnn_kind)cell_kind, input_size, H, p0.data(), (int)p0.size(), "", 1, max_batch * max_src_len, &g->enc0);
  if (rc) return fail(rc);
  rc = tn_birnn_create(ctx, (tn_rnn_kind)cell_kind, 2 * H, H, p1.data(), (int)p1.size(), "", 0, max_batch * max_src_len, &g->enc1);
  if (rc) return fail(rc);
  // decoder
  const float *a;
#define UP(dst, name, n) do { a = get(pre + name, (int64_t)(n)); if (!a) return fail(TN_ERR_MISSING); dst = g->pool.upload(a, (size_t)(n)); } while (0)
  UP(g->wi0, "dec_rnn0_i2h_weight", (int64_t)G3 * (embed + H)); UP(g->wh0, "dec_rnn0_h2h_weight", (int64_t)G3 * H);
  UP(g->bi0, "dec_rnn0_i2h_bias", G3); UP(g->bh0, "dec_rnn0_h2h_bias", G3);
  UP(g->wi1, "dec_rnn1_i2h_weight", (int64_t)G3 * 2 * H); UP(g->wh1, "dec_rnn1_h2h_weight", (int64_t)G3 * H);
  UP(g->bi1, "dec_rnn1_i2h_bias", G3); UP(g->bh1, "dec_rnn1_h2h_bias", G3);
  UP(g->wk, "dec_attention_key_weight", (int64_t)H * H);
  UP(g->wp, "tgt_proj_weight", (int64_t)vocab * H); UP(g->bp, "tgt_proj_bias", vocab);
  UP(g->emb, "tgt_embed_weight", (int64_t)vocab * embed);
#undef UP
  {
    // one GEMM per decoder cell: rows of the stacked matrix = 4H gate columns over [cell input | h_prev].
    // LSTM: [Wi | Wh], bias bi + bh.  GRU: r and z likewise; the candidate keeps its two branches apart
    // (n = tanh(n_i2h + r * n_h2h)): rows 2H..3H = [Wi_n | 0], rows 3H..4H = [0 | Wh_n].
    auto stack = [&](const std::string &cell, int in_dim, float **w_out, float **b_out) -> bool {
      const float *wi = get(pre + cell + "i2h_weight", (int64_t)G3 * in_dim), *wh = get(pre + cell + "h2h_weight", (int64_t)G3 * H);
      const float *bi = get(pre + cell + "i2h_bias", G3), *bh = get(pre + cell + "h2h_bias", G3);
      if (!wi || !wh || !bi || !bh) return false;
      const int Kc = in_dim + H;
      std::vector<float> w((size_t)4 * H * Kc, 0.f), bv(4 * H, 0.f);
      for (int row = 0; row < 4 * H; ++row) {
        float *d = &w[(size_t)row * Kc];
        if (g->G == 4 || row < 2 * H) {
          memcpy(d, wi + (size_t)row * in_dim, sizeof(float) * in_dim);
          memcpy(d + in_dim, wh + (size_t)row * H, sizeof(float) * H);
          bv[row] = bi[row] + bh[row];
        } else if (row < 3 * H) {
          memcpy(d, wi + (size_t)row * in_dim, sizeof(float) * in_dim);
          bv[row] = bi[row];
        } else {
          memcpy(d + in_dim, wh + (size_t)(row - H) * H, sizeof(float) * H);
          bv[row] = bh[row - H];
        }
      }
      *w_out = g->pool.upload(w.data(), w.size());
      *b_out = g->pool.upload(bv.data(), bv.size());
      return true;
    };
    if (!stack("dec_rnn0_", embed + H, &g->w0c, &g->b0c) || !stack("dec_rnn1_", 2 * H, &g->w1c, &g->b1c)) return fail(TN_ERR_MISSING);
    const float *wp = get(pre + "tgt_proj_weight", (int64_t)vocab * H);
    std::vector<float> wt((size_t)H * vocab);
    for (int v = 0; v < vocab; ++v)
      for (int k = 0; k < H; ++k) wt[(size_t)k * vocab + v] = wp[(size_t)v * H + k];
    g->wpT = g->pool.upload(wt.data(), wt.size());
  }
  const size_t BT = (size_t)max_batch * max_src_len, R = (size_t)max_batch * beam;
  g->seq0 = g->pool.alloc<float>(BT * 2 * H); g->mem = g->pool.alloc<float>(BT * H); g->keyproj = g->pool.alloc<float>(BT * H);
  g->hl0 = g->pool.alloc<float>(2 * (size_t)max_batch * H); g->hl1 = g->pool.alloc<float>((size_t)max_batch * H);
  g->cl0 = g->pool.alloc<float>(2 * (size_t)max_batch * H); g->cl1 = g->pool.alloc<float>((size_t)max_batch * H);
  g->vl = g->pool.alloc<int32_t>(max_batch);
  for (int i = 0; i < 2; ++i) {
    g->h0[i] = g->pool.alloc<float>(R * H); g->h1[i] = g->pool.alloc<float>(R * H); g->att[i] = g->pool.alloc<float>(R * H);
    g->c0[i] = g->pool.alloc<float>(R * H); g->c1[i] = g->pool.alloc<float>(R * H);
    g->samples[i] = g->pool.alloc<int32_t>(R * g->maxL);
  }
  g->x0 = g->pool.alloc<float>(R * (embed + H)); g->x1 = g->pool.alloc<float>(R * 2 * H);
  g->gi = g->pool.alloc<float>(R * G3); g->gh = g->pool.alloc<float>(R * G3); g->logits = g->pool.alloc<float>(R * vocab);
  g->scores = g->pool.alloc<float>(R); g->alive = g->pool.alloc<int32_t>(R); g->vlen = g->pool.alloc<int32_t>(R);
  g->tok = g->pool.alloc<int32_t>(R); g->gather = g->pool.alloc<int32_t>(R); g->flag = g->pool.alloc<int32_t>(1);
  g->sx0 = g->pool.alloc<float>(R * (embed + 2 * H)); g->sx1 = g->pool.alloc<float>(R * 3 * H);
  g->g0 = g->pool.alloc<float>(R * 4 * H); g->g1 = g->pool.alloc<float>(R * 4 * H);
  g->h0n = g->pool.alloc<float>(R * H); g->ctxn = g->pool.alloc<float>(R * H); g->c0n = g->pool.alloc<float>(R * H);
  g->c0cur = g->pool.alloc<float>(R * H); g->c1cur = g->pool.alloc<float>(R * H);
  if (g->pool.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  *out = g;
  return TN_OK;
}

// GNMTEncoder.forward + the attention key projection; keeps mem / states inside the handle.
// mem_out (B,T,H) may be NULL.
extern "C" int tn_gnmt_encode(tn_gnmt *g, const float *src, const int32_t *valid_len, int batch, int steps, float *mem_out) {
  TN_REQUIRE(g && src && valid_len, "tn_gnmt_encode: null argument");
  TN_REQUIRE(batch > 0 && batch <= g->maxB && steps > 0 && steps <= g->maxT, "tn_gnmt_encode: batch/steps exceed the handle");
  TN_HIP_CHECK(hipSetDevice(g->ctx->device));
  hipStream_t s = g->ctx->stream;
  const int H = g->H;
  TN_HIP_CHECK(hipMemcpyAsync(g->vl, valid_len, sizeof(int32_t) * batch, hipMemcpyDeviceToDevice, s));
  int rc = tn_birnn_forward(g->enc0, src, batch, steps, g->vl, g->seq0, g->hl0, g->cl0);   // hl0 / cl0 = [fwd, bwd] final states
  if (rc) return rc;
  rc = tn_birnn_forward(g->enc1, g->seq0, batch, steps, g->vl, g->mem, g->hl1, g->cl1);
  if (rc) return rc;
  rc = launch_linear_f32(g->mem, H, g->wk, H, nullptr, g->keyproj, H, batch * steps, H, H, 0, s);
  if (rc) return rc;
  if (mem_out) TN_HIP_CHECK(hipMemcpyAsync(mem_out, g->mem, sizeof(float) * (size_t)batch * steps * H, hipMemcpyDeviceToDevice, s));
  g->B = batch; g->T = steps;
  return TN_OK;
}

// BeamSearchTranslator.translate after tn_gnmt_encode.  samples (B,beam,max_length+2) int32 padded
// with -1, scores (B,beam), valid_length (B,beam) are DEVICE buffers; *length_host receives the
// number of valid columns of `samples` (what the reference's sampler would have returned).
extern "C" int tn_gnmt_beam_search(tn_gnmt *g, int bos, int eos, float alpha, float K, int max_length, int32_t *samples,
                                   float *scores, int32_t *valid_length, int *length_host) {
  TN_REQUIRE(g && samples && scores && valid_length && length_host, "tn_gnmt_beam_search: null argument");
  TN_REQUIRE(g->B > 0, "tn_gnmt_beam_search: call tn_gnmt_encode first");
  TN_REQUIRE(max_length >= 1 && max_length + 2 <= g->maxL, "tn_gnmt_beam_search: max_length exceeds the handle");
  TN_REQUIRE(bos >= 0 && bos < g->V && eos >= 0 && eos < g->V, "tn_gnmt_beam_search: bos/eos outside the vocabulary");
  TN_HIP_CHECK(hipSetDevice(g->ctx->device));
  hipStream_t s = g->ctx->stream;
  const int B = g->B, T = g->T, H = g->H, E = g->E, V = g->V, beam = g->beam, R = B * beam, L = g->maxL;
  const int K0 = E + 2 * H, K1 = 3 * H;
  const bool lstm = g->G == 4;
  const int nbm = beam <= 4 ? 4 : beam <= 8 ? 8 : 16;
  const size_t att_lds = (size_t)(H + T + 256) * sizeof(float);
  const size_t beam_lds = ((size_t)2 * nbm * H + (size_t)5 * beam * V + 16) * sizeof(float);
  TN_REQUIRE(beam_lds <= 64 * 1024 && att_lds <= 64 * 1024,
             "tn_gnmt_beam_search: beam * (2*hidden + 5*vocab) or hidden + source length exceeds the step kernels' 64 KiB of LDS");
  TN_HIP_CHECK(hipMemsetAsync(g->samples[0], 0xff, sizeof(int32_t) * (size_t)R * L, s));
  TN_HIP_CHECK(hipMemsetAsync(g->samples[1], 0xff, sizeof(int32_t) * (size_t)R * L, s));
  // decoder layer 0 starts from the encoder's BACKWARD layer-0 state, layer 1 from the uni layer (gnmt.py:146-150,224-252)
  hipLaunchKernelGGL(dec_init_kernel, dim3(R), dim3(256), 0, s, (const float *)g->emb, bos, (const float *)(g->hl0 + (size_t)B * H),
                     (const float *)g->hl1, lstm ? (const float *)(g->cl0 + (size_t)B * H) : (const float *)nullptr,
                     (const float *)g->cl1, g->sx0, g->sx1, g->c0cur, g->c1cur, beam, H, E);
  hipLaunchKernelGGL(beam_init_kernel, dim3((R + 255) / 256), dim3(256), 0, s, g->scores, g->alive, g->vlen, g->tok, g->samples[0], L, B, beam, bos);
  int steps_done = 0, all_dead = 0;
  // one step = 4 launches: the loop is bound by launch-to-launch dependency latency, not by arithmetic
  for (int i = 0; i < max_length; ++i) {
    const int step = i + 1;
    if ((i & 15) == 0) TN_HIP_CHECK(hipMemsetAsync(g->flag, 0, sizeof(int32_t), s));
    int rc = launch_linear_f32(g->sx0, K0, g->w0c, K0, g->b0c, g->g0, 4 * H, R, 4 * H, K0, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_attention_kernel, dim3(R), dim3(256), att_lds, s, (const float *)g->g0, (const float *)(g->sx0 + E + H), K0,
                       (const float *)g->c0cur, lstm ? 1 : 0, g->h0n, g->c0n, g->sx1, K1, (const float *)g->keyproj,
                       (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, beam, T, H);
    rc = launch_linear_f32(g->sx1, K1, g->w1c, K1, g->b1c, g->g1, 4 * H, R, 4 * H, K1, 0, s);
    if (rc) return rc;
#define TN_BEAM_LAUNCH(NBM)                                                                                              \
  hipLaunchKernelGGL(dec_beam_kernel<NBM>, dim3(B), dim3(kBeamThreads), beam_lds, s, (const float *)g->g1, g->sx1,       \
                     lstm ? 1 : 0, g->c1cur, (const float *)g->wpT, (const float *)g->bp, (const float *)g->h0n,          \
                     (const float *)g->ctxn, (const float *)g->c0n, g->sx0, g->c0cur, (const float *)g->emb, H, E, V,     \
                     beam, step, alpha, K, eos, g->scores, g->alive, g->vlen, (const int32_t *)g->samples[0],             \
                     g->samples[1], L, g->flag)
    if (nbm == 4) TN_BEAM_LAUNCH(4);
    else if (nbm == 8) TN_BEAM_LAUNCH(8);
    else TN_BEAM_LAUNCH(16);
#undef TN_BEAM_LAUNCH
    {
      int32_t *tmp = g->samples[0]; g->samples[0] = g->samples[1]; g->samples[1] = tmp;
    }
    TN_HIP_CHECK(hipGetLastError());
    steps_done = step;
    if ((i & 15) == 15 || i == max_length - 1) {   // look at the device flag every 16 steps
      int32_t f = 1;
      TN_HIP_CHECK(hipMemcpyAsync(&f, g->flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      TN_HIP_CHECK(hipStreamSynchronize(s));
      if (!f) { all_dead = 1; break; }
    }
  }
  // g->samples[0] now holds the newest samples
  if (!all_dead) {
    hipLaunchKernelGGL(beam_finalize_kernel, dim3((R + 255) / 256), dim3(256), 0, s, g->alive, g->vlen, g->samples[0], L, steps_done + 1, R, eos);
    *length_host = steps_done + 2;
  }
  TN_HIP_CHECK(hipMemcpyAsync(samples, g->samples[0], sizeof(int32_t) * (size_t)R * L, hipMemcpyDeviceToDevice, s));
  TN_HIP_CHECK(hipMemcpyAsync(scores, g->scores, sizeof(float) * R, hipMemcpyDeviceToDevice, s));
  TN_HIP_CHECK(hipMemcpyAsync(valid_length, g->vlen, sizeof(int32_t) * R, hipMemcpyDeviceToDevice, s));
  if (all_dead) {   // the sampler returns as soon as every beam has finished: width = longest sample
    std::vector<int32_t> vl(R);
    TN_HIP_CHECK(hipMemcpyAsync(vl.data(), g->vlen, sizeof(int32_t) * R, hipMemcpyDeviceToHost, s));
    TN_HIP_CHECK(hipStreamSynchronize(s));
    int mx = 0;
    for (int v : vl) mx = v > mx ? v : mx;
    *length_host = mx;
  }
  return TN_OK;
}

// Teacher-forced decoding, NMTModel.forward -> GNMTDecoder.decode_seq (reference
// models/captioning/gnmt.py:254-304) as called by evaluate() (train_gnmt.py:280): feeds
// tgt[:, 0..L-1] one step at a time from the encoder state left by tn_gnmt_encode and writes the
// projected logits (B, L, V).  tgt is a DEVICE (B, ld) int32 array.
extern "C" int tn_gnmt_decode_seq(tn_gnmt *g, const int32_t *tgt, int ld, int steps, float *logits) {
  TN_REQUIRE(g && tgt && logits, "tn_gnmt_decode_seq: null argument");
  TN_REQUIRE(g->B > 0, "tn_gnmt_decode_seq: call tn_gnmt_encode first");
  TN_REQUIRE(steps >= 1 && ld >= steps, "tn_gnmt_decode_seq: bad target length");
  TN_HIP_CHECK(hipSetDevice(g->ctx->device));
  hipStream_t s = g->ctx->stream;
  const int B = g->B, T = g->T, H = g->H, E = g->E, V = g->V, G3 = g->G * H, R = B;
  const bool lstm = g->G == 4;
  const int nb = (R * H + 255) / 256;
  hipLaunchKernelGGL(expand_rows_kernel, dim3(nb), dim3(256), 0, s, (const float *)(g->hl0 + (size_t)B * H), g->h0[0], B, 1, H);
  hipLaunchKernelGGL(expand_rows_kernel, dim3(nb), dim3(256), 0, s, (const float *)g->hl1, g->h1[0], B, 1, H);
  hipLaunchKernelGGL(expand_rows_kernel, dim3(nb), dim3(256), 0, s, (const float *)nullptr, g->att[0], B, 1, H);
  if (lstm) {
    hipLaunchKernelGGL(expand_rows_kernel, dim3(nb), dim3(256), 0, s, (const float *)(g->cl0 + (size_t)B * H), g->c0[0], B, 1, H);
    hipLaunchKernelGGL(expand_rows_kernel, dim3(nb), dim3(256), 0, s, (const float *)g->cl1, g->c1[0], B, 1, H);
  }
  const size_t att_lds = (size_t)(H + T + 256) * sizeof(float);
  int cur = 0;
  for (int i = 0; i < steps; ++i) {
    const int nxt = cur ^ 1;
    hipLaunchKernelGGL(take_column_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tgt, ld, i, g->tok, B);
    hipLaunchKernelGGL(embed_concat_kernel, dim3(R), dim3(128), 0, s, g->emb, g->tok, g->att[cur], g->x0, R, E, H);
    int rc = launch_linear_f32(g->x0, E + H, g->wi0, E + H, g->bi0, g->gi, G3, R, G3, E + H, 0, s);
    if (rc) return rc;
    rc = launch_linear_f32(g->h0[cur], H, g->wh0, H, g->bh0, g->gh, G3, R, G3, H, 0, s);
    if (rc) return rc;
    if (lstm) hipLaunchKernelGGL(lstm_gate_kernel, dim3(nb), dim3(256), 0, s, g->gi, g->gh, g->c0[cur], g->h0[nxt], g->c0[nxt], g->x1, 2 * H, R, H);
    else hipLaunchKernelGGL(gru_gate_kernel, dim3(nb), dim3(256), 0, s, g->gi, g->gh, g->h0[cur], g->h0[nxt], g->x1, 2 * H, R, H);
    hipLaunchKernelGGL(attention_kernel, dim3(R), dim3(256), att_lds, s, g->h0[nxt], g->keyproj, g->mem, g->vl, g->att[nxt], g->x1, 1, T, H);
    rc = launch_linear_f32(g->x1, 2 * H, g->wi1, 2 * H, g->bi1, g->gi, G3, R, G3, 2 * H, 0, s);
    if (rc) return rc;
    rc = launch_linear_f32(g->h1[cur], H, g->wh1, H, g->bh1, g->gh, G3, R, G3, H, 0, s);
    if (rc) return rc;
    if (lstm) hipLaunchKernelGGL(lstm_gate_kernel, dim3(nb), dim3(256), 0, s, g->gi, g->gh, g->c1[cur], g->h1[nxt], g->c1[nxt], (float *)nullptr, 0, R, H);
    else hipLaunchKernelGGL(gru_gate_kernel, dim3(nb), dim3(256), 0, s, g->gi, g->gh, g->h1[cur], g->h1[nxt], (float *)nullptr, 0, R, H);
    rc = launch_linear_f32(g->h1[nxt], H, g->wp, H, g->bp, logits + (size_t)i * V, steps * V, R, V, H, 0, s);
    if (rc) return rc;
    cur = nxt;
  }
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// MaskedSoftmaxCELoss of gluonnlp as used at reference train_gnmt.py:256,281: loss (B,) =
// mean over the L steps of the label's negative log-probability, masked by valid_len.
extern "C" int tn_masked_softmax_ce(tn_ctx *ctx, const float *logits, const int32_t *labels, int ld_labels,
                                    const int32_t *valid_len, int batch, int steps, int vocab, float *loss) {
  TN_REQUIRE(ctx && logits && labels && valid_len && loss, "tn_masked_softmax_ce: null argument");
  TN_REQUIRE(batch > 0 && steps > 0 && vocab > 0 && ld_labels >= steps, "tn_masked_softmax_ce: bad shape");
  TN_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(masked_ce_kernel, dim3(batch), dim3(256), 0, ctx->stream, logits, labels, ld_labels, valid_len, loss, steps, vocab);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

extern "C" int tn_gnmt_destroy(tn_gnmt *g) {
  if (!g) return TN_OK;
  (void)hipSetDevice(g->ctx->device);
  tn_birnn_destroy(g->enc0);
  tn_birnn_destroy(g->enc1);
  g->pool.release();
  delete g;
  return TN_OK;
}

// GNMT captioner, inference, fp32 throughout:
//   encoder      GNMTEncoder.forward, reference models/captioning/gnmt.py:136-160
//                (bi-GRU with valid_length + uni-GRU, reusing the tn_birnn kernels)
//   init state   GNMTDecoder.init_state_from_encoder, gnmt.py:224-252
//   decode step  GNMTDecoder.hybrid_forward, gnmt.py:345-404, behind NMTModel.decode_step
//                [EXT]: tgt_embed -> GRUCell0([emb, att]) -> scaled-Luong attention ->
//                GRUCell1([h0, ctx]) -> tgt_proj -> log_softmax (utils/translation.py:51-53)
//   beam search  gluonnlp BeamSearchSampler/Scorer [EXT] as driven by
//                BeamSearchTranslator.translate, utils/translation.py:55-82
// One decode step = four launches, shared by the beam search and by teacher forcing (decode_seq): a stacked-gate
// GEMM per cell (linear.hip), dec_attention_kernel (cell-0 gates + attention) and dec_beam_kernel (cell-1 gates +
// projection + beam update + the next step's gathered inputs) resp. dec_tf_cell1_kernel + the projection GEMM.
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

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- beam-search step kernels: 1024 threads = 16 waves ----
constexpr int kBeamThreads = 1024;
#ifdef TN_DEC_STAMPS   // tuning builds only (EXTRA=-DTN_DEC_STAMPS): phase boundaries of workgroup 0, 100 MHz ticks
__device__ long long g_dec_stamps[24];
#define DEC_STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_stamps[i] = wall_clock64(); } while (0)
#else
#define DEC_STAMP(i)
#endif
// Beam-search step, launch 2 of 4, one workgroup (16 waves) per `rows` consecutive decoder rows of one clip (rows = 1:
// the step is latency-bound — few workgroups, dependent L2 round trips — so it pays to spread the rows over CUs even
// though each re-reads the clip's key projection and memory from L2):
//   first decoder cell's gate arithmetic on the stacked pre-activations g0 (R,4H) — GRU columns
//   [r, z, n_i2h, n_h2h] (r and z already summed over both branches), LSTM [i, f, g, o]; h_prev = last H columns
//   of the step input x0 — the new state goes to hn (R,H) (+ cn) and to x1[:, 0:H];
//   scaled-Luong scores against the TRANSPOSED key projection kpT (B,H,T) (thread = (quarter of H, source step):
//   coalesced along T), masked softmax (one wave per beam row), context from mem (B,T,H) (thread = (quarter of
//   the valid steps, unit)) -> ctx (R,H) and x1[:, H:2H].
template <int NBM>
__global__ __launch_bounds__(kBeamThreads) void dec_attention_kernel(
    const float *__restrict__ g0, const float *__restrict__ hprev, int ldh, const float *__restrict__ cprev, int lstm,
    float *__restrict__ hn, float *__restrict__ cn, float *__restrict__ x1, int ldx1,
    const float *__restrict__ kpT, const float *__restrict__ mem, const int32_t *__restrict__ valid_len,
    float *__restrict__ ctx, int beam, int rows, int T, int H) {
  constexpr int NP = (NBM + 3) & ~3;
  extern __shared__ float sm[];   // q[H][NP] | w[T][NP] | part[4][rows][max(T,H)]
  float *q = sm, *w = q + H * NP, *part = w + T * NP;
  const int r0 = blockIdx.x * rows, b = r0 / beam, t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int vl = min(max(valid_len[b], 0), T);
  const float inv = 1.0f / sqrtf((float)H);
  DEC_STAMP(0);
  // ---- cell 0 ----
  for (int idx = t; idx < NBM * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    float v = 0.f;
    if (k < rows) {
      const long r = (long)r0 + k;
      const float *g = g0 + r * 4 * H;
      if (lstm) {
        const float ig = sigm(g[u]), fg = sigm(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigm(g[3 * H + u]);
        const float c2 = fg * cprev[r * H + u] + ig * gg;
        cn[r * H + u] = c2;
        v = og * tanhf(c2);
      } else {
        const float rg = sigm(g[u]), zg = sigm(g[H + u]);
        const float ng = tanhf(g[2 * H + u] + rg * g[3 * H + u]);
        v = (1.f - zg) * ng + zg * hprev[r * ldh + u];
      }
      hn[r * H + u] = v;
      x1[r * ldx1 + u] = v;
    }
    q[u * NP + k] = v * inv;
  }
  __syncthreads();
  DEC_STAMP(1);
  // ---- scores: 4 partial sums over H per (row, source step) ----
  {
    const int hq = t >> 8, hn4 = H / 4, h0 = hq * hn4;
    const float *kp = kpT + (long)b * H * T;
    for (int s = t & 255; s < vl; s += 256) {
      float acc[NBM];
#pragma unroll
      for (int j = 0; j < NBM; ++j) acc[j] = 0.f;
      // loads are batched by hand (16 in flight): left to itself the compiler waits for each load before its fma
      int h = h0;
      for (; h + 16 <= h0 + hn4; h += 16) {
        float kv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) kv[i] = kp[(long)(h + i) * T + s];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int j = 0; j < NBM; ++j) acc[j] = fmaf(kv[i], q[(h + i) * NP + j], acc[j]);
      }
      for (; h < h0 + hn4; ++h) {
        const float kv = kp[(long)h * T + s];
#pragma unroll
        for (int j = 0; j < NBM; ++j) acc[j] = fmaf(kv, q[h * NP + j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < NBM; ++j)
        if (j < rows) part[(hq * rows + j) * T + s] = acc[j];
    }
  }
  __syncthreads();
  DEC_STAMP(2);
  // ---- masked softmax: wave k owns beam row k (masked -> -1e18, weights * mask) ----
  if (wid < rows) {
    const float *p0 = part + wid * T, *p1 = p0 + rows * T, *p2 = p1 + rows * T, *p3 = p2 + rows * T;
    float mx = -INFINITY;
    for (int s = lane; s < T; s += 64) {
      const float a = s < vl ? (p0[s] + p1[s]) + (p2[s] + p3[s]) : kNeg;
      w[s * NP + wid] = a;
      mx = fmaxf(mx, a);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int s = lane; s < T; s += 64) {
      const float e = expf(w[s * NP + wid] - mx);
      w[s * NP + wid] = e;
      sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float rs = 1.0f / sum;
    for (int s = lane; s < T; s += 64) w[s * NP + wid] = s < vl ? w[s * NP + wid] * rs : 0.f;
  } else if (wid < NP) {
    for (int s = lane; s < T; s += 64) w[s * NP + wid] = 0.f;      // padded rows feed unused accumulators
  }
  __syncthreads();
  DEC_STAMP(3);
  // ---- context: 4 partial sums over the valid steps per (row, unit); weights beyond valid_len are exactly 0 ----
  {
    const int sq = t >> 8, sn = (vl + 3) / 4, s0 = sq * sn, s1 = min(vl, s0 + sn);
    const float *mv = mem + (long)b * T * H;
    for (int u = t & 255; u < H; u += 256) {
      float acc[NBM];
#pragma unroll
      for (int j = 0; j < NBM; ++j) acc[j] = 0.f;
      int s = s0;
      for (; s + 16 <= s1; s += 16) {
        float m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = mv[(long)(s + i) * H + u];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int j = 0; j < NBM; ++j) acc[j] = fmaf(w[(s + i) * NP + j], m[i], acc[j]);
      }
      for (; s < s1; ++s) {
        const float m = mv[(long)s * H + u];
#pragma unroll
        for (int j = 0; j < NBM; ++j) acc[j] = fmaf(w[s * NP + j], m, acc[j]);
      }
#pragma unroll
      for (int j = 0; j < NBM; ++j)
        if (j < rows) part[(sq * rows + j) * H + u] = acc[j];
    }
  }
  __syncthreads();
  for (int idx = t; idx < rows * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    const long r = (long)r0 + k;
    const float a = (part[idx] + part[rows * H + idx]) + (part[2 * rows * H + idx] + part[3 * rows * H + idx]);
    ctx[r * H + u] = a;
    x1[r * ldx1 + H + u] = a;
  }
  DEC_STAMP(4);
}

// (B,T,H) -> (B,H,T): the key projection as the per-clip attention kernel reads it
__global__ void transpose_bth_kernel(const float *__restrict__ src, float *__restrict__ dst, int T, int H) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.y * 32, h0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (t0 + i < T && h0 + tx < H) tile[i][tx] = src[((long)b * T + t0 + i) * H + h0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (h0 + i < H && t0 + tx < T) dst[((long)b * H + h0 + i) * T + t0 + tx] = tile[tx][i];
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
  extern __shared__ float sm[];   // h1n[H][NP] | c1n[H][NP] | logits[beam*V] | part[4*beam*V] (cand aliases part) | lse[16]
  const int b = blockIdx.x, t = threadIdx.x, NC = beam * V + beam, K0 = E + 2 * H, K1 = 3 * H;
  constexpr int NP = (NBM + 3) & ~3;   // LDS pitch of one k: NBM beam rows padded to 16-byte multiples
  float *h1n = sm, *c1n = h1n + NP * H, *logits = c1n + NP * H, *part = logits + beam * V, *cand = part;
  float *lse = part + 4 * beam * V;
  __shared__ float sel_val[16], o_score[16], lps[2];
  __shared__ int sel_idx[16], sel_par[16], sel_word[16], o_alive[16], o_vlen[16];
  const int lane = t & 63, wid = t >> 6;
  DEC_STAMP(8);
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
    h1n[u * NP + k] = v;       // k-major: a projection thread reads its NBM rows with one or two wide LDS loads
    c1n[u * NP + k] = c2;
  }
  __syncthreads();
  DEC_STAMP(9);
  // ---- projection: thread = (quarter of K, vocabulary column) ----
  {
    const int kq = t >> 8, tv = t & 255, kn = H / 4, k0 = kq * kn;
    for (int v = tv; v < V; v += 256) {
      float acc[NBM];
#pragma unroll
      for (int q = 0; q < NBM; ++q) acc[q] = 0.f;
      int k = k0;
      for (; k + 16 <= k0 + kn; k += 16) {      // 16 loads in flight (batched by hand, see dec_attention_kernel)
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = wpT[(long)(k + i) * V + v];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int q = 0; q < NBM; ++q) acc[q] = fmaf(w[i], h1n[(k + i) * NP + q], acc[q]);
      }
      for (; k < k0 + kn; ++k) {
        const float w = wpT[(long)k * V + v];
#pragma unroll
        for (int q = 0; q < NBM; ++q) acc[q] = fmaf(w, h1n[k * NP + q], acc[q]);
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
  DEC_STAMP(10);
  // ---- log-sum-exp: wave k owns beam row k; the old beam state moves to LDS meanwhile ----
  if (wid < beam) {
    const float *z = logits + wid * V;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, z[v]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += expf(z[v] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) lse[wid] = mx + logf(sum);
  } else if (wid == 15) {
    if (lane < beam) { o_alive[lane] = alive[b * beam + lane]; o_vlen[lane] = vlen[b * beam + lane]; o_score[lane] = scores[b * beam + lane]; }
    if (lane == 0) {
      lps[0] = powf(Kp + (float)step, alpha) / powf(Kp + 1.f, alpha);
      lps[1] = step == 1 ? 1.f : powf(Kp + (float)(step - 1), alpha) / powf(Kp + 1.f, alpha);
    }
  }
  __syncthreads();
  DEC_STAMP(11);
  // ---- candidates (part is dead: cand aliases it) ----
  const float lp = lps[0], prev_lp = lps[1];
  for (int c = t; c < NC; c += kBeamThreads) {
    float v;
    if (c < beam * V) {
      const int k = c / V;
      const float logp = logits[c] - lse[k];
      v = o_alive[k] ? (o_score[k] * prev_lp + logp) / lp : kNeg;
    } else {
      const int k = c - beam * V;
      v = o_alive[k] ? kNeg : o_score[k];
    }
    cand[c] = v;
  }
  __syncthreads();
  DEC_STAMP(12);
  // ---- top-`beam`, descending, ties -> lowest index: one wave, no block barriers.  A lane keeps the best of its
  // strided candidates as a 64-bit key (order-preserving float bits | inverted index) and rescans only when it
  // loses that best; the wave maximum is a DPP reduction (row shifts + row broadcasts), result in lane 63 ----
  if (wid == 0) {
    auto lane_best = [&](unsigned &hi, unsigned &lo) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      int c = lane;
      for (; c + 7 * 64 < NC; c += 8 * 64) {          // LDS reads batched by hand, as the global ones above
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = cand[c + i * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (x[i] > bv) { bv = x[i]; bi = c + i * 64; }   // ascending index: ties keep the lowest
      }
      for (; c < NC; c += 64) {
        const float x = cand[c];
        if (x > bv) { bv = x; bi = c; }
      }
      const unsigned u = __float_as_uint(bv);
      hi = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      lo = 0xffffffffu - (unsigned)bi;
      if (bi == 0x7fffffff) { hi = 0u; lo = 0u; }          // no candidate left in this lane
    };
    unsigned hi, lo;
    lane_best(hi, lo);
    for (int k = 0; k < beam; ++k) {
      unsigned h = hi, l = lo;
#define TN_DPP_MAX(ctrl, rmask)                                                                       \
  {                                                                                                   \
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp((int)l, (int)l, ctrl, rmask, 0xf, false); \
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp((int)h, (int)h, ctrl, rmask, 0xf, false); \
    const bool gt = h2 > h || (h2 == h && l2 > l);                                                     \
    l = gt ? l2 : l;                                                                                  \
    h = gt ? h2 : h;                                                                                  \
  }
      TN_DPP_MAX(0x111, 0xf)   // row_shr:1
      TN_DPP_MAX(0x112, 0xf)   // row_shr:2
      TN_DPP_MAX(0x114, 0xf)   // row_shr:4
      TN_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row maximum
      TN_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1, 3
      TN_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave maximum
#undef TN_DPP_MAX
      const unsigned wl = (unsigned)__builtin_amdgcn_readlane((int)l, 63);
      const int i = (int)(0xffffffffu - wl);
      if (lo == wl && (hi | lo) != 0u) {            // this lane owned the winner: record, drop it, rescan its stride
        sel_idx[k] = i;
        sel_val[k] = cand[i];
        cand[i] = -INFINITY;
        lane_best(hi, lo);
      }
    }
  }
  __syncthreads();
  DEC_STAMP(13);
  // ---- bookkeeping ----
  if (t < beam) {
    const int idx = sel_idx[t];
    const bool use_prev = idx >= beam * V;
    const int word = use_prev ? -1 : idx % V;
    const int bid = use_prev ? idx - beam * V : idx / V;
    scores[b * beam + t] = sel_val[t];
    vlen[b * beam + t] = o_vlen[bid] + 1 - (use_prev ? 1 : 0);
    const int al = o_alive[bid] && word != eos;
    alive[b * beam + t] = al;
    sel_par[t] = bid;
    sel_word[t] = word;
    if (al) atomicOr(any_alive, 1);
  }
  __syncthreads();
  DEC_STAMP(14);
  // ---- samples: copy the chosen parent's prefix (step entries: BOS + step-1 words), append the word ----
  for (int k = 0; k < beam; ++k) {
    const int32_t *src = samples_in + ((long)b * beam + sel_par[k]) * L;
    int32_t *dst = samples_out + ((long)b * beam + k) * L;
    for (int i = t; i < step; i += kBeamThreads) dst[i] = src[i];
    if (t == 0) dst[step] = sel_word[k];
  }
  DEC_STAMP(15);
  // ---- next step's inputs, states re-gathered by parent beam ----
  for (int idx = t; idx < beam * K0; idx += kBeamThreads) {
    const int k = idx / K0, i = idx - k * K0;
    const long r = (long)b * beam + k, pr = (long)b * beam + sel_par[k];
    const int word = sel_word[k];
    float v;
    if (i < E) v = emb[(long)(word > 0 ? word : 0) * E + i];
    else if (i < E + H) v = ctx[pr * H + i - E];
    else v = h0n[pr * H + i - E - H];
    x0[r * K0 + i] = v;
  }
  for (int idx = t; idx < beam * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    const long r = (long)b * beam + k, pr = (long)b * beam + sel_par[k];
    x1[r * K1 + 2 * H + u] = h1n[u * NP + sel_par[k]];
    if (lstm) {
      c1cur[r * H + u] = c1n[u * NP + sel_par[k]];
      c0cur[r * H + u] = c0n[pr * H + u];
    }
  }
  DEC_STAMP(16);
}

// Teacher forcing (decode_seq): the same four-launch step as the beam search with beam = 1.  Before a step its inputs
// are assembled from the given token and the previous step's outputs (the encoder states / zero attention at step 0);
// after it the second cell's gates give h1 (the row the projection GEMM then maps to logits).
__global__ void dec_tf_prep_kernel(const float *__restrict__ emb, const int32_t *__restrict__ tgt, int ld, int col,
                                   const float *__restrict__ ctx_prev, const float *__restrict__ h0_prev,
                                   const float *__restrict__ h1_prev, const float *__restrict__ c0_prev,
                                   const float *__restrict__ c1_prev, float *__restrict__ x0, float *__restrict__ x1,
                                   float *__restrict__ c0cur, float *__restrict__ c1cur, int H, int E) {
  const int r = blockIdx.x, K0 = E + 2 * H, K1 = 3 * H;
  const int tok = tgt[(long)r * ld + col];
  const float *e = emb + (long)(tok > 0 ? tok : 0) * E;
  for (int i = threadIdx.x; i < K0; i += blockDim.x)
    x0[(long)r * K0 + i] = i < E ? e[i] : i < E + H ? (ctx_prev ? ctx_prev[(long)r * H + i - E] : 0.f) : h0_prev[(long)r * H + i - E - H];
  for (int u = threadIdx.x; u < H; u += blockDim.x) {
    x1[(long)r * K1 + 2 * H + u] = h1_prev[(long)r * H + u];
    if (c0_prev) { c0cur[(long)r * H + u] = c0_prev[(long)r * H + u]; c1cur[(long)r * H + u] = c1_prev[(long)r * H + u]; }
  }
}

__global__ void dec_tf_cell1_kernel(const float *__restrict__ g1, const float *__restrict__ x1, int lstm,
                                    const float *__restrict__ c1cur, float *__restrict__ h1n, float *__restrict__ c1n,
                                    int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const float *g = g1 + r * 4 * H;
  if (lstm) {
    const float ig = sigm(g[u]), fg = sigm(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigm(g[3 * H + u]);
    const float c2 = fg * c1cur[id] + ig * gg;
    c1n[id] = c2;
    h1n[id] = og * tanhf(c2);
  } else {
    const float rg = sigm(g[u]), zg = sigm(g[H + u]);
    const float ng = tanhf(g[2 * H + u] + rg * g[3 * H + u]);
    h1n[id] = (1.f - zg) * ng + zg * x1[r * 3 * H + 2 * H + u];
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
  float *wk, *wp, *bp, *emb;
  // per-call workspace
  int G;                   // gates per cell: 3 GRU, 4 LSTM
  float *seq0, *mem, *keyproj, *hl0, *hl1, *cl0, *cl1;
  int32_t *vl;
  float *scores;
  int32_t *alive, *vlen, *tok, *samples[2], *flag;
  // fused beam-search step: stacked [i2h | h2h] weights (4H rows each), transposed projection, step buffers
  float *w0c, *b0c, *w1c, *b1c, *wpT;
  float *sx0, *sx1, *g0, *g1, *h0n, *ctxn, *c0n, *c0cur, *c1cur, *keyprojT, *h1n, *c1n;
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
  for (int i = 0; i < 2; ++i) g->samples[i] = g->pool.alloc<int32_t>(R * g->maxL);
  g->scores = g->pool.alloc<float>(R); g->alive = g->pool.alloc<int32_t>(R); g->vlen = g->pool.alloc<int32_t>(R);
  g->tok = g->pool.alloc<int32_t>(R); g->flag = g->pool.alloc<int32_t>(1);
  g->h1n = g->pool.alloc<float>(R * H); g->c1n = g->pool.alloc<float>(R * H);
  g->sx0 = g->pool.alloc<float>(R * (embed + 2 * H)); g->sx1 = g->pool.alloc<float>(R * 3 * H);
  g->g0 = g->pool.alloc<float>(R * 4 * H); g->g1 = g->pool.alloc<float>(R * 4 * H);
  g->h0n = g->pool.alloc<float>(R * H); g->ctxn = g->pool.alloc<float>(R * H); g->c0n = g->pool.alloc<float>(R * H);
  g->c0cur = g->pool.alloc<float>(R * H); g->c1cur = g->pool.alloc<float>(R * H); g->keyprojT = g->pool.alloc<float>(BT * H);
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
  hipLaunchKernelGGL(transpose_bth_kernel, dim3((H + 31) / 32, (steps + 31) / 32, batch), dim3(256), 0, s, (const float *)g->keyproj,
                     g->keyprojT, steps, H);
  TN_HIP_CHECK(hipGetLastError());
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
  const int nbm = beam <= 4 ? 4 : beam == 5 ? 5 : beam <= 8 ? 8 : 16;   // beam 5: the reference's flag default
  const size_t att_lds = ((size_t)(H + T) * 4 + (size_t)4 * (T > H ? T : H)) * sizeof(float);   // one row per workgroup
  const size_t beam_lds = ((size_t)2 * ((nbm + 3) & ~3) * H + (size_t)5 * beam * V + 16) * sizeof(float);
  TN_REQUIRE(beam_lds <= 64 * 1024 && att_lds <= 64 * 1024,
             "tn_gnmt_beam_search: beam * (2*hidden + 5*vocab) or 8 * max(hidden, source length) exceeds the step kernels' 64 KiB of LDS");
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
    hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(R), dim3(kBeamThreads), att_lds, s, (const float *)g->g0,
                       (const float *)(g->sx0 + E + H), K0, (const float *)g->c0cur, lstm ? 1 : 0, g->h0n, g->c0n, g->sx1, K1,
                       (const float *)g->keyprojT, (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, beam, 1, T, H);
    rc = launch_linear_f32(g->sx1, K1, g->w1c, K1, g->b1c, g->g1, 4 * H, R, 4 * H, K1, 0, s);
    if (rc) return rc;
#define TN_BEAM_LAUNCH(NBM)                                                                                              \
  hipLaunchKernelGGL(dec_beam_kernel<NBM>, dim3(B), dim3(kBeamThreads), beam_lds, s, (const float *)g->g1, g->sx1,       \
                     lstm ? 1 : 0, g->c1cur, (const float *)g->wpT, (const float *)g->bp, (const float *)g->h0n,          \
                     (const float *)g->ctxn, (const float *)g->c0n, g->sx0, g->c0cur, (const float *)g->emb, H, E, V,     \
                     beam, step, alpha, K, eos, g->scores, g->alive, g->vlen, (const int32_t *)g->samples[0],             \
                     g->samples[1], L, g->flag)
    if (nbm == 4) TN_BEAM_LAUNCH(4);
    else if (nbm == 5) TN_BEAM_LAUNCH(5);
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
  const int B = g->B, T = g->T, H = g->H, E = g->E, V = g->V, R = B, K0 = E + 2 * H, K1 = 3 * H;
  const bool lstm = g->G == 4;
  const int nb = (R * H + 255) / 256;
  const size_t att_lds = ((size_t)(H + T) * 4 + (size_t)4 * (T > H ? T : H)) * sizeof(float);
  TN_REQUIRE(att_lds <= 64 * 1024, "tn_gnmt_decode_seq: 8 * max(hidden, source length) exceeds the step kernel's 64 KiB of LDS");
  for (int i = 0; i < steps; ++i) {
    // decoder layer 0 starts from the encoder's BACKWARD layer-0 state, layer 1 from the uni layer (gnmt.py:146-150,224-252)
    const float *h0p = i ? g->h0n : g->hl0 + (size_t)B * H, *h1p = i ? g->h1n : g->hl1;
    const float *c0p = !lstm ? nullptr : i ? g->c0n : g->cl0 + (size_t)B * H, *c1p = !lstm ? nullptr : i ? g->c1n : g->cl1;
    hipLaunchKernelGGL(dec_tf_prep_kernel, dim3(R), dim3(256), 0, s, (const float *)g->emb, tgt, ld, i,
                       i ? (const float *)g->ctxn : (const float *)nullptr, h0p, h1p, c0p, c1p, g->sx0, g->sx1, g->c0cur, g->c1cur, H, E);
    int rc = launch_linear_f32(g->sx0, K0, g->w0c, K0, g->b0c, g->g0, 4 * H, R, 4 * H, K0, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(R), dim3(kBeamThreads), att_lds, s, (const float *)g->g0,
                       (const float *)(g->sx0 + E + H), K0, (const float *)g->c0cur, lstm ? 1 : 0, g->h0n, g->c0n, g->sx1, K1,
                       (const float *)g->keyprojT, (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, 1, 1, T, H);
    rc = launch_linear_f32(g->sx1, K1, g->w1c, K1, g->b1c, g->g1, 4 * H, R, 4 * H, K1, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_tf_cell1_kernel, dim3(nb), dim3(256), 0, s, (const float *)g->g1, (const float *)g->sx1, lstm ? 1 : 0,
                       (const float *)g->c1cur, g->h1n, g->c1n, R, H);
    rc = launch_linear_f32(g->h1n, H, g->wp, H, g->bp, logits + (size_t)i * V, steps * V, R, V, H, 0, s);
    if (rc) return rc;
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

#ifdef TN_DEC_STAMPS
extern "C" int tn_dbg_dec_stamps(long long *out) {
  TN_HIP_CHECK(hipDeviceSynchronize());
  TN_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_stamps), sizeof(long long) * 24));
  return TN_OK;
}
#endif

extern "C" int tn_gnmt_destroy(tn_gnmt *g) {
  if (!g) return TN_OK;
  (void)hipSetDevice(g->ctx->device);
  tn_birnn_destroy(g->enc0);
  tn_birnn_destroy(g->enc1);
  g->pool.release();
  delete g;
  return TN_OK;
}

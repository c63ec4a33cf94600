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
#include "rnn.h"
#include "train.h"

namespace {

constexpr float kNeg = -1e18f;
// element (row r, column k) of a decoder cell's step-input matrix: row-major with pitch ld > 0, or k-group-major
// ([k/4][rows][4], the operand form of lat_tile_f32<.., true>) with -ld rows
__device__ __forceinline__ long xidx(long r, int k, int ld) { return ld > 0 ? r * ld + k : ((long)(k >> 2) * (-ld) + r) * 4 + (k & 3); }
constexpr int kGemmPitchPad = 16;   // floats; measured at config C5: 0 / 32 / 64 / 96: 44.2 us per step, 16 / 48 / 80: 42.2

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// Wave-wide maximum / sum in six DPP stages (row shifts + row broadcasts, result in lane 63, handed to every lane through an SGPR)
__device__ __forceinline__ float wave_max(float v) {
  float m = v;
#define TN_DPP_F(ctrl, rmask) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), ctrl, rmask, 0xf, false)));
  TN_DPP_F(0x111, 0xf) TN_DPP_F(0x112, 0xf) TN_DPP_F(0x114, 0xf) TN_DPP_F(0x118, 0xf) TN_DPP_F(0x142, 0xa) TN_DPP_F(0x143, 0xc)
#undef TN_DPP_F
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
  float m = v;
#define TN_DPP_F(ctrl, rmask) m += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), ctrl, rmask, 0xf, false));
  TN_DPP_F(0x111, 0xf) TN_DPP_F(0x112, 0xf) TN_DPP_F(0x114, 0xf) TN_DPP_F(0x118, 0xf) TN_DPP_F(0x142, 0xa) TN_DPP_F(0x143, 0xc)
#undef TN_DPP_F
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
}


// ---- beam-search step kernels: 1024 threads = 16 waves ----
constexpr int kBeamThreads = 1024;
// dynamic LDS the step kernels may ask for: a launch above 64 KiB needs the kernel's limit raised first (160 KiB per CU on
// gfx950; 8 KiB are left for the kernels' static arrays)
constexpr size_t kStepLdsMax = 152 * 1024;
template <typename KernT>
static int allow_lds(KernT kern, size_t bytes) {
  if (bytes > 64 * 1024)
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStepLdsMax));
  return TN_OK;
}
#ifdef TN_DEC_STAMPS   // tuning builds only (EXTRA=-DTN_DEC_STAMPS): phase boundaries of workgroup 0, 100 MHz ticks
__device__ long long g_dec_stamps[24];
#define DEC_STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_stamps[i] = wall_clock64(); } while (0)
#else
#define DEC_STAMP(i)
#endif
// Work split of the step kernels' three streaming loops (attention scores, attention context, vocabulary projection): a
// thread owns FOUR consecutive outputs (one 16-byte load per operand row) and a slice of the contraction; n4 groups of four
// outputs are dealt over 64 / 128 / 256 threads and the remaining factor of the 1024 threads (16 / 8 / 4) splits the
// contraction, the partial sums meet in LDS.  Each of these loops used to be one scalar load + one LDS read per FMA row:
// the LDS pipe (every lane reading the same broadcast value) and the number of loads in flight bound them, not the bytes.
// `split` (16, 8, 4, 2 or 1) caps the number of partial sums: the host lowers it when 16 partial rows do not fit the LDS
// (a wide beam with a large vocabulary, a very long source).
__device__ __host__ __forceinline__ int step_groups(int n4, int split) {
  const int g = n4 <= 64 ? 64 : n4 <= 128 ? 128 : 256, floor_g = 1024 / split;
  return g > floor_g ? g : floor_g;
}
static size_t att_lds_bytes(int T, int H, int split) {      // q[H] | w[Tp] | part[partials * max(Tp, H)]
  const int Tp = (T + 3) & ~3;
  const int hg = 1024 / step_groups(Tp / 4, split), sg = 1024 / step_groups(H / 4, split);
  const size_t a = (size_t)hg * Tp, b = (size_t)sg * H;
  return ((size_t)H + Tp + (a > b ? a : b)) * sizeof(float);
}
// the largest split whose attention kernel fits the step kernels' LDS (0: not even one partial row does)
static int att_split(int T, int H, size_t limit) {
  for (int sp = 16; sp >= 1; sp >>= 1)
    if (att_lds_bytes(T, H, sp) <= limit) return sp;
  return 0;
}
// Beam-search step, first launch, one workgroup (16 waves) per decoder row (the step is latency-bound - few workgroups,
// dependent L2 round trips - so it pays to spread the rows over CUs even though the beams of a clip each re-read the clip's key
// projection and memory from L2):
//   first decoder cell's gate arithmetic on the stacked pre-activations g0 (R,4H) — GRU columns
//   [r, z, n_i2h, n_h2h] (r and z already summed over both branches), LSTM [i, f, g, o]; h_prev = last H columns
//   of the step input x0 — the new state goes to hn (R,H) (+ cn) and to x1[:, 0:H];
//   scaled-Luong scores against the TRANSPOSED key projection kpT (B,H,Tp) (Tp = T rounded up to 4), masked softmax (one
//   wave), context from mem (B,T,H) -> ctx (R,H) and x1[:, H:2H].
template <int NBM>
__global__ __launch_bounds__(kBeamThreads) void dec_attention_kernel(
    const float *__restrict__ g0, const float *__restrict__ hprev, int ldh, const float *__restrict__ cprev, int lstm,
    float *__restrict__ hn, float *__restrict__ cn, float *__restrict__ x1, int ldx1,
    const float *__restrict__ kpT, const float *__restrict__ mem, const int32_t *__restrict__ valid_len,
    float *__restrict__ ctx, int beam, int rows, int T, int H, float *__restrict__ wsave = nullptr,
    const int32_t *__restrict__ tok = nullptr, const int32_t *__restrict__ par = nullptr, const float *__restrict__ ew = nullptr,
    const float *__restrict__ p0 = nullptr, int split = 16) {
  // ew != NULL (beam search, two decoder layers, from the second step on): the row's pre-activations are put together here,
  // g = ew[tok[row]] + p0[parent row] (dec_beam_kernel), and h_prev / c_prev are the PARENT row's (hprev / cprev then hold
  // the previous step's new states, one row per beam, pitch ldh / H)
  static_assert(NBM == 1, "one decoder row per workgroup");
  extern __shared__ float sm[];   // q[H] | w[Tp] | part[16 * max(Tp, H)]
  const int Tp = (T + 3) & ~3;
  float *q = sm, *w = q + H, *part = w + Tp;
  // workgroup -> decoder row: workgroups go to the 8 XCDs round robin, and the `beam` rows of a clip read the same key projection
  // and memory (2 T H floats): rows of one clip are given to workgroups of one XCD, so that a clip's data sits in one L2
  // (4 MB each) instead of `beam` of them
  long r = blockIdx.x;
  {
    const int nclips = (int)gridDim.x / beam, full = (nclips / 8) * 8 * beam, id = blockIdx.x;
    if (id < full) {
      const int slot = id >> 3;
      r = (long)((id & 7) + 8 * (slot / beam)) * beam + slot % beam;
    }
  }
  const int b = (int)(r / beam), t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int vl = min(max(valid_len[b], 0), T);
  const float inv = 1.0f / sqrtf((float)H);
  DEC_STAMP(0);
  // ---- cell 0 ----
  for (int u = t; u < H; u += kBeamThreads) {
    const long rp = ew ? (long)b * beam + par[r] : r;
    float ga, gb, gc, gd;
    if (ew) {
      const int word = tok[r];
      const float *e = ew + (long)(word > 0 ? word : 0) * 4 * H, *q0 = p0 + rp * 4 * H;
      ga = e[u] + q0[u]; gb = e[H + u] + q0[H + u]; gc = e[2 * H + u] + q0[2 * H + u]; gd = e[3 * H + u] + q0[3 * H + u];
    } else {
      const float *g = g0 + r * 4 * H;
      ga = g[u]; gb = g[H + u]; gc = g[2 * H + u]; gd = g[3 * H + u];
    }
    float v;
    if (lstm) {
      const float ig = sigm(ga), fg = sigm(gb), gg = tanhf(gc), og = sigm(gd);
      const float c2 = fg * cprev[rp * H + u] + ig * gg;
      cn[r * H + u] = c2;
      v = og * tanhf(c2);
    } else {
      const float rg = sigm(ga), zg = sigm(gb);
      const float ng = tanhf(gc + rg * gd);
      v = (1.f - zg) * ng + zg * hprev[rp * ldh + u];
    }
    hn[r * H + u] = v;
    x1[xidx(r, u, ldx1)] = v;
    q[u] = v * inv;
  }
  __syncthreads();
  DEC_STAMP(1);
  // ---- scores: thread = (slice of H, four source steps); HG partial sums per step ----
  const int S4 = (vl + 3) >> 2, CG = step_groups(S4, split), HG = kBeamThreads / CG;
  {
    const int hg = t / CG, hn_ = (H + HG - 1) / HG, h0 = hg * hn_, h1 = min(H, h0 + hn_);
    const float *kp = kpT + (long)b * H * Tp;
    for (int s4 = t % CG; s4 < S4; s4 += CG) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int h = h0;
      for (; h + 16 <= h1; h += 16) {           // 16 x 16 bytes in flight
        float4 kv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) kv[i] = *(const float4 *)(kp + (long)(h + i) * Tp + 4 * s4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float qv = q[h + i];
          acc.x = fmaf(kv[i].x, qv, acc.x); acc.y = fmaf(kv[i].y, qv, acc.y); acc.z = fmaf(kv[i].z, qv, acc.z); acc.w = fmaf(kv[i].w, qv, acc.w);
        }
      }
      for (; h < h1; ++h) {
        const float4 kv = *(const float4 *)(kp + (long)h * Tp + 4 * s4);
        const float qv = q[h];
        acc.x = fmaf(kv.x, qv, acc.x); acc.y = fmaf(kv.y, qv, acc.y); acc.z = fmaf(kv.z, qv, acc.z); acc.w = fmaf(kv.w, qv, acc.w);
      }
      *(float4 *)(part + hg * Tp + 4 * s4) = acc;
    }
  }
  __syncthreads();
  DEC_STAMP(2);
  // ---- the HG partial sums of every step (all threads), then the masked softmax in one wave (masked -> -1e18, weights * mask) ----
  for (int s = t; s < T; s += kBeamThreads) {
    float a = kNeg;
    if (s < vl) {
      a = part[s];
      for (int g = 1; g < HG; ++g) a += part[g * Tp + s];
    }
    w[s] = a;
  }
  __syncthreads();
  if (wid == 0) {
    float mx = -INFINITY;
    for (int s = lane; s < T; s += 64) mx = fmaxf(mx, w[s]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < T; s += 64) {
      const float e = expf(w[s] - mx);
      w[s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float rs = 1.0f / sum;
    for (int s = lane; s < T; s += 64) {
      const float v = s < vl ? w[s] * rs : 0.f;
      w[s] = v;
      if (wsave) wsave[r * T + s] = v;      // training keeps the attention weights
    }
  }
  __syncthreads();
  DEC_STAMP(3);
  // ---- context: thread = (slice of the valid steps, four units); SG partial sums per unit; weights beyond valid_len are 0 ----
  const int U4 = H >> 2, CG2 = step_groups(U4, split), SG = kBeamThreads / CG2;
  {
    const int sg = t / CG2, sn = (vl + SG - 1) / SG, s0 = sg * sn, s1 = min(vl, s0 + sn);
    const float *mv = mem + (long)b * T * H;
    for (int u4 = t % CG2; u4 < U4; u4 += CG2) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = s0;
      for (; s + 16 <= s1; s += 16) {
        float4 m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = *(const float4 *)(mv + (long)(s + i) * H + 4 * u4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float wv = w[s + i];
          acc.x = fmaf(wv, m[i].x, acc.x); acc.y = fmaf(wv, m[i].y, acc.y); acc.z = fmaf(wv, m[i].z, acc.z); acc.w = fmaf(wv, m[i].w, acc.w);
        }
      }
      if (s < s1) {                              // the rest of the slice, still all in flight at once
        float4 m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = s + i < s1 ? *(const float4 *)(mv + (long)(s + i) * H + 4 * u4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float wv = s + i < s1 ? w[s + i] : 0.f;
          acc.x = fmaf(wv, m[i].x, acc.x); acc.y = fmaf(wv, m[i].y, acc.y); acc.z = fmaf(wv, m[i].z, acc.z); acc.w = fmaf(wv, m[i].w, acc.w);
        }
      }
      *(float4 *)(part + sg * H + 4 * u4) = acc;
    }
  }
  __syncthreads();
  for (int u = t; u < H; u += kBeamThreads) {
    float a = part[u];
    for (int g = 1; g < SG; ++g) a += part[g * H + u];
    ctx[r * H + u] = a;
    x1[xidx(r, H + u, ldx1)] = a;
  }
  DEC_STAMP(4);
}

// (B,T,H) -> (B,H,Tp), Tp = T rounded up to 4: the key projection as the attention kernel reads it (16-byte loads along T)
__global__ void transpose_bth_kernel(const float *__restrict__ src, float *__restrict__ dst, int T, int H) {
  __shared__ float tile[32][33];
  const int Tp = (T + 3) & ~3;
  const int b = blockIdx.z, t0 = blockIdx.y * 32, h0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) tile[i][tx] = (t0 + i < T && h0 + tx < H) ? src[((long)b * T + t0 + i) * H + h0 + tx] : 0.f;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (h0 + i < H && t0 + tx < Tp) dst[((long)b * H + h0 + i) * Tp + t0 + tx] = tile[tx][i];
}

// Wave-wide best of per-lane (value, index) pairs - value descending, then index ascending - in two DPP reductions of one
// instruction per stage (row shifts + row broadcasts, result in lane 63): the maximum value, then the minimum index among the
// lanes that hold it.  A lane without a candidate passes (-inf, 0xffffffff).
__device__ __forceinline__ void wave_best(float v, unsigned ix, float &wv, unsigned &wi) {
  float m = v;
#define TN_DPP_F(ctrl, rmask) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), ctrl, rmask, 0xf, false)));
  TN_DPP_F(0x111, 0xf) TN_DPP_F(0x112, 0xf) TN_DPP_F(0x114, 0xf) TN_DPP_F(0x118, 0xf)   // row_shr:1,2,4,8 -> lane 15 of a row
  TN_DPP_F(0x142, 0xa) TN_DPP_F(0x143, 0xc)                                             // row_bcast:15 / :31 -> lane 63
#undef TN_DPP_F
  wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
  unsigned c = v == wv ? ix : 0xffffffffu;
#define TN_DPP_U(ctrl, rmask) c = min(c, (unsigned)__builtin_amdgcn_update_dpp((int)c, (int)c, ctrl, rmask, 0xf, false));
  TN_DPP_U(0x111, 0xf) TN_DPP_U(0x112, 0xf) TN_DPP_U(0x114, 0xf) TN_DPP_U(0x118, 0xf)
  TN_DPP_U(0x142, 0xa) TN_DPP_U(0x143, 0xc)
#undef TN_DPP_U
  wi = (unsigned)__builtin_amdgcn_readlane((int)c, 63);
}

// The `count` largest of n <= 256 elements held four per lane in registers (x[i], index xi[i]; -inf / 0xffffffff where the lane
// has none), descending, ties -> lowest index: one wave, no barrier, nothing re-read.  A round is wave_best plus, in the
// winning lane only, dropping the element.  out_val / out_idx (LDS) receive the winners.
__device__ __forceinline__ void wave_topk_regs(float (&x)[4], unsigned (&xi)[4], int count, float *out_val, int *out_idx) {
  float bv;
  unsigned bix;
  int bpos;
  auto scan = [&]() {
    bv = -INFINITY; bix = 0xffffffffu; bpos = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool better = (x[i] > bv) | ((x[i] == bv) & (xi[i] < bix));     // (no short circuit: straight-line selects)
      bv = better ? x[i] : bv; bix = better ? xi[i] : bix; bpos = better ? i : bpos;
    }
  };
  scan();
  for (int k = 0; k < count; ++k) {
    float wv;
    unsigned wi;
    wave_best(bv, bix, wv, wi);
    const bool won = (bpos >= 0) & (bix == wi) & (bv == wv);       // this lane owned the winner: record, drop it, look again
    if (won) {
      out_idx[k] = (int)wi;
      out_val[k] = wv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool drop = won & (i == bpos);
      x[i] = drop ? -INFINITY : x[i];
      xi[i] = drop ? 0xffffffffu : xi[i];
    }
    scan();
  }
}
// The same over arr[0..n) in LDS for any n, element p has index idx[p] (idx != NULL) or base + p: a lane owns the elements
// p = lane mod 64 and rescans them when it has won (the dropped element becomes -inf in LDS).
__device__ __forceinline__ void wave_topk(float *arr, const int *idx, int base, int n, int count, float *out_val, int *out_idx, int lane) {
  if (n <= 256) {
    float x[4];
    unsigned xi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = lane + 64 * i;
      x[i] = p < n ? arr[p] : -INFINITY;
      xi[i] = p < n ? (unsigned)(idx ? idx[p] : base + p) : 0xffffffffu;
    }
    wave_topk_regs(x, xi, count, out_val, out_idx);
    return;
  }
  float bv;
  unsigned bix;
  int bpos;
  auto scan = [&]() {
    bv = -INFINITY; bix = 0xffffffffu; bpos = -1;
    for (int p = lane; p < n; p += 64) {
      const float v = arr[p];
      const unsigned ix = (unsigned)(idx ? idx[p] : base + p);
      if (v > bv || (v == bv && ix < bix)) { bv = v; bix = ix; bpos = p; }
    }
  };
  scan();
  for (int k = 0; k < count; ++k) {
    float wv;
    unsigned wi;
    wave_best(bv, bix, wv, wi);
    if (bpos >= 0 && bix == wi && bv == wv) {
      out_idx[k] = (int)wi;
      out_val[k] = wv;
      arr[bpos] = -INFINITY;
      scan();
    }
  }
}

// Beam-search step, last launch, one workgroup per source clip:
//   last decoder cell's gates on g1 (R rows of ldg floats, the first 4H of a row; column layout as in dec_attention_kernel;
//   h_prev / c_prev are the clip's rows of x1[:, 2H:3H] / c1cur) -> projection to the vocabulary (Wp^T streamed once for all
//   beams, 4 partial sums over K) -> per beam row, inside one wave: logits, log-sum-exp, length-penalised candidates and the
//   row's `beam` best -> top-`beam` over [the rows' bests | finished] (the same order as a top-`beam` over [beam*V | finished]:
//   value descending, ties to the lowest index) -> bookkeeping, the step's (parent, word) back-pointers (the token prefixes are
//   rebuilt once at the end, beam_samples_kernel) -> the NEXT step's inputs, re-gathered by parent beam.  Two forms:
//     tok_out == NULL: x0 = [embed(word), ctx[parent], h0[parent]], c0cur; the first cell's gate GEMM is the next step's first launch;
//     tok_out != NULL (two decoder layers): the first cell's pre-activations are linear in x0 and the next step's attention
//       kernel puts them together itself, g0 = ew[word] + p0[parent] with ew = embed . W0[:, 0:E]^T (V,4H; once per model) and
//       p0 = [h0, ctx] . W0[:, E:]^T + b0.  p0 only needs what the attention kernel left in x1, so it is computed by EXTRA
//       workgroups of this launch (blockIdx >= nclips, four 16 x 16 tiles each, lat_tile_f32) on the CUs the clips do not use:
//       3 launches per step, and the gate GEMM on the critical path stays the last cell's.  This kernel then only hands on
//       the words and the parents.
//   Always: x1[:, 2H:3H] = h1[parent], c1cur (LSTM).
struct LatGemmArgs {
  const float *X, *W, *bias;
  float *Y;
  int ldx, ldw, ldy, M, N, K;
};
template <int NBM>
__global__ __launch_bounds__(kBeamThreads) void dec_beam_kernel(
    const float *__restrict__ g1, int ldg, float *__restrict__ x1, int K1, int lstm, float *__restrict__ c1cur,
    const float *__restrict__ wpT, const float *__restrict__ bp, const float *__restrict__ h0n,
    const float *__restrict__ ctx, const float *__restrict__ c0n, float *__restrict__ x0, float *__restrict__ c0cur,
    const float *__restrict__ emb, int H, int E, int V, int beam, int step, float lp, float prev_lp, int eos,
    float *__restrict__ scores, int32_t *__restrict__ alive, int32_t *__restrict__ vlen,
    int32_t *__restrict__ bp_par, int32_t *__restrict__ bp_word, int R, int32_t *__restrict__ any_alive,
    float *__restrict__ hstate, int32_t *__restrict__ parent_out, int32_t *__restrict__ tok_out, int nclips, LatGemmArgs gm, int split) {
  // hstate != NULL: use_residual (gnmt.py:394-395) - the projection sees h + the cell's input x1[:, 0:H], the recurrent
  // state stays h and goes through this (R,H) scratch; parent_out: the chosen parent beam of every row, for the states of
  // the decoder layers between the first and the last one (num_layers > 2)
  extern __shared__ float sm[];   // h1n[H][NP] | c1n[H][NP] | logits[beam*Vp] (the candidates in place) | part[KQ*beam*Vp]
  const int b = blockIdx.x, t = threadIdx.x, K0 = E + 2 * H;     // (K1: pitch of x1 = [input of the last cell (2H) | h_prev])
  constexpr int NP = (NBM + 3) & ~3;   // LDS pitch of one k: NBM beam rows padded to 16-byte multiples
  float *h1n = sm, *c1n = h1n + NP * H, *logits = c1n + NP * H, *part = logits + beam * ((V + 3) & ~3);
  __shared__ float sel_val[16], o_score[16], m_val[16 * 16 + 16];
  __shared__ int sel_idx[16], sel_par[16], sel_word[16], o_alive[16], o_vlen[16], m_idx[16 * 16 + 16];
  const int lane = t & 63, wid = t >> 6;
  if (b >= nclips) {      // p0 tiles for the next step (see above); no output that this launch's clips read
    const int tile = (b - nclips) * 4 + (wid >> 2), ntn = gm.N / 16;
    if (gm.ldx < 0)
      lat_tile_f32<false, true>(gm.X, -4 * gm.ldx, gm.W, gm.ldw, gm.bias, gm.Y, gm.ldy, gm.M, gm.N, gm.K, (tile / ntn) * 16, (tile % ntn) * 16, wid & 3,
                                lane, (float (*)[64][4])(sm + (wid >> 2) * (4 * 64 * 4)));
    else
      lat_tile_f32(gm.X, gm.ldx, gm.W, gm.ldw, gm.bias, gm.Y, gm.ldy, gm.M, gm.N, gm.K, (tile / ntn) * 16, (tile % ntn) * 16, wid & 3, lane,
                   (float (*)[64][4])(sm + (wid >> 2) * (4 * 64 * 4)));
    return;
  }
  // the old state of the beam row this wave will own (loaded now, used after the projection)
  const int rowk = wid < beam ? wid : 0;
  const int al = alive[b * beam + rowk], ovl = vlen[b * beam + rowk];
  const float osc = scores[b * beam + rowk];
  DEC_STAMP(8);
  // ---- cell 1 ----
  for (int idx = t; idx < NBM * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    float v = 0.f, c2 = 0.f;
    if (k < beam) {
      const long r = (long)b * beam + k;
      const float *g = g1 + r * ldg;
      if (lstm) {
        const float ig = sigm(g[u]), fg = sigm(g[H + u]), gg = tanhf(g[2 * H + u]), og = sigm(g[3 * H + u]);
        c2 = fg * c1cur[r * H + u] + ig * gg;
        v = og * tanhf(c2);
      } else {
        const float rg = sigm(g[u]), zg = sigm(g[H + u]);
        const float ng = tanhf(g[2 * H + u] + rg * g[3 * H + u]);
        v = (1.f - zg) * ng + zg * x1[xidx(r, 2 * H + u, K1)];
      }
      if (hstate) {
        hstate[r * H + u] = v;
        v += x1[xidx(r, u, K1)];
      }
    }
    h1n[u * NP + k] = v;       // k-major: a projection thread reads its NBM rows with one or two wide LDS loads
    c1n[u * NP + k] = c2;
  }
  __syncthreads();
  DEC_STAMP(9);
  // ---- projection: thread = (slice of K, four vocabulary columns), see step_groups; KQ partial sums per logit ----
  const int Vp = (V + 3) & ~3, V4 = Vp >> 2, CGv = step_groups(V4, split), KQ = kBeamThreads / CGv;
  {
    constexpr int NLD = NBM <= 5 ? 16 : NBM <= 8 ? 8 : 2;      // 16-byte loads in flight (registers: NBM float4 sums beside them)
    const int kq = t / CGv, kn = (H + KQ - 1) / KQ, k0 = kq * kn, k1 = min(H, k0 + kn);
    for (int v4 = t % CGv; v4 < V4; v4 += CGv) {
      float4 acc[NBM];
      const float4 bz = kq == 0 ? *(const float4 *)(bp + 4 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);   // (bp padded to Vp)
#pragma unroll
      for (int q = 0; q < NBM; ++q) acc[q] = bz;
      int k = k0;
      for (; k + NLD <= k1; k += NLD) {
        float4 w[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) w[i] = *(const float4 *)(wpT + (long)(k + i) * Vp + 4 * v4);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
#pragma unroll
          for (int q = 0; q < NBM; ++q) {
            const float hv = h1n[(k + i) * NP + q];
            acc[q].x = fmaf(w[i].x, hv, acc[q].x); acc[q].y = fmaf(w[i].y, hv, acc[q].y);
            acc[q].z = fmaf(w[i].z, hv, acc[q].z); acc[q].w = fmaf(w[i].w, hv, acc[q].w);
          }
      }
      for (; k < k1; ++k) {
        const float4 w = *(const float4 *)(wpT + (long)k * Vp + 4 * v4);
#pragma unroll
        for (int q = 0; q < NBM; ++q) {
          const float hv = h1n[k * NP + q];
          acc[q].x = fmaf(w.x, hv, acc[q].x); acc[q].y = fmaf(w.y, hv, acc[q].y);
          acc[q].z = fmaf(w.z, hv, acc[q].z); acc[q].w = fmaf(w.w, hv, acc[q].w);
        }
      }
#pragma unroll
      for (int q = 0; q < NBM; ++q)
        if (q < beam) *(float4 *)(part + (long)(kq * beam + q) * Vp + 4 * v4) = acc[q];
    }
  }
  __syncthreads();
  DEC_STAMP(10);
  // ---- logits = the KQ partial sums (the first one starts from the bias), all threads ----
  for (int c = t; c < beam * V; c += kBeamThreads) {
    const int k = c / V, v = c - k * V;
    float x = part[k * Vp + v];
    for (int g = 1; g < KQ; ++g) x += part[(g * beam + k) * Vp + v];
    logits[k * Vp + v] = x;
  }
  __syncthreads();
  // ---- wave k owns beam row k: log-sum-exp, length-penalised candidates and the row's `beam` best (V <= 256: in registers) ----
  if (wid < beam) {
    const int k = wid;
    float *z = logits + k * Vp;
    if (V <= 256) {
      float x[4];
      unsigned xi[4];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = lane + 64 * i;
        x[i] = v < V ? z[v] : -INFINITY;
        xi[i] = v < V ? (unsigned)(k * V + v) : 0xffffffffu;
        mx = fmaxf(mx, x[i]);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < V) sum += expf(x[i] - mx);
      sum = wave_sum(sum);
      const float lse = mx + logf(sum);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < V) x[i] = al ? (osc * prev_lp + (x[i] - lse)) / lp : kNeg;
      wave_topk_regs(x, xi, beam, m_val + k * beam, m_idx + k * beam);
    } else {
      float mx = -INFINITY;
      for (int v = lane; v < V; v += 64) mx = fmaxf(mx, z[v]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int v = lane; v < V; v += 64) sum += expf(z[v] - mx);
      sum = wave_sum(sum);
      const float lse = mx + logf(sum);
      for (int v = lane; v < V; v += 64) {
        const float logp = z[v] - lse;
        z[v] = al ? (osc * prev_lp + logp) / lp : kNeg;
      }
      wave_topk(z, nullptr, k * V, V, beam, m_val + k * beam, m_idx + k * beam, lane);
    }
    if (lane == 0) {
      o_alive[k] = al; o_vlen[k] = ovl; o_score[k] = osc;
      m_val[beam * beam + k] = al ? kNeg : osc;           // a finished beam competes with its own score
      m_idx[beam * beam + k] = beam * V + k;
    }
  }
  __syncthreads();
  DEC_STAMP(12);
  // ---- top-`beam` of the clip: the rows' bests and the finished beams; the same wave then does the bookkeeping ----
  if (wid == 0) {
    // every candidate counts the candidates that beat it (value, then lower index): the ones beaten by fewer than `beam` are
    // the winners and the count is their place - no dependent rounds.  Up to 64 candidates: one per lane, the others' values
    // come through v_readlane (branch-free); more (beam >= 8): broadcast LDS reads.
    const int n2 = beam * beam + beam;
    if (n2 <= 64) {
      const float v = lane < n2 ? m_val[lane] : -INFINITY;
      const int ix = lane < n2 ? m_idx[lane] : 0x7fffffff;
      int rank = 0;
      for (int j = 0; j < n2; ++j) {
        const float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
        const int ij = __builtin_amdgcn_readlane(ix, j);
        rank += (int)(vj > v) | ((int)(vj == v) & (int)(ij < ix));
      }
      if (lane < n2 && rank < beam) { sel_val[rank] = v; sel_idx[rank] = ix; }
    } else {
      for (int p = lane; p < n2; p += 64) {
        const float v = m_val[p];
        const int ix = m_idx[p];
        int rank = 0;
        for (int j = 0; j < n2; ++j) {
          const float vj = m_val[j];
          const int ij = m_idx[j];
          rank += (int)(vj > v) | ((int)(vj == v) & (int)(ij < ix));
        }
        if (rank < beam) { sel_val[rank] = v; sel_idx[rank] = ix; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
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
    if (parent_out) parent_out[b * beam + t] = bid;
    bp_par[(long)(step - 1) * R + b * beam + t] = bid;
    bp_word[(long)(step - 1) * R + b * beam + t] = word;
    if (al) atomicOr(any_alive, 1);
  }
  __syncthreads();
  DEC_STAMP(14);
  // ---- next step's inputs, states re-gathered by parent beam ----
  if (tok_out) {
    if (t < beam) tok_out[b * beam + t] = sel_word[t];
  } else {
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
  }
  for (int idx = t; idx < beam * H; idx += kBeamThreads) {
    const int k = idx / H, u = idx - k * H;
    const long r = (long)b * beam + k, pr = (long)b * beam + sel_par[k];
    x1[xidx(r, 2 * H + u, K1)] = hstate ? hstate[pr * H + u] : h1n[u * NP + sel_par[k]];
    if (lstm) {
      c1cur[r * H + u] = c1n[u * NP + sel_par[k]];
      if (!tok_out) c0cur[r * H + u] = c0n[pr * H + u];
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
                                    int R, int H, float *__restrict__ out = nullptr) {
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
  if (out) out[id] = h1n[id] + x1[r * 3 * H + u];     // use_residual: what the projection sees
}

// A decoder cell between the first and the last one (num_layers > 2; gnmt.py:385-397): gates on the stacked
// pre-activations g (R,4H) of x = [out of the layer below, attention, h_prev]; the new state goes to hn / cn, the layer's
// output (h, or h + the layer's input with use_residual) and the attention vector to the next layer's input.
__global__ void dec_mid_cell_kernel(const float *__restrict__ g, const float *__restrict__ x, int lstm,
                                    const float *__restrict__ ccur, float *__restrict__ hn, float *__restrict__ cn,
                                    float *__restrict__ xnext, int residual, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const float *gr = g + r * 4 * H;
  float h;
  if (lstm) {
    const float ig = sigm(gr[u]), fg = sigm(gr[H + u]), gg = tanhf(gr[2 * H + u]), og = sigm(gr[3 * H + u]);
    const float c2 = fg * ccur[id] + ig * gg;
    cn[id] = c2;
    h = og * tanhf(c2);
  } else {
    const float rg = sigm(gr[u]), zg = sigm(gr[H + u]);
    const float ng = tanhf(gr[2 * H + u] + rg * gr[3 * H + u]);
    h = (1.f - zg) * ng + zg * x[r * 3 * H + 2 * H + u];
  }
  hn[id] = h;
  xnext[r * 3 * H + u] = residual ? h + x[r * 3 * H + u] : h;
  xnext[r * 3 * H + H + u] = x[r * 3 * H + H + u];
}

// The recurrent state of such a layer for the next step: x[:, 2H:3H] = hn[row or its parent beam], ccur likewise (LSTM).
// parent == NULL (teacher forcing, or hn / cn = the encoder's states of the clip with `beam` rows per clip): by row / clip.
__global__ void dec_mid_state_kernel(const int32_t *__restrict__ parent, const float *__restrict__ hn,
                                     const float *__restrict__ cn, float *__restrict__ x, float *__restrict__ ccur,
                                     int from_clip, int beam, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const long pr = from_clip ? r / beam : parent ? (r / beam) * beam + parent[r] : r;
  x[r * 3 * H + 2 * H + u] = hn[pr * H + u];
  if (cn) ccur[id] = cn[pr * H + u];
}

__global__ void add_inplace_kernel(float *__restrict__ y, const float *__restrict__ x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// first step's inputs: x0 = [embed(bos), 0, h0 of the clip], x1[:, 2H:3H] = h1 of the clip, cell states (LSTM)
__global__ void dec_init_kernel(const float *__restrict__ emb, int bos, const float *__restrict__ h0c,
                                const float *__restrict__ h1c, const float *__restrict__ c0c,
                                const float *__restrict__ c1c, float *__restrict__ x0, float *__restrict__ x1,
                                float *__restrict__ c0cur, float *__restrict__ c1cur, int beam, int H, int E, int K1) {
  const int r = blockIdx.x, b = r / beam, K0 = E + 2 * H;
  for (int i = threadIdx.x; i < K0; i += blockDim.x)
    x0[(long)r * K0 + i] = i < E ? emb[(long)bos * E + i] : i < E + H ? 0.f : h0c[(long)b * H + i - E - H];
  for (int u = threadIdx.x; u < H; u += blockDim.x) {
    x1[xidx(r, 2 * H + u, K1)] = h1c[(long)b * H + u];
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

// The token prefixes of the final beams from the steps' (parent, word) back-pointers: sample[row][s] for s = steps .. 1 is the
// word chosen at step s by the row's ancestor of that step (-1 where a finished beam was carried over), sample[row][0] = bos -
// what copying the parent's prefix and appending the word at every step (BeamSearchSampler [EXT]) leaves behind.  One
// workgroup per clip; the clip's pointers pass through LDS in chunks of kSampChunk steps, newest first.
constexpr int kSampChunk = 256;
__global__ __launch_bounds__(256) void beam_samples_kernel(const int32_t *__restrict__ bp_par, const int32_t *__restrict__ bp_word,
                                                           int32_t *__restrict__ samples, int L, int steps, int R, int beam, int bos) {
  __shared__ int32_t par[kSampChunk * 16], word[kSampChunk * 16];
  const int b = blockIdx.x, t = threadIdx.x;
  int cur = t;
  int32_t *dst = samples + ((long)b * beam + (t < beam ? t : 0)) * L;
  for (int hi = steps; hi > 0; hi -= kSampChunk) {
    const int lo = hi > kSampChunk ? hi - kSampChunk : 0, n = hi - lo;
    __syncthreads();
    for (int i = t; i < n * beam; i += 256) {
      const int s = i / beam, k = i - s * beam;
      par[i] = bp_par[(long)(lo + s) * R + b * beam + k];
      word[i] = bp_word[(long)(lo + s) * R + b * beam + k];
    }
    __syncthreads();
    if (t < beam)
      for (int s = n - 1; s >= 0; --s) {
        dst[lo + s + 1] = word[s * beam + cur];
        cur = par[s * beam + cur];
      }
  }
  if (t < beam) dst[0] = bos;
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

// ---- training step of the captioner (GRU cells) ------------------------------------------------------
// Teacher-forced forward with everything the backward pass needs kept per step (step-major (L,B,.) arrays), the
// gradient of the token-averaged masked cross-entropy, back-propagation through the decoder steps, the attention, the
// key projection and the two encoder layers, Adam.  Reference: train_gnmt.py:310,328-337.

// step inputs of teacher forcing, strided sources: x0 = [embed(tok), ctx_prev | 0, h0_prev], x1[:, 2H:3H] = h1_prev
__global__ void trn_prep_kernel(const float *__restrict__ emb, const int32_t *__restrict__ tgt, int ld, int col,
                                const float *__restrict__ ctx_prev, int ldc, const float *__restrict__ h0_prev, int ld0,
                                const float *__restrict__ h1_prev, int ld1, float *__restrict__ x0, float *__restrict__ x1,
                                int H, int E) {
  const int r = blockIdx.x, K0 = E + 2 * H, K1 = 3 * H;
  const int tok = tgt[(long)r * ld + col];
  const float *e = emb + (long)(tok > 0 ? tok : 0) * E;
  for (int i = threadIdx.x; i < K0; i += blockDim.x)
    x0[(long)r * K0 + i] = i < E ? e[i] : i < E + H ? (ctx_prev ? ctx_prev[(long)r * ldc + i - E] : 0.f) : h0_prev[(long)r * ld0 + i - E - H];
  for (int u = threadIdx.x; u < H; u += blockDim.x) x1[(long)r * K1 + 2 * H + u] = h1_prev[(long)r * ld1 + u];
}

// d loss / d logits for loss = sum over valid (b, t) of -log softmax(logits[b, t])[label] / (number of valid tokens);
// logits (B, L, V) as decode_seq lays them out, dlogits STEP-major (L, B, V) like every other per-step array here
__global__ __launch_bounds__(256) void trn_ce_bwd_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels,
                                                         int ldl, const int32_t *__restrict__ valid_len, int B, int L, int V,
                                                         float *__restrict__ dlogits, float *__restrict__ loss_rows) {
  __shared__ float red[256];
  const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  int ntok = 0;
  for (int q = 0; q < B; ++q) ntok += min(max(valid_len[q], 0), L);
  const float *z = logits + ((long)b * L + i) * V;
  float *d = dlogits + ((long)i * B + b) * V;
  const bool valid = i < valid_len[b];
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
  const float lse = mx + logf(red[0]);
  const int lab = labels[(long)b * ldl + i];
  const float sc = valid && ntok > 0 ? 1.0f / (float)ntok : 0.f;
  for (int v = t; v < V; v += 256) d[v] = (expf(z[v] - lse) - (v == lab ? 1.f : 0.f)) * sc;
  if (t == 0) loss_rows[(long)i * B + b] = valid ? (lse - z[lab]) * sc : 0.f;     // summed over (i, b) = the loss
}

// GRU cell backward on the stacked pre-activations g (R,4H) = [r, z, n_i2h, n_h2h]:
//   dh = sum of up to four addends (each (R,H) with its own row stride, null = absent)
//   dg (R,4H) = [d r, d z, d n, d n * r]   (the stacked weight matrix routes them to both branches)
//   dhz (R,H) = dh * z                     (the direct path to h_prev; the rest comes back through dg x W)
__global__ void trn_gru_bwd_kernel(const float *__restrict__ g, const float *__restrict__ hprev, int ldh,
                                   const float *a0, int l0, const float *a1, int l1, const float *a2, int l2,
                                   const float *a3, int l3, float *__restrict__ dg, float *dhz, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const float *q = g + r * 4 * H;
  const float rg = sigm(q[u]), zg = sigm(q[H + u]), anh = q[3 * H + u];
  const float ng = tanhf(q[2 * H + u] + rg * anh);
  const float hp = hprev[r * ldh + u];
  float dh = 0.f;
  if (a0) dh += a0[r * l0 + u];
  if (a1) dh += a1[r * l1 + u];
  if (a2) dh += a2[r * l2 + u];
  if (a3) dh += a3[r * l3 + u];
  const float dn = dh * (1.f - zg) * (1.f - ng * ng);
  float *o = dg + r * 4 * H;
  o[u] = dn * anh * rg * (1.f - rg);
  o[H + u] = dh * (hp - ng) * zg * (1.f - zg);
  o[2 * H + u] = dn;
  o[3 * H + u] = dn * rg;
  dhz[id] = dh * zg;
}

// LSTM cell backward on the stacked pre-activations g (R,4H) = [i, f, g, o]: dc = dc_carry + dh o (1 - tanh^2 c);
// dg (R,4H); dcz = dc f (to c_prev); there is no direct path to h_prev (dhz = 0)
__global__ void trn_lstm_bwd_kernel(const float *__restrict__ g, const float *__restrict__ cprev, const float *__restrict__ cnew,
                                    const float *a0, int l0, const float *a1, int l1, const float *a2, int l2,
                                    const float *a3, int l3, const float *dcc, float *__restrict__ dg, float *dhz, float *dcz,
                                    int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const float *q = g + r * 4 * H;
  const float ig = sigm(q[u]), fg = sigm(q[H + u]), gg = tanhf(q[2 * H + u]), og = sigm(q[3 * H + u]);
  const float tc = tanhf(cnew[id]);
  float dh = 0.f;
  if (a0) dh += a0[r * l0 + u];
  if (a1) dh += a1[r * l1 + u];
  if (a2) dh += a2[r * l2 + u];
  if (a3) dh += a3[r * l3 + u];
  const float dct = (dcc ? dcc[id] : 0.f) + dh * og * (1.f - tc * tc);
  float *o = dg + r * 4 * H;
  o[u] = dct * gg * ig * (1.f - ig);
  o[H + u] = dct * cprev[id] * fg * (1.f - fg);
  o[2 * H + u] = dct * ig * (1.f - gg * gg);
  o[3 * H + u] = dh * tc * og * (1.f - og);
  dhz[id] = 0.f;
  dcz[id] = dct * fg;
}

// Backward of one decoder step's attention (scaled Luong: ctx = sum_t w_t mem_t, w = softmax(s), s_t = q . kp_t, q = h0 / sqrt(H)) for
// one clip: dctx = up to two addends; dw_t = mem_t . dctx; ds = w (dw - w . dw); dq = sum_t ds_t kp_t -> dh0 = dq / sqrt(H).
// The step's share of d mem and d keyproj is NOT accumulated here (that was a read-modify-write of 2 T H floats per clip and
// step behind one workgroup, 123 us per step at config C5): the step only leaves ds (L,B,T) and dctx (L,B,H) behind, and
// trn_att_outer_kernel forms d mem = sum over steps of w (x) dctx and d keyproj = sum of ds (x) q once after the loop.
__global__ __launch_bounds__(kBeamThreads) void trn_att_bwd_kernel(const float *__restrict__ aw, const float *__restrict__ mem,
                                                                   const float *__restrict__ keyproj, const float *c0, int lc0, const float *c1, int lc1,
                                                                   const int32_t *__restrict__ valid_len, float *__restrict__ ds_out,
                                                                   float *__restrict__ dctx_out, float *__restrict__ dh0, int T, int H) {
  extern __shared__ float sm[];    // dctx[H] | ds[T] | part[16][H] | red[16]
  float *dctx = sm, *ds = dctx + H, *part = ds + T, *red = part + 16 * H;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int vl = min(max(valid_len[b], 0), T);
  const float inv = 1.0f / sqrtf((float)H);
  for (int u = t; u < H; u += kBeamThreads) {
    const float d = (c0 ? c0[(long)b * lc0 + u] : 0.f) + (c1 ? c1[(long)b * lc1 + u] : 0.f);
    dctx[u] = d;
    dctx_out[(long)b * H + u] = d;
  }
  __syncthreads();
  const float *w = aw + (long)b * T;
  const float *mv = mem + (long)b * T * H, *kp = keyproj + (long)b * T * H;
  // dw_t = mem_t . dctx: wave `wid` takes the steps wid, wid + 16, ...; a lane owns four units per 256 (16-byte loads along H),
  // eight steps in flight
  for (int s0 = wid; s0 < vl; s0 += 16 * 8) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
    for (int u = 4 * lane; u < H; u += 256) {
      const float4 d = *(const float4 *)(dctx + u);
      float4 m[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s = s0 + 16 * i;
        m[i] = s < vl ? *(const float4 *)(mv + (long)s * H + u) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = fmaf(m[i].x, d.x, fmaf(m[i].y, d.y, fmaf(m[i].z, d.z, fmaf(m[i].w, d.w, a[i]))));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = wave_sum(a[i]);
      if (lane == 0 && s0 + 16 * i < vl) ds[s0 + 16 * i] = v;
    }
  }
  __syncthreads();
  float pw = 0.f;
  for (int s = t; s < vl; s += kBeamThreads) pw += w[s] * ds[s];
  pw = wave_sum(pw);
  if (lane == 0) red[wid] = pw;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dot += red[i];
  for (int s = t; s < T; s += kBeamThreads) {
    const float v = s < vl ? w[s] * (ds[s] - dot) : 0.f;
    ds_out[(long)b * T + s] = v;
    ds[s] = v;                  // each element is read (above) and written by the same thread
  }
  __syncthreads();
  // dq = sum_t ds_t kp_t: thread = (slice of the valid steps, four units), 16 partial sums per unit
  const int U4 = H >> 2, CG = step_groups(U4, 16), SG = kBeamThreads / CG;
  {
    const int sg = t / CG, sn = (vl + SG - 1) / SG, sa = sg * sn, sb = min(vl, sa + sn);
    for (int u4 = t % CG; u4 < U4; u4 += CG) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = sa; s < sb; s += 16) {
        float4 m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = s + i < sb ? *(const float4 *)(kp + (long)(s + i) * H + 4 * u4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dv = s + i < sb ? ds[s + i] : 0.f;
          acc.x = fmaf(dv, m[i].x, acc.x); acc.y = fmaf(dv, m[i].y, acc.y); acc.z = fmaf(dv, m[i].z, acc.z); acc.w = fmaf(dv, m[i].w, acc.w);
        }
      }
      *(float4 *)(part + sg * H + 4 * u4) = acc;
    }
  }
  __syncthreads();
  for (int u = t; u < H; u += kBeamThreads) {
    float a = part[u];
    for (int g = 1; g < SG; ++g) a += part[g * H + u];
    dh0[(long)b * H + u] = a * inv;
  }
}

// d mem (B,T,H) = sum over the `steps` decoder steps of w_step (x) dctx_step, d keyproj = sum of ds_step (x) h0_step / sqrt(H):
// thread = (source step, four units), the steps' factors are (steps, B, T) / (steps, B, H) arrays (h0: pitch ldh per row).
__global__ __launch_bounds__(256) void trn_att_outer_kernel(const float *__restrict__ aw, const float *__restrict__ ds,
                                                            const float *__restrict__ dctx, const float *__restrict__ h0, int ldh,
                                                            float *__restrict__ dmem, float *__restrict__ dkp, int steps, int B, int T, int H) {
  const int b = blockIdx.y, u4 = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= T) return;
  const float inv = 1.0f / sqrtf((float)H);
  for (int u = 4 * u4; u < H; u += 256) {
    float4 am = make_float4(0.f, 0.f, 0.f, 0.f), ak = am;
    for (int l = 0; l < steps; ++l) {
      const long r = (long)l * B + b;
      const float w = aw[r * T + s], d = ds[r * T + s] * inv;
      const float4 c = *(const float4 *)(dctx + r * H + u), q = *(const float4 *)(h0 + r * ldh + u);
      am.x = fmaf(w, c.x, am.x); am.y = fmaf(w, c.y, am.y); am.z = fmaf(w, c.z, am.z); am.w = fmaf(w, c.w, am.w);
      ak.x = fmaf(d, q.x, ak.x); ak.y = fmaf(d, q.y, ak.y); ak.z = fmaf(d, q.z, ak.z); ak.w = fmaf(d, q.w, ak.w);
    }
    *(float4 *)(dmem + ((long)b * T + s) * H + u) = am;
    *(float4 *)(dkp + ((long)b * T + s) * H + u) = ak;
  }
}

// d embedding: rows of the step-major input gradients summed per token, in (step, row) order (deterministic)
__global__ void trn_emb_grad_kernel(const float *__restrict__ dx0, int K0, const int32_t *__restrict__ tgt, int ld, int B,
                                    int L, int E, int V, float *__restrict__ demb) {
  const int v = blockIdx.x;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float a = 0.f;
    for (int i = 0; i < L; ++i)
      for (int b = 0; b < B; ++b) {
        const int tok = tgt[(long)b * ld + i];
        if ((tok > 0 ? tok : 0) == v) a += dx0[((long)i * B + b) * K0 + e];
      }
    demb[(long)v * E + e] = a;
  }
}

// Gluon cell parameters <-> the stacked (4H, in+H) matrix of the step GEMM (GRU: rows r, z = [Wi | Wh], bias bi + bh;
// rows 2H..3H = [Wi_n | 0], bias bi_n; rows 3H..4H = [0 | Wh_n], bias bh_n)
// (LSTM: every row = [Wi | Wh], bias bi + bh)
__global__ void trn_stack_kernel(const float *__restrict__ wi, const float *__restrict__ wh, const float *__restrict__ bi,
                                 const float *__restrict__ bh, int in, int H, int lstm, float *__restrict__ wc,
                                 float *__restrict__ bc) {
  const int row = blockIdx.x, Kc = in + H;
  const int src_i = (lstm || row < 3 * H) ? row : -1, src_h = (lstm || row < 2 * H) ? row : row >= 3 * H ? row - H : -1;
  for (int k = threadIdx.x; k < Kc; k += blockDim.x)
    wc[(long)row * Kc + k] = k < in ? (src_i >= 0 ? wi[(long)src_i * in + k] : 0.f) : (src_h >= 0 ? wh[(long)src_h * H + k - in] : 0.f);
  if (threadIdx.x == 0) bc[row] = (src_i >= 0 ? bi[src_i] : 0.f) + (src_h >= 0 ? bh[src_h] : 0.f);
}
__global__ void trn_unstack_kernel(const float *__restrict__ dwc, const float *__restrict__ dbc, int in, int H, int lstm,
                                   float *__restrict__ dwi, float *__restrict__ dwh, float *__restrict__ dbi,
                                   float *__restrict__ dbh) {
  const int row = blockIdx.x, Kc = in + H;     // row of the Gluon (G*H, .) matrices
  const int ri = row, rh = (lstm || row < 2 * H) ? row : row + H;
  for (int k = threadIdx.x; k < Kc; k += blockDim.x) {
    if (k < in) dwi[(long)row * in + k] = dwc[(long)ri * Kc + k];
    else dwh[(long)row * H + k - in] = dwc[(long)rh * Kc + k];
  }
  if (threadIdx.x == 0) { dbi[row] = dbc[ri]; dbh[row] = dbc[rh]; }
}

__global__ void trn_add2_kernel(const float *__restrict__ a, int la, const float *__restrict__ b, int lb,
                                float *__restrict__ out, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  out[id] = a[r * la + u] + b[r * lb + u];
}

// inverted dropout (gluon nn.Dropout in training mode): mask = keep ? 1/(1-p) : 0 from a counter-based hash (splitmix64),
// so a step's masks depend only on (seed, step counter, element index); y = x * mask
__global__ void trn_dropout_mask_kernel(float *__restrict__ mask, long n, float p, unsigned long long key) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = key + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // uniform [0, 1)
  mask[i] = u < p ? 0.f : 1.0f / (1.0f - p);
}
__global__ void trn_mul_kernel(const float *__restrict__ x, const float *__restrict__ m, float *__restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * m[i];
}

// MXNet Adam [EXT]: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m / (sqrt(v) + eps), lr_t bias-corrected
__global__ void trn_adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                float *__restrict__ v, long n, float lr_t, float b1, float b2, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi, vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  w[i] -= lr_t * mi / (sqrtf(vi) + eps);
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

// a decoder cell between the first and the last one (num_layers > 2)
struct GnmtMid { float *w, *b, *sx, *g, *hn, *cn, *ccur; };

struct tn_gnmt {
  tn_ctx *ctx;
  DevBuf pool;
  // encoder: num_bi_layers bidirectional layers (F | 2H -> 2H), then uni-directional ones (2H | H -> H); gnmt.py:84-111
  std::vector<tn_birnn *> enc;
  int NL, NBI;
  bool residual;            // use_residual (gnmt.py:155-157, 394-395)
  std::vector<float *> hl, cl;   // final states per encoder layer (bi: [fwd, bwd]); the decoder layer i starts from layer i's
  float *seqB;              // second sequence buffer (layers ping-pong between seq0 and seqB; the last one writes mem)
  std::vector<GnmtMid> mid; // decoder layers 1 .. NL-2
  float *hstate;            // last decoder layer's state when use_residual (the LDS copy then holds h + input)
  int32_t *parent;          // parent beam per row (for the states of `mid`)
  int F, H, E, V, maxB, maxT, beam, maxL;
  float *wk, *wp, *bp, *emb;
  // per-call workspace
  int G;                   // gates per cell: 3 GRU, 4 LSTM
  float *seq0, *mem, *keyproj;
  int32_t *vl;
  float *scores;
  int32_t *alive, *vlen, *tok, *samples[2], *flag;
  // fused beam-search step: stacked [i2h | h2h] weights (4H rows each), transposed projection, step buffers
  float *w0c, *b0c, *w1c, *b1c, *wpT, *bpp;   // (wpT, bpp: Wp^T and its bias with rows / length padded to a multiple of 4)
  float *sx0, *sx1, *g0, *g1, *h0n, *ctxn, *c0n, *c0cur, *c1cur, *keyprojT, *h1n, *c1n;
  // two decoder layers: the first cell's gate GEMM leaves the step's critical path (dec_beam_kernel, `tok_out`):
  // w1x (4H, 3H) = the first cell's columns of h0 and ctx in the order of x1 = [h0, ctx, h1_prev] (the last H columns unused),
  // b1x = b0, ew (V, 4H) = embedding . first cell's embedding columns (made on the first search), p0 (R, 4H) the [h0, ctx]
  // share of the next step's pre-activations, h0n2 / c0n2 the second buffers of the first cell's states (a step reads the
  // previous step's by parent row while it writes its own)
  float *w1x, *b1x, *ew, *p0, *h0n2, *c0n2, *w1cp, *sx1p, *w1k4;
  int K1p;                 // pitch of w1cp / w1x / sx1p, see kGemmPitchPad
  bool ew_ready;
  int32_t *bp_par, *bp_word;   // (maxL, R) back-pointers of the search
  int B, T;
};

extern "C" int tn_gnmt_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, int cell_kind,
                                 int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                                 int max_batch, int max_src_len, int beam, int max_length, int flags, tn_gnmt **out);

extern "C" int tn_gnmt_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, int cell_kind,
                              int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                              int max_batch, int max_src_len, int beam, int max_length, tn_gnmt **out) {
  return tn_gnmt_create_ex(ctx, params, n_params, prefix_c, cell_kind, input_size, hidden, embed, vocab, num_layers, num_bi_layers,
                           max_batch, max_src_len, beam, max_length, 0, out);
}

extern "C" int tn_gnmt_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, int cell_kind,
                                 int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                                 int max_batch, int max_src_len, int beam, int max_length, int flags, tn_gnmt **out) {
  TN_REQUIRE(ctx && params && prefix_c && out, "tn_gnmt_create: null argument");
  TN_REQUIRE(cell_kind == TN_RNN_GRU || cell_kind == TN_RNN_LSTM, "tn_gnmt_create: cell_type must be 'gru' or 'lstm'");
  TN_REQUIRE((flags & ~TN_GNMT_USE_RESIDUAL) == 0, "tn_gnmt_create_ex: unknown flag");
  // gnmt.py:78-80 asserts num_bi_layers <= num_layers; with num_bi_layers == num_layers the memory is 2H wide and gluonnlp's
  // Luong-style attention (key width = units = H) refuses it [EXT]; a one-layer decoder has no cell behind the attention
  TN_REQUIRE(num_layers >= 2 && num_layers <= 8 && num_bi_layers >= 0 && num_bi_layers < num_layers,
             "tn_gnmt_create: need 2 <= num_layers <= 8 and 0 <= num_bi_layers < num_layers");
  TN_REQUIRE(beam >= 1 && beam <= 16 && vocab >= beam && max_length >= 1, "tn_gnmt_create: bad beam/vocab/max_length");
  TN_REQUIRE(input_size > 0 && hidden > 0 && hidden % 4 == 0 && embed > 0 && max_batch > 0 && max_src_len > 0,
             "tn_gnmt_create: bad shape");
  TN_ON_DEVICE(ctx->device);
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
  g->NL = num_layers; g->NBI = num_bi_layers; g->residual = (flags & TN_GNMT_USE_RESIDUAL) != 0;
  auto fail = [&](int code) { g->pool.release(); for (tn_birnn *e : g->enc) tn_birnn_destroy(e); delete g; return code; };
  const char *sfx[4] = {"i2h_weight", "h2h_weight", "i2h_bias", "h2h_bias"};
  const int H = hidden, G3 = g->G * hidden;   // (G3: gates * hidden, 3H or 4H)
  // encoder layers: "<pre>enc_rnn{i}_{l,r}_*" (bidirectional) / "<pre>enc_rnn{i}_*" -> the "{l,r}0_*" names of tn_birnn
  for (int i = 0, fin = input_size; i < num_layers; ++i) {
    const bool bi = i < num_bi_layers;
    std::vector<tn_param> pl;
    std::vector<std::string> names;
    names.reserve(8);
    for (int d = 0; d < (bi ? 2 : 1); ++d)
      for (int k = 0; k < 4; ++k) {
        const std::string src = pre + "enc_rnn" + std::to_string(i) + "_" + (bi ? (d ? "r_" : "l_") : "") + sfx[k];
        const int64_t n = k == 0 ? (int64_t)G3 * fin : k == 1 ? (int64_t)G3 * H : G3;
        const float *v = get(src, n);
        if (!v) return fail(TN_ERR_MISSING);
        names.push_back(std::string(d ? "r0_" : "l0_") + sfx[k]);
        pl.push_back(tn_param{nullptr, v, n});
      }
    for (size_t k = 0; k < pl.size(); ++k) pl[k].name = names[k].c_str();
    tn_birnn *e = nullptr;
    const int rc = tn_birnn_create(ctx, (tn_rnn_kind)cell_kind, fin, H, pl.data(), (int)pl.size(), "", bi ? 1 : 0, max_batch * max_src_len, &e);
    if (rc) return fail(rc);
    g->enc.push_back(e);
    fin = bi ? 2 * H : H;
  }
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
    std::vector<float> last_w, last_b;      // host copies of the matrix stack() made last
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
      last_w.swap(w); last_b.swap(bv);
      return true;
    };
    if (!stack("dec_rnn0_", embed + H, &g->w0c, &g->b0c)) return fail(TN_ERR_MISSING);
    std::vector<float> w0, b0;
    w0.swap(last_w); b0.swap(last_b);
    if (!stack("dec_rnn" + std::to_string(num_layers - 1) + "_", 2 * H, &g->w1c, &g->b1c)) return fail(TN_ERR_MISSING);
    if (num_layers == 2) {
      const int K0 = embed + 2 * H, K1 = 3 * H;
      // the step GEMMs read 16 operand rows per load instruction: with rows 3H floats apart (3 KiB at H = 256) the sixteen
      // lines of an instruction fall into a quarter of the L2 channels; the rows of the beam search's operands get a pitch
      // that spreads them (tuning: TN_GNMT_PAD=<floats>)
      const char *pe = getenv("TN_GNMT_PAD");
      const int K1p = K1 + (((pe ? std::max(0, atoi(pe)) : kGemmPitchPad) + 3) & ~3);      // (rows stay 16-byte aligned)
      g->K1p = K1p;
      const std::vector<float> &w1 = last_w;
      std::vector<float> w1p((size_t)4 * H * K1p, 0.f);
      for (int row = 0; row < 4 * H; ++row) memcpy(&w1p[(size_t)row * K1p], &w1[(size_t)row * K1], sizeof(float) * K1);
      g->w1cp = g->pool.upload(w1p.data(), w1p.size());
      {   // the same weights k-group-major for the step's gate GEMM (lat_tile_f32<true>)
        std::vector<float> wk((size_t)K1 * 4 * H);
        for (int row = 0; row < 4 * H; ++row)
          for (int k = 0; k < K1; ++k) wk[((size_t)(k >> 2) * 4 * H + row) * 4 + (k & 3)] = w1[(size_t)row * K1 + k];
        g->w1k4 = g->pool.upload(wk.data(), wk.size());
      }
      g->sx1p = g->pool.alloc<float>((size_t)max_batch * beam * K1p);      // (k-group-major in the beam search: K1 x rows floats)
      std::vector<float> wx((size_t)4 * H * K1p, 0.f);
      for (int row = 0; row < 4 * H; ++row) {
        float *d = &wx[(size_t)row * K1p];
        const float *sw = &w0[(size_t)row * K0];
        memcpy(d, sw + embed + H, sizeof(float) * H);        // x1[:, 0:H] = h0   <- x0[:, E+H:]
        memcpy(d + H, sw + embed, sizeof(float) * H);        // x1[:, H:2H] = ctx <- x0[:, E:E+H]
      }
      g->w1x = g->pool.upload(wx.data(), wx.size());
      g->b1x = g->pool.upload(b0.data(), b0.size());
      g->ew = g->pool.alloc<float>((size_t)vocab * 4 * H);
    }
    g->mid.resize(num_layers - 2);
    for (int i = 1; i + 1 < num_layers; ++i)
      if (!stack("dec_rnn" + std::to_string(i) + "_", 2 * H, &g->mid[i - 1].w, &g->mid[i - 1].b)) return fail(TN_ERR_MISSING);
    const float *wp = get(pre + "tgt_proj_weight", (int64_t)vocab * H);
    const int vp = (vocab + 3) & ~3;        // rows of Wp^T padded to 16 bytes (dec_beam_kernel loads four columns at a time)
    std::vector<float> wt((size_t)H * vp, 0.f);
    for (int v = 0; v < vocab; ++v)
      for (int k = 0; k < H; ++k) wt[(size_t)k * vp + v] = wp[(size_t)v * H + k];
    g->wpT = g->pool.upload(wt.data(), wt.size());
    const float *bpv = get(pre + "tgt_proj_bias", vocab);
    std::vector<float> bpad(vp, 0.f);
    if (bpv) memcpy(bpad.data(), bpv, sizeof(float) * vocab);
    g->bpp = g->pool.upload(bpad.data(), bpad.size());
  }
  const size_t BT = (size_t)max_batch * max_src_len, R = (size_t)max_batch * beam;
  g->seq0 = g->pool.alloc<float>(BT * 2 * H); g->mem = g->pool.alloc<float>(BT * H); g->keyproj = g->pool.alloc<float>(BT * H);
  g->seqB = num_layers > 2 ? g->pool.alloc<float>(BT * 2 * H) : nullptr;
  for (int i = 0; i < num_layers; ++i) {
    g->hl.push_back(g->pool.alloc<float>(2 * (size_t)max_batch * H));
    g->cl.push_back(g->pool.alloc<float>(2 * (size_t)max_batch * H));
  }
  for (GnmtMid &m : g->mid) {
    m.sx = g->pool.alloc<float>(R * 3 * H); m.g = g->pool.alloc<float>(R * 4 * H);
    m.hn = g->pool.alloc<float>(R * H); m.cn = g->pool.alloc<float>(R * H); m.ccur = g->pool.alloc<float>(R * H);
  }
  g->hstate = g->residual ? g->pool.alloc<float>(R * H) : nullptr;
  g->parent = g->pool.alloc<int32_t>(R);
  g->vl = g->pool.alloc<int32_t>(max_batch);
  g->samples[0] = g->pool.alloc<int32_t>(R * g->maxL);
  g->bp_par = g->pool.alloc<int32_t>(R * g->maxL); g->bp_word = g->pool.alloc<int32_t>(R * g->maxL);
  g->scores = g->pool.alloc<float>(R); g->alive = g->pool.alloc<int32_t>(R); g->vlen = g->pool.alloc<int32_t>(R);
  g->tok = g->pool.alloc<int32_t>(R); g->flag = g->pool.alloc<int32_t>(1);
  g->h1n = g->pool.alloc<float>(R * H); g->c1n = g->pool.alloc<float>(R * H);
  g->sx0 = g->pool.alloc<float>(R * (embed + 2 * H)); g->sx1 = g->pool.alloc<float>(R * 3 * H);
  g->g0 = g->pool.alloc<float>(R * 4 * H); g->g1 = g->pool.alloc<float>(R * 4 * H);
  if (num_layers == 2) {
    g->p0 = g->pool.alloc<float>(R * 4 * H); g->h0n2 = g->pool.alloc<float>(R * H); g->c0n2 = g->pool.alloc<float>(R * H);
  }
  g->h0n = g->pool.alloc<float>(R * H); g->ctxn = g->pool.alloc<float>(R * H); g->c0n = g->pool.alloc<float>(R * H);
  g->c0cur = g->pool.alloc<float>(R * H); g->c1cur = g->pool.alloc<float>(R * H); g->keyprojT = g->pool.alloc<float>((size_t)max_batch * ((max_src_len + 3) & ~3) * H);
  if (g->pool.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  *out = g;
  return TN_OK;
}

// GNMTEncoder.forward + the attention key projection; keeps mem / states inside the handle.
// mem_out (B,T,H) may be NULL.
extern "C" int tn_gnmt_encode(tn_gnmt *g, const float *src, const int32_t *valid_len, int batch, int steps, float *mem_out) {
  TN_REQUIRE(g && src && valid_len, "tn_gnmt_encode: null argument");
  TN_REQUIRE(batch > 0 && batch <= g->maxB && steps > 0 && steps <= g->maxT, "tn_gnmt_encode: batch/steps exceed the handle");
  TN_ON_DEVICE(g->ctx->device);
  hipStream_t s = g->ctx->stream;
  const int H = g->H;
  TN_HIP_CHECK(hipMemcpyAsync(g->vl, valid_len, sizeof(int32_t) * batch, hipMemcpyDeviceToDevice, s));
  // layer i reads the previous layer's sequence and writes the other buffer; the last one writes mem.  Outputs past
  // valid_len are zero after every layer (tn_birnn_forward), so the closing SequenceMask (gnmt.py:159-161) is already applied.
  int rc = TN_OK;
  const float *in = src;
  for (int i = 0; i < g->NL; ++i) {
    float *outp = i == g->NL - 1 ? g->mem : (i & 1) ? g->seqB : g->seq0;
    rc = tn_birnn_forward(g->enc[i], in, batch, steps, g->vl, outp, g->hl[i], g->cl[i]);   // bi: hl / cl = [fwd, bwd] final states
    if (rc) return rc;
    if (g->residual && i > g->NBI) {      // gnmt.py:155-157: outputs + inputs from the SECOND uni-directional layer on
      const long nel = (long)batch * steps * H;
      hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, s, outp, in, nel);
    }
    in = outp;
  }
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
  TN_ON_DEVICE(g->ctx->device);
  hipStream_t s = g->ctx->stream;
  const int B = g->B, T = g->T, H = g->H, E = g->E, V = g->V, beam = g->beam, R = B * beam, L = g->maxL;
  const int K0 = E + 2 * H, K1 = 3 * H;
  const bool lstm = g->G == 4;
  const int nbm = beam <= 4 ? 4 : beam == 5 ? 5 : beam <= 8 ? 8 : 16;   // beam 5: the reference's flag default
  const int asplit = att_split(T, H, kStepLdsMax);
  const int Vp = (V + 3) & ~3;
  auto beam_lds_of = [&](int sp) { return ((size_t)2 * ((nbm + 3) & ~3) * H + (size_t)(1 + kBeamThreads / step_groups(Vp / 4, sp)) * beam * Vp + 16) * sizeof(float); };
  int bsplit = 16;
  while (bsplit > 1 && beam_lds_of(bsplit) > kStepLdsMax) bsplit >>= 1;
  const size_t beam_lds = beam_lds_of(bsplit);
  TN_REQUIRE(beam_lds <= kStepLdsMax && asplit > 0,
             "tn_gnmt_beam_search: beam * (2*hidden + 2*vocab) or 3 * max(hidden, source length) floats exceed the step kernels' 152 KiB of LDS");
  const size_t att_lds = att_lds_bytes(T, H, asplit);
  if (int rc = allow_lds(dec_attention_kernel<1>, att_lds)) return rc;
  if (int rc = nbm == 4 ? allow_lds(dec_beam_kernel<4>, beam_lds) : nbm == 5 ? allow_lds(dec_beam_kernel<5>, beam_lds)
               : nbm == 8 ? allow_lds(dec_beam_kernel<8>, beam_lds) : allow_lds(dec_beam_kernel<16>, beam_lds)) return rc;
  TN_HIP_CHECK(hipMemsetAsync(g->samples[0], 0xff, sizeof(int32_t) * (size_t)R * L, s));
  // decoder layer i starts from encoder layer i's state, the BACKWARD direction's for a bidirectional layer (gnmt.py:146-150,224-252)
  const int NL = g->NL, nmid = NL - 2;
  auto hinit = [&](int i) { return (const float *)(g->hl[i] + (i < g->NBI ? (size_t)B * H : 0)); };
  auto cinit = [&](int i) { return (const float *)(g->cl[i] + (i < g->NBI ? (size_t)B * H : 0)); };
  const bool fused0 = NL == 2;
  float *sx1 = fused0 ? g->sx1p : g->sx1;            // the last cell's step input, pitch ld1
  const int ld1 = fused0 ? -(g->maxB * beam) : K1;      // two layers: k-group-major, the form the gate GEMM reads (xidx)
  hipLaunchKernelGGL(dec_init_kernel, dim3(R), dim3(256), 0, s, (const float *)g->emb, bos, hinit(0), hinit(NL - 1),
                     lstm ? cinit(0) : (const float *)nullptr, cinit(NL - 1), g->sx0, sx1, g->c0cur, g->c1cur, beam, H, E, ld1);
  const int nbm_ = (R * H + 255) / 256;
  for (int j = 0; j < nmid; ++j)
    hipLaunchKernelGGL(dec_mid_state_kernel, dim3(nbm_), dim3(256), 0, s, (const int32_t *)nullptr, hinit(j + 1),
                       lstm ? cinit(j + 1) : (const float *)nullptr, g->mid[j].sx, g->mid[j].ccur, 1, beam, R, H);
  float *x_after0 = nmid ? g->mid[0].sx : sx1;          // input of the cell behind the attention
  const int ld_after0 = nmid ? K1 : ld1;
  hipLaunchKernelGGL(beam_init_kernel, dim3((R + 255) / 256), dim3(256), 0, s, g->scores, g->alive, g->vlen, g->tok, g->samples[0], L, B, beam, bos);
  // Two decoder layers (the reference's default): 3 launches per step.  The first cell's pre-activations g0 are linear in
  // x0 = [embed(word), ctx, h0]: step i + 1's attention kernel puts them together from ew[word] and the [h0, ctx] share p0 that
  // extra workgroups of step i's beam launch computed beside the clips (dec_beam_kernel).  More layers: the cells between
  // attention and the last one have their own operands, the first cell keeps its own GEMM (2 + 2 per layer launches).
  LatGemmArgs gm{};
  int gemm_wgs = 0;
  size_t beam_lds_launch = beam_lds;
  if (fused0) {
    if (!g->ew_ready) {      // embed . W0[:, 0:E]^T: (V, E) x (4H rows of K0, the first E columns) -> (V, 4H)
      if (int rc = launch_linear_f32(g->emb, E, g->w0c, K0, nullptr, g->ew, 4 * H, V, 4 * H, E, 0, s)) return rc;
      g->ew_ready = true;
    }
    gm = LatGemmArgs{sx1, g->w1x, g->b1x, g->p0, ld1, g->K1p, 4 * H, R, 4 * H, 2 * H};
    gemm_wgs = (((R + 15) / 16) * (4 * H / 16) + 3) / 4;
    if (beam_lds_launch < 4 * 4 * 64 * 4 * sizeof(float)) beam_lds_launch = 4 * 4 * 64 * 4 * sizeof(float);   // the tiles' partial sums
  }
  float *h0buf[2] = {g->h0n, fused0 ? g->h0n2 : g->h0n}, *c0buf[2] = {g->c0n, fused0 ? g->c0n2 : g->c0n};
  int steps_done = 0, all_dead = 0;
  // the loop is bound by launch-to-launch dependency latency, not by arithmetic
  for (int i = 0; i < max_length; ++i) {
    const int step = i + 1, cur = step & 1;
    if ((i & 15) == 0) TN_HIP_CHECK(hipMemsetAsync(g->flag, 0, sizeof(int32_t), s));
    int rc = TN_OK;
    if (!fused0 || i == 0) {
      rc = launch_linear_f32_lat(g->sx0, K0, g->w0c, K0, g->b0c, g->g0, 4 * H, R, 4 * H, K0, s);
      if (rc) return rc;
      hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(R), dim3(kBeamThreads), att_lds, s, (const float *)g->g0,
                         (const float *)(g->sx0 + E + H), K0, (const float *)g->c0cur, lstm ? 1 : 0, h0buf[cur], c0buf[cur], x_after0, ld_after0,
                         (const float *)g->keyprojT, (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, beam, 1, T, H,
                         (float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, (const float *)nullptr, (const float *)nullptr, asplit);
    } else {
      hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(R), dim3(kBeamThreads), att_lds, s, (const float *)nullptr,
                         (const float *)h0buf[cur ^ 1], H, (const float *)c0buf[cur ^ 1], lstm ? 1 : 0, h0buf[cur], c0buf[cur], x_after0, ld_after0,
                         (const float *)g->keyprojT, (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, beam, 1, T, H,
                         (float *)nullptr, (const int32_t *)g->tok, (const int32_t *)g->parent, (const float *)g->ew, (const float *)g->p0, asplit);
    }
    for (int j = 0; j < nmid; ++j) {
      GnmtMid &m = g->mid[j];
      rc = launch_linear_f32_lat(m.sx, K1, m.w, K1, m.b, m.g, 4 * H, R, 4 * H, K1, s);
      if (rc) return rc;
      hipLaunchKernelGGL(dec_mid_cell_kernel, dim3(nbm_), dim3(256), 0, s, (const float *)m.g, (const float *)m.sx, lstm ? 1 : 0,
                         (const float *)m.ccur, m.hn, m.cn, j + 1 < nmid ? g->mid[j + 1].sx : g->sx1, g->residual ? 1 : 0, R, H);
    }
    rc = fused0 ? launch_linear_f32_lat_wk4(sx1, ld1, g->w1k4, g->b1c, g->g1, 4 * H, R, 4 * H, K1, s)
                : launch_linear_f32_lat(g->sx1, K1, g->w1c, K1, g->b1c, g->g1, 4 * H, R, 4 * H, K1, s);
    if (rc) return rc;
    // BeamSearchScorer [EXT gluonnlp]: length penalty ((K + length) / (K + 1)) ^ alpha of this step and of the previous one
    const float lp = powf(K + (float)step, alpha) / powf(K + 1.f, alpha);
    const float prev_lp = step == 1 ? 1.f : powf(K + (float)(step - 1), alpha) / powf(K + 1.f, alpha);
#define TN_BEAM_LAUNCH(NBM)                                                                                              \
  hipLaunchKernelGGL(dec_beam_kernel<NBM>, dim3(B + gemm_wgs), dim3(kBeamThreads), beam_lds_launch, s,                   \
                     (const float *)g->g1, 4 * H, sx1, ld1, lstm ? 1 : 0, g->c1cur, (const float *)g->wpT,                \
                     (const float *)g->bpp, (const float *)g->h0n, (const float *)g->ctxn, (const float *)g->c0n, g->sx0, \
                     g->c0cur, (const float *)g->emb, H, E, V, beam, step, lp, prev_lp, eos, g->scores, g->alive,         \
                     g->vlen, g->bp_par, g->bp_word, R, g->flag, g->hstate, g->parent,                                    \
                     fused0 ? g->tok : (int32_t *)nullptr, B, gm, bsplit)
    if (nbm == 4) TN_BEAM_LAUNCH(4);
    else if (nbm == 5) TN_BEAM_LAUNCH(5);
    else if (nbm == 8) TN_BEAM_LAUNCH(8);
    else TN_BEAM_LAUNCH(16);
#undef TN_BEAM_LAUNCH
    for (int j = 0; j < nmid; ++j)      // the middle layers' states follow their rows' parent beams
      hipLaunchKernelGGL(dec_mid_state_kernel, dim3(nbm_), dim3(256), 0, s, (const int32_t *)g->parent, (const float *)g->mid[j].hn,
                         lstm ? (const float *)g->mid[j].cn : (const float *)nullptr, g->mid[j].sx, g->mid[j].ccur, 0, beam, R, H);
    TN_HIP_CHECK(hipGetLastError());
    steps_done = step;
    if ((i & 15) == 15 || i == max_length - 1) {   // look at the device flag every 16 steps
      int32_t f = 1;
      TN_HIP_CHECK(hipMemcpyAsync(&f, g->flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      TN_HIP_CHECK(hipStreamSynchronize(s));
      if (!f) { all_dead = 1; break; }
    }
  }
  // the token prefixes, from the back-pointers of the steps taken
  hipLaunchKernelGGL(beam_samples_kernel, dim3(B), dim3(256), 0, s, (const int32_t *)g->bp_par, (const int32_t *)g->bp_word, g->samples[0], L,
                     steps_done, R, beam, bos);
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
  TN_ON_DEVICE(g->ctx->device);
  hipStream_t s = g->ctx->stream;
  const int B = g->B, T = g->T, H = g->H, E = g->E, V = g->V, R = B, K0 = E + 2 * H, K1 = 3 * H;
  const bool lstm = g->G == 4;
  const int nb = (R * H + 255) / 256;
  const int asplit = att_split(T, H, kStepLdsMax);
  TN_REQUIRE(asplit > 0, "tn_gnmt_decode_seq: 3 * max(hidden, source length) floats exceed the step kernel's 152 KiB of LDS");
  const size_t att_lds = att_lds_bytes(T, H, asplit);
  if (int rc = allow_lds(dec_attention_kernel<1>, att_lds)) return rc;
  const int NL = g->NL, nmid = NL - 2;
  auto hinit = [&](int i) { return (const float *)(g->hl[i] + (i < g->NBI ? (size_t)B * H : 0)); };
  auto cinit = [&](int i) { return (const float *)(g->cl[i] + (i < g->NBI ? (size_t)B * H : 0)); };
  float *x_after0 = nmid ? g->mid[0].sx : g->sx1;
  float *proj_in = g->residual ? g->hstate : g->h1n;     // use_residual: the projection sees h + the last cell's input
  for (int i = 0; i < steps; ++i) {
    // decoder layer j starts from encoder layer j's state, the BACKWARD direction's for a bidirectional layer (gnmt.py:146-150,224-252)
    const float *h0p = i ? g->h0n : hinit(0), *h1p = i ? g->h1n : hinit(NL - 1);
    const float *c0p = !lstm ? nullptr : i ? g->c0n : cinit(0), *c1p = !lstm ? nullptr : i ? g->c1n : cinit(NL - 1);
    hipLaunchKernelGGL(dec_tf_prep_kernel, dim3(R), dim3(256), 0, s, (const float *)g->emb, tgt, ld, i,
                       i ? (const float *)g->ctxn : (const float *)nullptr, h0p, h1p, c0p, c1p, g->sx0, g->sx1, g->c0cur, g->c1cur, H, E);
    for (int j = 0; j < nmid; ++j)
      hipLaunchKernelGGL(dec_mid_state_kernel, dim3(nb), dim3(256), 0, s, (const int32_t *)nullptr, i ? (const float *)g->mid[j].hn : hinit(j + 1),
                         !lstm ? (const float *)nullptr : i ? (const float *)g->mid[j].cn : cinit(j + 1), g->mid[j].sx, g->mid[j].ccur, 0, 1, R, H);
    int rc = launch_linear_f32_lat(g->sx0, K0, g->w0c, K0, g->b0c, g->g0, 4 * H, R, 4 * H, K0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(R), dim3(kBeamThreads), att_lds, s, (const float *)g->g0,
                       (const float *)(g->sx0 + E + H), K0, (const float *)g->c0cur, lstm ? 1 : 0, g->h0n, g->c0n, x_after0, K1,
                       (const float *)g->keyprojT, (const float *)g->mem, (const int32_t *)g->vl, g->ctxn, 1, 1, T, H,
                       (float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, (const float *)nullptr, (const float *)nullptr, asplit);
    for (int j = 0; j < nmid; ++j) {
      GnmtMid &m = g->mid[j];
      rc = launch_linear_f32_lat(m.sx, K1, m.w, K1, m.b, m.g, 4 * H, R, 4 * H, K1, s);
      if (rc) return rc;
      hipLaunchKernelGGL(dec_mid_cell_kernel, dim3(nb), dim3(256), 0, s, (const float *)m.g, (const float *)m.sx, lstm ? 1 : 0,
                         (const float *)m.ccur, m.hn, m.cn, j + 1 < nmid ? g->mid[j + 1].sx : g->sx1, g->residual ? 1 : 0, R, H);
    }
    rc = g->w1k4 ? launch_linear_f32_lat_wk4(g->sx1, K1, g->w1k4, g->b1c, g->g1, 4 * H, R, 4 * H, K1, s)      // (two layers: the k-group-major copy)
                 : launch_linear_f32_lat(g->sx1, K1, g->w1c, K1, g->b1c, g->g1, 4 * H, R, 4 * H, K1, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_tf_cell1_kernel, dim3(nb), dim3(256), 0, s, (const float *)g->g1, (const float *)g->sx1, lstm ? 1 : 0,
                       (const float *)g->c1cur, g->h1n, g->c1n, R, H, g->residual ? g->hstate : (float *)nullptr);
    rc = launch_linear_f32_lat(proj_in, H, g->wp, H, g->bp, logits + (size_t)i * V, steps * V, R, V, H, s);
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
  TN_ON_DEVICE(ctx->device);
  hipLaunchKernelGGL(masked_ce_kernel, dim3(batch), dim3(256), 0, ctx->stream, logits, labels, ld_labels, valid_len, loss, steps, vocab);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// ---- captioner training handle ------------------------------------------------------------------------------------------
// Round 4: any num_layers >= 2 with num_bi_layers < num_layers and use_residual, as the reference passes them into the model it
// trains (train_gnmt.py:58-61,223-227; gnmt.py:71-111,136-160,369-404).  Every encoder / decoder layer is a record of its
// parameter offsets, the forward state kept for the backward pass and its backward workspace; the step below walks the records.
namespace {

// a decoder cell behind the first one: gates on the stacked pre-activations g (R,4H) of x = [layer input, attention, h_prev];
// the new state to hn (+ cn); the layer's OUTPUT = dropout(h) (+ the layer's input with use_residual, gnmt.py:393-396) to
// out (row stride ldo: the next layer's step input, or the rows the projection reads); the attention vector is copied along
__global__ void trn_cell_fwd_kernel(const float *__restrict__ g, const float *__restrict__ x, int lstm, const float *__restrict__ cprev,
                                    float *__restrict__ hn, float *__restrict__ cn, const float *__restrict__ mask, int residual,
                                    float *__restrict__ out, int ldo, float *__restrict__ att_dst, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  const float *gr = g + r * 4 * H;
  float h;
  if (lstm) {
    const float ig = sigm(gr[u]), fg = sigm(gr[H + u]), gg = tanhf(gr[2 * H + u]), og = sigm(gr[3 * H + u]);
    const float c2 = fg * cprev[id] + ig * gg;
    cn[id] = c2;
    h = og * tanhf(c2);
  } else {
    const float rg = sigm(gr[u]), zg = sigm(gr[H + u]);
    const float ng = tanhf(gr[2 * H + u] + rg * gr[3 * H + u]);
    h = (1.f - zg) * ng + zg * x[r * 3 * H + 2 * H + u];
  }
  hn[id] = h;
  out[r * ldo + u] = (mask ? h * mask[id] : h) + (residual ? x[r * 3 * H + u] : 0.f);
  if (att_dst) att_dst[r * ldo + u] = x[r * 3 * H + H + u];
}

// out (R,H) = a[r * la + u] (* m[r * H + u]) (+ b[r * lb + u])
__global__ void trn_combine_kernel(const float *__restrict__ a, int la, const float *__restrict__ m, const float *__restrict__ b, int lb,
                                   float *__restrict__ out, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  float v = a[r * la + u];
  if (m) v *= m[id];
  if (b) v += b[r * lb + u];
  out[id] = v;
}
// dst[r * ld + u] = src[r * ls + u]
__global__ void trn_copy_kernel(const float *__restrict__ src, int ls, float *__restrict__ dst, int ld, int R, int H) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)R * H) return;
  const long r = id / H;
  const int u = (int)(id - r * H);
  dst[r * ld + u] = src[r * ls + u];
}
// y = x * m + r (element-wise over n; m / r optional)
__global__ void trn_mul_add_kernel(const float *__restrict__ x, const float *__restrict__ m, const float *__restrict__ r,
                                   float *__restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  if (m) v *= m[i];
  if (r) v += r[i];
  y[i] = v;
}

struct TrnEnc {          // encoder layer i: bidirectional (i < num_bi_layers: in -> 2H) or uni-directional (in -> H)
  int in, D, out;        // input width, directions, output width D * H
  bool res;              // use_residual && i > num_bi_layers (gnmt.py:155-157)
  long o_wi, o_bi, o_wh, o_bh;     // the directions adjacent per tensor kind: one GEMM / one recurrent launch serves both
  float *whT, *wiT;      // per direction (H, GH); (in, D GH) [layers behind the first]
  float *gi, *seq, *sav, *hl, *cl, *M, *xn;       // xn: what the next layer (or the attention) reads = dropout(seq) (+ input)
  float *dxn, *dseq, *dgi, *dgh, *hp, *dhl, *dcl; // dxn: gradient w.r.t. xn
};
struct TrnDec {          // decoder cell j: step input x = [embedding | layer input, attention, h_prev], K = in + H columns
  int in, K;
  long o_wi, o_bi, o_wh, o_bh;
  float *wc, *bc, *wcT;  // stacked (4H, K) step matrix, its bias, its transpose
  float *X, *G, *C, *Hs, *M;       // per step (L, B, .): inputs, stacked pre-activations, cell states (LSTM), states h (j >= 1), dropout masks (j >= 1)
  float *dG, *dX, *dhz, *dcz, *dW, *db;
};

}  // namespace

struct tn_gnmt_trainer {
  tn_ctx *ctx;
  DevBuf pool;
  int F, H, E, V, maxB, maxT, maxL, NL, NBI;
  bool residual;
  int G;                          // gates per cell: 3 GRU, 4 LSTM
  std::string prefix;
  long n, step;
  long o_wk, o_wp, o_bp, o_emb;
  float *w, *g, *am, *av;         // flat parameter / gradient / Adam-moment buffers
  std::vector<TrnEnc> enc;
  std::vector<TrnDec> dec;
  float *wpT, *wkT;
  float *DS, *DCTX;          // per decoder step: the attention scores' gradient (L,B,T) and the context's (L,B,H)
  float *keyproj, *keyprojT, *AW, *h0tmp, *ctxtmp, *logits, *lossrows, *Out, *dlog, *dOut, *dq, *dkp, *tmpA, *tmpB, *tmpM, *tmpAtt;
  int32_t *vl, *tvl;
  float drop_p;
  unsigned long long drop_seed, drop_count;
};

static int trainer_refresh(tn_gnmt_trainer *t) {
  hipStream_t s = t->ctx->stream;
  const int H = t->H, V = t->V, GH = t->G * H, lstm = t->G == 4;
  int rc;
#define TN_TRY(e) do { rc = (e); if (rc) return rc; } while (0)
  for (size_t i = 0; i < t->enc.size(); ++i) {
    TrnEnc &e = t->enc[i];
    for (int d = 0; d < e.D; ++d) TN_TRY(launch_transpose_f32(t->w + e.o_wh + (long)d * GH * H, GH, H, e.whT + (long)d * H * GH, s));
    if (i > 0) TN_TRY(launch_transpose_f32(t->w + e.o_wi, e.D * GH, e.in, e.wiT, s));
  }
  for (TrnDec &d : t->dec) {
    hipLaunchKernelGGL(trn_stack_kernel, dim3(4 * H), dim3(256), 0, s, (const float *)(t->w + d.o_wi), (const float *)(t->w + d.o_wh),
                       (const float *)(t->w + d.o_bi), (const float *)(t->w + d.o_bh), d.in, H, lstm, d.wc, d.bc);
    TN_TRY(launch_transpose_f32(d.wc, 4 * H, d.K, d.wcT, s));
  }
  TN_TRY(launch_transpose_f32(t->w + t->o_wp, V, H, t->wpT, s));
  TN_TRY(launch_transpose_f32(t->w + t->o_wk, H, H, t->wkT, s));
#undef TN_TRY
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

extern "C" int tn_gnmt_trainer_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, tn_rnn_kind cell_kind,
                                         int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers, int flags,
                                         int max_batch, int max_src_len, int max_tgt_len, tn_gnmt_trainer **out) {
  TN_REQUIRE(ctx && params && prefix_c && out, "tn_gnmt_trainer_create: null argument");
  TN_REQUIRE(cell_kind == TN_RNN_GRU || cell_kind == TN_RNN_LSTM, "tn_gnmt_trainer_create: cell_type must be 'gru' or 'lstm'");
  TN_REQUIRE((flags & ~TN_GNMT_USE_RESIDUAL) == 0, "tn_gnmt_trainer_create_ex: unknown flag");
  // the same shapes tn_gnmt_create_ex serves (gnmt.py:78-80; a memory of 2H columns does not fit the attention's key width)
  TN_REQUIRE(num_layers >= 2 && num_layers <= 8 && num_bi_layers >= 0 && num_bi_layers < num_layers,
             "tn_gnmt_trainer_create: need 2 <= num_layers <= 8 and 0 <= num_bi_layers < num_layers");
  const int G_ = cell_kind == TN_RNN_GRU ? 3 : 4;
  TN_REQUIRE(input_size > 0 && hidden > 0 && hidden % 4 == 0 && G_ * hidden <= 1024 && embed > 0 && vocab > 1 && max_batch > 0 &&
                 max_src_len > 0 && max_tgt_len > 1, "tn_gnmt_trainer_create: bad shape (gates*hidden <= 1024, hidden % 4 == 0)");
  TN_ON_DEVICE(ctx->device);
  const std::string pre(prefix_c);
  std::map<std::string, const tn_param *> pm;
  for (int i = 0; i < n_params; ++i) pm[params[i].name] = &params[i];
  tn_gnmt_trainer *t = new tn_gnmt_trainer();
  t->ctx = ctx; t->F = input_size; t->H = hidden; t->E = embed; t->V = vocab; t->maxB = max_batch; t->maxT = max_src_len;
  t->maxL = max_tgt_len - 1; t->prefix = pre; t->step = 0; t->G = G_; t->NL = num_layers; t->NBI = num_bi_layers;
  t->residual = (flags & TN_GNMT_USE_RESIDUAL) != 0;
  const long H = hidden, E = embed, V = vocab, GH = (long)G_ * H;
  long o = 0;
  auto take = [&](long cnt) { const long r = o; o += cnt; return r; };
  t->enc.resize(num_layers); t->dec.resize(num_layers);
  int fin = input_size;
  for (int i = 0; i < num_layers; ++i) {
    TrnEnc &e = t->enc[i];
    e.in = fin; e.D = i < num_bi_layers ? 2 : 1; e.out = e.D * hidden; e.res = t->residual && i > num_bi_layers;
    e.o_wi = take(e.D * GH * e.in); e.o_bi = take(e.D * GH); e.o_wh = take(e.D * GH * H); e.o_bh = take(e.D * GH);
    fin = e.out;
  }
  for (int j = 0; j < num_layers; ++j) {
    TrnDec &d = t->dec[j];
    d.in = j == 0 ? embed + hidden : 2 * hidden; d.K = d.in + hidden;
    d.o_wi = take(GH * d.in); d.o_bi = take(GH); d.o_wh = take(GH * H); d.o_bh = take(GH);
  }
  t->o_wk = take(H * H); t->o_wp = take(V * H); t->o_bp = take(V); t->o_emb = take(V * E);
  t->n = o;
  std::vector<float> w(t->n);
  auto fail = [&](int code) { t->pool.release(); delete t; return code; };
  bool ok = true;
  auto put = [&](const std::string &name, long off, long cnt) {
    auto it = pm.find(pre + name);
    if (it == pm.end()) { tn_set_error("missing parameter: " + pre + name); ok = false; return; }
    if (it->second->numel != cnt) { tn_set_error("parameter " + pre + name + " has the wrong size"); ok = false; return; }
    memcpy(&w[off], it->second->data_host, sizeof(float) * cnt);
  };
  for (int i = 0; i < num_layers && ok; ++i) {
    const TrnEnc &e = t->enc[i];
    for (int d = 0; d < e.D && ok; ++d) {
      const std::string c = "enc_rnn" + std::to_string(i) + (e.D == 2 ? (d ? "_r_" : "_l_") : "_");
      put(c + "i2h_weight", e.o_wi + d * GH * e.in, GH * e.in); put(c + "i2h_bias", e.o_bi + d * GH, GH);
      put(c + "h2h_weight", e.o_wh + d * GH * H, GH * H); put(c + "h2h_bias", e.o_bh + d * GH, GH);
    }
  }
  for (int j = 0; j < num_layers && ok; ++j) {
    const TrnDec &d = t->dec[j];
    const std::string c = "dec_rnn" + std::to_string(j) + "_";
    put(c + "i2h_weight", d.o_wi, GH * d.in); put(c + "i2h_bias", d.o_bi, GH);
    put(c + "h2h_weight", d.o_wh, GH * H); put(c + "h2h_bias", d.o_bh, GH);
  }
  if (ok) {
    put("dec_attention_key_weight", t->o_wk, H * H); put("tgt_proj_weight", t->o_wp, V * H); put("tgt_proj_bias", t->o_bp, V);
    put("tgt_embed_weight", t->o_emb, V * E);
  }
  if (!ok) return fail(TN_ERR_MISSING);
  t->w = t->pool.upload(w.data(), w.size());
  auto fl = [&](size_t n) { return t->pool.alloc<float>(n); };
  t->g = fl(t->n); t->am = fl(t->n); t->av = fl(t->n);
  const size_t B = max_batch, T = max_src_len, L = t->maxL, BT = B * T, LB = L * B;
  for (int i = 0; i < num_layers; ++i) {
    TrnEnc &e = t->enc[i];
    const size_t D = e.D, DG = D * GH, O = e.out;
    e.whT = fl(D * H * GH); e.wiT = i ? fl((size_t)e.in * DG) : nullptr;
    e.gi = fl(BT * DG); e.seq = fl(BT * O); e.sav = fl(D * BT * (G_ + 1) * H); e.hl = fl(D * B * H); e.cl = fl(D * B * H);
    e.M = fl(BT * O); e.xn = fl(BT * O);
    e.dxn = fl(BT * O); e.dseq = fl(BT * O); e.dgi = fl(BT * DG); e.dgh = fl(BT * DG); e.hp = fl(D * BT * H);
    e.dhl = fl(D * B * H); e.dcl = fl(D * B * H);
  }
  for (int j = 0; j < num_layers; ++j) {
    TrnDec &d = t->dec[j];
    const size_t K = d.K;
    d.wc = fl(4 * H * K); d.bc = fl(4 * H); d.wcT = fl(4 * H * K);
    d.X = fl(LB * K); d.G = fl(LB * 4 * H); d.C = fl(LB * H); d.Hs = fl(LB * H); d.M = fl(LB * H);
    d.dG = fl(LB * 4 * H); d.dX = fl(LB * K); d.dhz = fl(B * H); d.dcz = fl(B * H); d.dW = fl(4 * H * K); d.db = fl(4 * H);
  }
  t->wpT = fl(V * H); t->wkT = fl(H * H);
  t->keyproj = fl(BT * H); t->keyprojT = fl(B * ((T + 3) & ~(size_t)3) * H); t->AW = fl(LB * T); t->DS = fl(LB * T); t->DCTX = fl(LB * H); t->h0tmp = fl(B * H); t->ctxtmp = fl(B * H);
  t->logits = fl(LB * V); t->lossrows = fl(LB); t->Out = fl(LB * H); t->dlog = fl(LB * V); t->dOut = fl(LB * H); t->dq = fl(B * H);
  t->dkp = fl(BT * H); t->tmpA = fl(B * H); t->tmpB = fl(B * H); t->tmpM = fl(B * H); t->tmpAtt = fl(B * H);
  t->vl = t->pool.alloc<int32_t>(B); t->tvl = t->pool.alloc<int32_t>(B);
  t->drop_p = 0.f; t->drop_seed = 0; t->drop_count = 0;
  if (t->pool.failed) { tn_set_error("device allocation failed"); return fail(TN_ERR_NOMEM); }
  TN_HIP_CHECK(hipMemsetAsync(t->g, 0, sizeof(float) * t->n, ctx->stream));
  TN_HIP_CHECK(hipMemsetAsync(t->am, 0, sizeof(float) * t->n, ctx->stream));
  TN_HIP_CHECK(hipMemsetAsync(t->av, 0, sizeof(float) * t->n, ctx->stream));
  const int rc = trainer_refresh(t);
  if (rc) return fail(rc);
  *out = t;
  return TN_OK;
}

// the reference's flag defaults (train_gnmt.py:58-61): two layers, the first bidirectional, no residual connections
extern "C" int tn_gnmt_trainer_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix_c, tn_rnn_kind cell_kind,
                                      int input_size, int hidden, int embed, int vocab, int max_batch, int max_src_len,
                                      int max_tgt_len, tn_gnmt_trainer **out) {
  return tn_gnmt_trainer_create_ex(ctx, params, n_params, prefix_c, cell_kind, input_size, hidden, embed, vocab, 2, 1, 0, max_batch,
                                   max_src_len, max_tgt_len, out);
}

__global__ void trn_dec_len_kernel(const int32_t *__restrict__ tgt_vl, int32_t *__restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = tgt_vl[b] - 1;          // train_gnmt.py:331-332: the model and the loss see tgt_valid_length - 1
}

// One training step's forward + backward.  src (B,T,F) fp32, src_valid_len (B), tgt (B, ld) token ids with tgt_len >= 2
// columns used (inputs tgt[:, :-1], labels tgt[:, 1:]), tgt_valid_len (B) as the data loader gives it (BOS and EOS
// counted); all DEVICE buffers.  loss (1 float, device) = summed NLL of the valid target tokens / their number
// (= the scalar of train_gnmt.py:332-333); logits_out (B, tgt_len-1, V) optional.  Gradients of that loss land in the
// flat gradient buffer (tn_gnmt_trainer_buffers).
extern "C" int tn_gnmt_trainer_forward_backward(tn_gnmt_trainer *t, const float *src, const int32_t *src_valid_len,
                                                const int32_t *tgt, int ld, const int32_t *tgt_valid_len, int batch, int steps,
                                                int tgt_len, float *loss, float *logits_out) {
  TN_REQUIRE(t && src && src_valid_len && tgt && tgt_valid_len && loss, "tn_gnmt_trainer_forward_backward: null argument");
  TN_REQUIRE(batch > 0 && batch <= t->maxB && steps > 0 && steps <= t->maxT && tgt_len >= 2 && tgt_len - 1 <= t->maxL && ld >= tgt_len,
             "tn_gnmt_trainer_forward_backward: batch / source steps / target length exceed the handle");
  TN_ON_DEVICE(t->ctx->device);
  hipStream_t s = t->ctx->stream;
  const int B = batch, T = steps, L = tgt_len - 1, H = t->H, E = t->E, V = t->V, G = t->G, GH = G * H, NL = t->NL;
  const int K0 = E + 2 * H, K1 = 3 * H;
  const int BT = B * T, LB = L * B;
  const bool lstm = G == 4;
  const int asplit = att_split(T, H, kStepLdsMax);
  const size_t attb_lds = (size_t)(17 * H + T + 16) * sizeof(float);
  TN_REQUIRE(asplit > 0 && attb_lds <= kStepLdsMax, "tn_gnmt_trainer: 3 * max(hidden, source length) floats exceed 152 KiB of LDS");
  const size_t att_lds = att_lds_bytes(T, H, asplit);
  if (int rc = allow_lds(dec_attention_kernel<1>, att_lds)) return rc;
  if (int rc = allow_lds(trn_att_bwd_kernel, attb_lds)) return rc;
  float *w = t->w, *g = t->g;
  int rc;
#define TN_TRY(e) do { rc = (e); if (rc) return rc; } while (0)
  const int nbH = (B * H + 255) / 256;
  const float *nul = nullptr;
  auto blocks = [](long n) { return dim3((unsigned)((n + 255) / 256)); };
  // ---------------- forward: encoder (gnmt.py:136-160) ----------------
  TN_HIP_CHECK(hipMemcpyAsync(t->vl, src_valid_len, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(trn_dec_len_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tgt_valid_len, t->tvl, B);
  const bool drop = t->drop_p > 0.f;
  const unsigned long long key = drop ? t->drop_seed * 0x2545F4914F6CDD1Dull + (++t->drop_count) * 0xD1342543DE82EF95ull : 0ull;
  std::vector<const float *> xin(NL + 1);     // layer i reads xin[i]; xin[NL] = the memory
  xin[0] = src;
  for (int i = 0; i < NL; ++i) {
    TrnEnc &e = t->enc[i];
    const int DG = e.D * GH;
    TN_TRY(launch_linear_f32(xin[i], e.in, w + e.o_wi, e.in, w + e.o_bi, e.gi, DG, BT, DG, e.in, 0, s));
    TN_HIP_CHECK(hipMemsetAsync(e.seq, 0, sizeof(float) * (size_t)BT * e.out, s));
    TN_TRY(launch_rnn_recurrent(G, e.gi, DG, e.whT, w + e.o_bh, t->vl, e.seq, e.out, e.hl, lstm ? e.cl : nullptr, B, T, H, e.D, s, e.sav));
    // dropout on the layer's output (the states are not dropped), then the residual connection (gnmt.py:152-157)
    const long no = (long)BT * e.out;
    if (drop) hipLaunchKernelGGL(trn_dropout_mask_kernel, blocks(no), dim3(256), 0, s, e.M, no, t->drop_p, key ^ (0x1111ull * (i + 1)));
    if (drop || e.res) {
      hipLaunchKernelGGL(trn_mul_add_kernel, blocks(no), dim3(256), 0, s, (const float *)e.seq, drop ? (const float *)e.M : nul,
                         e.res ? xin[i] : nul, e.xn, no);
      xin[i + 1] = e.xn;
    } else {
      xin[i + 1] = e.seq;
    }
  }
  const float *mem = xin[NL];
  TN_TRY(launch_linear_f32(mem, H, w + t->o_wk, H, nullptr, t->keyproj, H, BT, H, H, 0, s));
  hipLaunchKernelGGL(transpose_bth_kernel, dim3((H + 31) / 32, (T + 31) / 32, B), dim3(256), 0, s, (const float *)t->keyproj, t->keyprojT, T, H);
  // the decoder layer j starts from encoder layer j's final state: the BACKWARD direction's for a bidirectional layer (gnmt.py:146-150)
  auto h_init = [&](int j) { return (const float *)(t->enc[j].hl + (size_t)(t->enc[j].D - 1) * B * H); };
  auto c_init = [&](int j) { return (const float *)(t->enc[j].cl + (size_t)(t->enc[j].D - 1) * B * H); };
  // ---------------- forward: decoder steps (gnmt.py:369-404) ----------------
  if (drop)
    for (int j = 1; j < NL; ++j)
      hipLaunchKernelGGL(trn_dropout_mask_kernel, blocks((long)LB * H), dim3(256), 0, s, t->dec[j].M, (long)LB * H, t->drop_p, key ^ (0x3333ull * j));
  TrnDec &d0 = t->dec[0], &d1 = t->dec[1];
  for (int i = 0; i < L; ++i) {
    const size_t so = (size_t)i * B, sp = (size_t)(i - 1) * B;
    float *X0 = d0.X + so * K0, *X1 = d1.X + so * K1;
    const float *X1p = d1.X + sp * K1;
    hipLaunchKernelGGL(trn_prep_kernel, dim3(B), dim3(256), 0, s, (const float *)(w + t->o_emb), tgt, ld, i,
                       i ? X1p + H : nul, K1, i ? X1p : h_init(0), i ? K1 : H,
                       i ? (const float *)(d1.Hs + sp * H) : h_init(1), H, X0, X1, H, E);
    for (int j = 2; j < NL; ++j)
      hipLaunchKernelGGL(trn_copy_kernel, dim3(nbH), dim3(256), 0, s, i ? (const float *)(t->dec[j].Hs + sp * H) : h_init(j), H,
                         t->dec[j].X + so * K1 + 2 * H, K1, B, H);
    TN_TRY(launch_linear_f32_lat(X0, K0, d0.wc, K0, d0.bc, d0.G + so * 4 * H, 4 * H, B, 4 * H, K0, s));
    const float *c0p = !lstm ? nul : i ? (const float *)(d0.C + sp * H) : c_init(0);
    hipLaunchKernelGGL(dec_attention_kernel<1>, dim3(B), dim3(kBeamThreads), att_lds, s, (const float *)(d0.G + so * 4 * H), (const float *)(X0 + E + H), K0,
                       c0p, lstm ? 1 : 0, t->h0tmp, lstm ? d0.C + so * H : (float *)nullptr, X1, K1, (const float *)t->keyprojT, mem,
                       (const int32_t *)t->vl, t->ctxtmp, 1, 1, T, H, t->AW + so * T, (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, asplit);
    for (int j = 1; j < NL; ++j) {
      TrnDec &d = t->dec[j];
      const bool top = j == NL - 1;
      float *Xj = d.X + so * K1;
      TN_TRY(launch_linear_f32_lat(Xj, K1, d.wc, K1, d.bc, d.G + so * 4 * H, 4 * H, B, 4 * H, K1, s));
      const float *cp = !lstm ? nul : i ? (const float *)(d.C + sp * H) : c_init(j);
      float *outp = top ? t->Out + so * H : t->dec[j + 1].X + so * K1;
      hipLaunchKernelGGL(trn_cell_fwd_kernel, dim3(nbH), dim3(256), 0, s, (const float *)(d.G + so * 4 * H), (const float *)Xj, lstm ? 1 : 0, cp,
                         d.Hs + so * H, lstm ? d.C + so * H : (float *)nullptr, drop ? (const float *)(d.M + so * H) : nul, t->residual ? 1 : 0,
                         outp, top ? H : K1, top ? (float *)nullptr : outp + H, B, H);
    }
    TN_TRY(launch_linear_f32_lat(t->Out + so * H, H, w + t->o_wp, H, w + t->o_bp, t->logits + (size_t)i * V, L * V, B, V, H, s));
  }
  // ---------------- loss and its gradient ----------------
  hipLaunchKernelGGL(trn_ce_bwd_kernel, dim3(L, B), dim3(256), 0, s, (const float *)t->logits, tgt + 1, ld, (const int32_t *)t->tvl, B, L, V,
                     t->dlog, t->lossrows);
  TN_TRY(launch_colsum_f32(t->lossrows, 1, LB, 1, loss, s));
  if (logits_out) TN_HIP_CHECK(hipMemcpyAsync(logits_out, t->logits, sizeof(float) * (size_t)LB * V, hipMemcpyDeviceToDevice, s));
  // ---------------- backward: projection ----------------
  TN_TRY(launch_linear_f32(t->dlog, V, t->wpT, V, nullptr, t->dOut, H, LB, H, V, 0, s));
  TN_TRY(launch_gemm_tn_f32(t->dlog, V, t->Out, H, g + t->o_wp, H, V, H, LB, s));
  TN_TRY(launch_colsum_f32(t->dlog, V, LB, V, g + t->o_bp, s));
  float *dmem = t->enc[NL - 1].dxn;      // (written, with t->dkp, by trn_att_outer_kernel after the loop)
  // ---------------- backward: decoder steps, last to first ----------------
  for (int i = L - 1; i >= 0; --i) {
    const bool last = i == L - 1;
    const size_t so = (size_t)i * B, sp = (size_t)(i - 1) * B, sn = (size_t)(i + 1) * B;
    // d (output of layer j), walking down from the rows the projection read: pointer + row stride
    const float *dcur = t->dOut + so * H;
    int ldcur = H;
    float *pingpong[2] = {t->tmpA, t->tmpB};
    int pp = 0;
    for (int j = NL - 1; j >= 1; --j) {
      TrnDec &d = t->dec[j];
      const float *Xj = d.X + so * K1, *Gj = d.G + so * 4 * H;
      float *dGj = d.dG + so * 4 * H, *dXj = d.dX + so * K1;
      const float *dXjn = d.dX + sn * K1;
      // through the dropout mask to the cell's h
      const float *dh = dcur;
      int ldh = ldcur;
      if (drop) {
        hipLaunchKernelGGL(trn_combine_kernel, dim3(nbH), dim3(256), 0, s, dcur, ldcur, (const float *)(d.M + so * H), nul, 0, t->tmpM, B, H);
        dh = t->tmpM; ldh = H;
      }
      if (lstm)
        hipLaunchKernelGGL(trn_lstm_bwd_kernel, dim3(nbH), dim3(256), 0, s, Gj, i ? (const float *)(d.C + sp * H) : c_init(j), (const float *)(d.C + so * H),
                           dh, ldh, last ? nul : (const float *)d.dhz, H, last ? nul : dXjn + 2 * H, K1, nul, 0, last ? nul : (const float *)d.dcz,
                           dGj, d.dhz, d.dcz, B, H);
      else
        hipLaunchKernelGGL(trn_gru_bwd_kernel, dim3(nbH), dim3(256), 0, s, Gj, Xj + 2 * H, K1, dh, ldh, last ? nul : (const float *)d.dhz, H,
                           last ? nul : dXjn + 2 * H, K1, nul, 0, dGj, d.dhz, B, H);
      TN_TRY(launch_linear_f32_lat(dGj, 4 * H, d.wcT, 4 * H, nullptr, dXj, K1, B, K1, 4 * H, s));
      // d (the layer's input) = its column block of dX (+ the residual path)
      if (t->residual) {
        float *dst = pingpong[pp]; pp ^= 1;
        hipLaunchKernelGGL(trn_combine_kernel, dim3(nbH), dim3(256), 0, s, (const float *)dXj, K1, nul, dcur, ldcur, dst, B, H);
        dcur = dst; ldcur = H;
      } else {
        dcur = dXj; ldcur = K1;
      }
    }
    // attention: d ctx = the attention columns of every layer's dX at this step + the first cell's at the next step
    const float *datt = t->dec[1].dX + so * K1 + H;
    int ldatt = K1;
    if (NL > 2) {
      hipLaunchKernelGGL(trn_combine_kernel, dim3(nbH), dim3(256), 0, s, datt, K1, nul, (const float *)(t->dec[2].dX + so * K1 + H), K1, t->tmpAtt, B, H);
      for (int j = 3; j < NL; ++j)
        hipLaunchKernelGGL(trn_combine_kernel, dim3(nbH), dim3(256), 0, s, (const float *)t->tmpAtt, H, nul, (const float *)(t->dec[j].dX + so * K1 + H), K1, t->tmpAtt, B, H);
      datt = t->tmpAtt; ldatt = H;
    }
    const float *dX0n = d0.dX + sn * K0;
    hipLaunchKernelGGL(trn_att_bwd_kernel, dim3(B), dim3(kBeamThreads), attb_lds, s, (const float *)(t->AW + so * T), mem, (const float *)t->keyproj,
                       datt, ldatt, last ? nul : dX0n + E, K0, (const int32_t *)t->vl, t->DS + so * T, t->DCTX + so * H, t->dq, T, H);
    // first cell: d h0 = d (layer 1's input) + the query's gradient + the recurrent paths
    const float *X0 = d0.X + so * K0, *G0 = d0.G + so * 4 * H;
    float *dG0 = d0.dG + so * 4 * H, *dX0 = d0.dX + so * K0;
    if (lstm)
      hipLaunchKernelGGL(trn_lstm_bwd_kernel, dim3(nbH), dim3(256), 0, s, G0, i ? (const float *)(d0.C + sp * H) : c_init(0), (const float *)(d0.C + so * H),
                         dcur, ldcur, (const float *)t->dq, H, last ? nul : (const float *)d0.dhz, H, last ? nul : dX0n + E + H, K0,
                         last ? nul : (const float *)d0.dcz, dG0, d0.dhz, d0.dcz, B, H);
    else
      hipLaunchKernelGGL(trn_gru_bwd_kernel, dim3(nbH), dim3(256), 0, s, G0, X0 + E + H, K0, dcur, ldcur, (const float *)t->dq, H,
                         last ? nul : (const float *)d0.dhz, H, last ? nul : dX0n + E + H, K0, dG0, d0.dhz, B, H);
    TN_TRY(launch_linear_f32_lat(dG0, 4 * H, d0.wcT, 4 * H, nullptr, dX0, K0, B, K0, 4 * H, s));
  }
  // gradients reaching the encoder's final states: decoder layer j's initial state is encoder layer j's (backward direction of a bi layer)
  for (int j = 0; j < NL; ++j) {
    TrnEnc &e = t->enc[j];
    TrnDec &d = t->dec[j];
    if (e.D == 2) {
      TN_HIP_CHECK(hipMemsetAsync(e.dhl, 0, sizeof(float) * (size_t)B * H, s));
      if (lstm) TN_HIP_CHECK(hipMemsetAsync(e.dcl, 0, sizeof(float) * (size_t)B * H, s));
    }
    const size_t off = (size_t)(e.D - 1) * B * H;
    hipLaunchKernelGGL(trn_add2_kernel, dim3(nbH), dim3(256), 0, s, (const float *)d.dhz, H, (const float *)(d.dX + (j ? 2 * H : E + H)), d.K,
                       e.dhl + off, B, H);
    if (lstm) TN_HIP_CHECK(hipMemcpyAsync(e.dcl + off, d.dcz, sizeof(float) * (size_t)B * H, hipMemcpyDeviceToDevice, s));
  }
  // ---------------- backward: decoder weights, attention key projection, embedding ----------------
  for (int j = 0; j < NL; ++j) {
    TrnDec &d = t->dec[j];
    TN_TRY(launch_gemm_tn_f32(d.dG, 4 * H, d.X, d.K, d.dW, d.K, 4 * H, d.K, LB, s));
    TN_TRY(launch_colsum_f32(d.dG, 4 * H, LB, 4 * H, d.db, s));
    hipLaunchKernelGGL(trn_unstack_kernel, dim3(GH), dim3(256), 0, s, (const float *)d.dW, (const float *)d.db, d.in, H, lstm ? 1 : 0, g + d.o_wi,
                       g + d.o_wh, g + d.o_bi, g + d.o_bh);
  }
  hipLaunchKernelGGL(trn_att_outer_kernel, dim3((T + 3) / 4, B), dim3(256), 0, s, (const float *)t->AW, (const float *)t->DS, (const float *)t->DCTX,
                     (const float *)d1.X, K1, dmem, t->dkp, L, B, T, H);
  TN_TRY(launch_gemm_tn_f32(t->dkp, H, mem, H, g + t->o_wk, H, H, H, BT, s));
  TN_TRY(launch_linear_f32(t->dkp, H, t->wkT, H, nullptr, dmem, H, BT, H, H, 1, s));
  hipLaunchKernelGGL(trn_emb_grad_kernel, dim3(V), dim3(64), 0, s, (const float *)d0.dX, K0, tgt, ld, B, L, E, V, g + t->o_emb);
  // ---------------- backward: encoder, last layer to first ----------------
  for (int i = NL - 1; i >= 0; --i) {
    TrnEnc &e = t->enc[i];
    const int DG = e.D * GH;
    const long no = (long)BT * e.out;
    // e.dxn = d (what the next layer read); through the dropout mask to the recurrence's outputs
    const float *dseq = e.dxn;
    if (drop) {
      hipLaunchKernelGGL(trn_mul_add_kernel, blocks(no), dim3(256), 0, s, (const float *)e.dxn, (const float *)e.M, nul, e.dseq, no);
      dseq = e.dseq;
    }
    TN_HIP_CHECK(hipMemsetAsync(e.dgi, 0, sizeof(float) * (size_t)BT * DG, s));
    TN_HIP_CHECK(hipMemsetAsync(e.dgh, 0, sizeof(float) * (size_t)BT * DG, s));
    TN_HIP_CHECK(hipMemsetAsync(e.hp, 0, sizeof(float) * (size_t)e.D * BT * H, s));
    if (lstm) TN_TRY(launch_lstm_train_bwd(e.seq, e.sav, dseq, w + e.o_wh, e.dgi, e.hp, B, T, H, s, e.D, t->vl, e.dhl, e.dcl));
    else TN_TRY(launch_gru_train_bwd(e.seq, e.sav, dseq, w + e.o_wh, e.dgi, e.dgh, e.hp, B, T, H, s, e.D, t->vl, e.dhl));
    const float *dgh = lstm ? e.dgi : e.dgh;      // LSTM: one pre-activation gradient feeds both branches
    TN_TRY(launch_gemm_tn_f32(e.dgi, DG, xin[i], e.in, g + e.o_wi, e.in, DG, e.in, BT, s));
    TN_TRY(launch_colsum_f32(e.dgi, DG, BT, DG, g + e.o_bi, s));
    for (int d = 0; d < e.D; ++d)
      TN_TRY(launch_gemm_tn_f32(dgh + (size_t)d * GH, DG, e.hp + (size_t)d * BT * H, H, g + e.o_wh + (long)d * GH * H, H, GH, H, BT, s));
    TN_TRY(launch_colsum_f32(dgh, DG, BT, DG, g + e.o_bh, s));
    if (i > 0) {       // d (this layer's input) = d (the layer below's xn): through the input weights (+ the residual path)
      TrnEnc &below = t->enc[i - 1];
      TN_TRY(launch_linear_f32(e.dgi, DG, e.wiT, DG, nullptr, below.dxn, e.in, BT, e.in, DG, 0, s));
      if (e.res) hipLaunchKernelGGL(add_inplace_kernel, blocks((long)BT * e.in), dim3(256), 0, s, below.dxn, (const float *)e.dxn, (long)BT * e.in);
    }
  }
#undef TN_TRY
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

// dropout rate of the encoder / decoder layers (train_gnmt.py --dropout, default 0.2 there; 0 here until set) and the
// seed of the counter-based mask generator.  MXNet's own random stream cannot be reproduced: masks differ from the
// reference's, their distribution and placement do not.
extern "C" int tn_gnmt_trainer_set_dropout(tn_gnmt_trainer *t, float p, uint64_t seed) {
  TN_REQUIRE(t, "tn_gnmt_trainer_set_dropout: null handle");
  TN_REQUIRE(p >= 0.f && p < 1.f, "tn_gnmt_trainer_set_dropout: rate must be in [0, 1)");
  t->drop_p = p; t->drop_seed = seed; t->drop_count = 0;
  return TN_OK;
}
// the masks of the last forward_backward (device pointers) - test hook.  which = 0 .. num_layers - 1: encoder layer's (B*T, D H);
// num_layers + j, j = 1 .. num_layers - 1: decoder layer j's (L*B, H) step-major
extern "C" int tn_gnmt_trainer_dropout_mask(tn_gnmt_trainer *t, int which, float **mask) {
  TN_REQUIRE(t && mask, "tn_gnmt_trainer_dropout_mask: null argument");
  TN_REQUIRE(which >= 0 && which < 2 * t->NL && which != t->NL, "tn_gnmt_trainer_dropout_mask: no such mask");
  *mask = which < t->NL ? t->enc[which].M : t->dec[which - t->NL].M;
  return TN_OK;
}
// (the two-layer form of round 2: encoder layers 0 / 1 and the top decoder layer)
extern "C" int tn_gnmt_trainer_dropout_masks(tn_gnmt_trainer *t, float **m_enc0, float **m_enc1, float **m_dec) {
  TN_REQUIRE(t && m_enc0 && m_enc1 && m_dec, "tn_gnmt_trainer_dropout_masks: null argument");
  *m_enc0 = t->enc[0].M; *m_enc1 = t->enc[1].M; *m_dec = t->dec[t->NL - 1].M;
  return TN_OK;
}

extern "C" int tn_gnmt_trainer_buffers(tn_gnmt_trainer *t, float **params_dev, float **grads_dev, int64_t *numel) {
  TN_REQUIRE(t, "tn_gnmt_trainer_buffers: null handle");
  if (params_dev) *params_dev = t->w;
  if (grads_dev) *grads_dev = t->g;
  if (numel) *numel = t->n;
  return TN_OK;
}

// gluon.Trainer(params, 'adam', {'learning_rate': lr}).step(1) (train_gnmt.py:310,337): MXNet Adam, beta1 0.9, beta2 0.999,
// epsilon 1e-8, no weight decay, no clipping, rescale_grad 1
extern "C" int tn_gnmt_trainer_adam_step(tn_gnmt_trainer *t, float lr, float beta1, float beta2, float epsilon) {
  TN_REQUIRE(t, "tn_gnmt_trainer_adam_step: null handle");
  TN_ON_DEVICE(t->ctx->device);
  t->step += 1;
  const double c1 = 1.0 - pow((double)beta1, (double)t->step), c2 = 1.0 - pow((double)beta2, (double)t->step);
  const float lr_t = (float)((double)lr * sqrt(c2) / c1);
  hipLaunchKernelGGL(trn_adam_kernel, dim3((t->n + 255) / 256), dim3(256), 0, t->ctx->stream, t->w, (const float *)t->g, t->am, t->av, t->n,
                     lr_t, beta1, beta2, epsilon);
  TN_HIP_CHECK(hipGetLastError());
  return trainer_refresh(t);
}

extern "C" int tn_gnmt_trainer_read_param(tn_gnmt_trainer *t, const char *name_c, int gradient, float *out_host, int64_t capacity,
                                          int64_t *numel) {
  TN_REQUIRE(t && name_c && out_host && numel, "tn_gnmt_trainer_read_param: null argument");
  const std::string name(name_c), pre = t->prefix;
  const long H = t->H, E = t->E, V = t->V, GH = (long)t->G * H;
  long off = -1, cnt = 0;
  auto cell = [&](const std::string &c, long owi, long obi, long owh, long obh, long in) {
    if (name == pre + c + "i2h_weight") { off = owi; cnt = GH * in; }
    if (name == pre + c + "i2h_bias") { off = obi; cnt = GH; }
    if (name == pre + c + "h2h_weight") { off = owh; cnt = GH * H; }
    if (name == pre + c + "h2h_bias") { off = obh; cnt = GH; }
  };
  for (int i = 0; i < t->NL; ++i) {
    const TrnEnc &e = t->enc[i];
    for (int d = 0; d < e.D; ++d)
      cell("enc_rnn" + std::to_string(i) + (e.D == 2 ? (d ? "_r_" : "_l_") : "_"), e.o_wi + d * GH * e.in, e.o_bi + d * GH, e.o_wh + d * GH * H,
           e.o_bh + d * GH, e.in);
    const TrnDec &dc = t->dec[i];
    cell("dec_rnn" + std::to_string(i) + "_", dc.o_wi, dc.o_bi, dc.o_wh, dc.o_bh, dc.in);
  }
  if (name == pre + "dec_attention_key_weight") { off = t->o_wk; cnt = H * H; }
  if (name == pre + "tgt_proj_weight") { off = t->o_wp; cnt = V * H; }
  if (name == pre + "tgt_proj_bias") { off = t->o_bp; cnt = V; }
  if (name == pre + "tgt_embed_weight") { off = t->o_emb; cnt = V * E; }
  TN_REQUIRE(off >= 0, "tn_gnmt_trainer_read_param: unknown parameter name");
  TN_REQUIRE(capacity >= cnt, "tn_gnmt_trainer_read_param: host buffer too small");
  TN_ON_DEVICE(t->ctx->device);
  TN_HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
  TN_HIP_CHECK(hipMemcpy(out_host, (gradient ? t->g : t->w) + off, sizeof(float) * cnt, hipMemcpyDeviceToHost));
  *numel = cnt;
  return TN_OK;
}

extern "C" int tn_gnmt_trainer_destroy(tn_gnmt_trainer *t) {
  if (!t) return TN_OK;
  TnDeviceGuard tn_dg_(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  t->pool.release();
  delete t;
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
  TnDeviceGuard tn_dg_(g->ctx->device);
  for (tn_birnn *e : g->enc) tn_birnn_destroy(e);
  g->pool.release();
  delete g;
  return TN_OK;
}

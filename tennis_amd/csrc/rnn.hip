// K9/K10/K11 — gluon.rnn.GRU / LSTM (layout 'NTC', optionally bidirectional)
// and the temporal max/mean that follows it (reference
// models/vision/definitions.py:94-96,106-107 and 66-69; one bidirectional layer
// of GNMTEncoder, models/captioning/gnmt.py:141-148).  fp32 throughout so the
// sequential recurrence stays within 1e-3 of the CPU oracle.
//
//   (a) one big i2h GEMM over all B*T rows and both directions (linear.hip);
//   (b) a persistent recurrent kernel: one workgroup per (direction, group of
//       NB batch rows) walks all T steps without leaving the device.  Thread j
//       owns gate row j: it streams column j of the k-major h2h matrix
//       (L2-resident, coalesced across threads) against h held in LDS, the
//       gate pre-activations meet in LDS and the same workgroup applies the
//       gate non-linearities.  valid_length semantics of
//       BidirectionalCell.unroll(valid_length=...) [EXT]: steps >= valid_len
//       do not update state and emit zeros; the reverse pass starts at
//       valid_len-1.
#include "common.h"
#include "linear.h"
#include "rnn.h"
#include "rnn_dot.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One workgroup = one direction x NB batch rows, all T steps; thread j owns gate row j of W_hh.
//   * Per step a CU needs 4*G*H*H bytes of W_hh: at 64 B/clk of L1 fill that stream alone costs as much as the
//     arithmetic, so the first KR k-values of the thread's weight column live in REGISTERS for the whole sequence
//     (KR = H when the block is small enough for a 256-register budget: nothing is re-read), the rest is streamed
//     with 16 loads in flight (batched by hand: the compiler otherwise waits for every group of four).
//   * NB = 4 rows when the batch fills the chip, 1 otherwise: the step is also bound by FMA issue and by the LDS
//     reads of h (both ~ NB*H per thread), so small batches are spread over four times as many CUs.
// The dot product runs over k in ascending order whatever KR and NB are (bit-identical results).
template <int G, int NB, int KR, int MAXT>  // G: 3 = GRU [r,z,n], 4 = LSTM [i,f,g,o]
__global__ __launch_bounds__(MAXT) void rnn_recurrent_kernel(
    const float *__restrict__ gi, int ldgi,  // [B*T][dirs*G*H] i2h + b_i2h
    const float *__restrict__ whT,           // [dirs][H][G*H]
    const float *__restrict__ bh,            // [dirs][G*H]
    const int32_t *__restrict__ valid_len,   // [B] or null
    float *__restrict__ seq, int ldo,        // [B*T][dirs*H]
    float *__restrict__ h_last, float *__restrict__ c_last,  // [dirs][B][H]
    float *__restrict__ save,                // [dirs][B*T][(G+1)*H] or null: what BPTT needs (train.hip)
    int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = G * H;
  float *hs = lds;                 // [NB][H]
  float *gh = hs + NB * H;         // [NB][GH]
  float *cs = gh + NB * GH;        // [NB][H] (LSTM)
  const int j = threadIdx.x;       // gate row
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * NB;
  const float *wcol = whT + (long)dir * H * GH + j;
  const float bj = bh[dir * GH + j];
  float wr[KR > 0 ? KR : 1];
#pragma unroll
  for (int k = 0; k < KR; ++k) wr[k] = wcol[(long)k * GH];

  for (int i = j; i < NB * H; i += GH) {
    hs[i] = 0.f;
    if (G == 4) cs[i] = 0.f;
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = bj;
#pragma unroll
    for (int k = 0; k < KR; k += 4) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 hv = *(const float4 *)(hs + b * H + k);
        acc[b] = fmaf(wr[k + 0], hv.x, acc[b]);
        acc[b] = fmaf(wr[k + 1], hv.y, acc[b]);
        acc[b] = fmaf(wr[k + 2], hv.z, acc[b]);
        acc[b] = fmaf(wr[k + 3], hv.w, acc[b]);
      }
    }
    int k = KR;
    for (; k + 16 <= H; k += 16) {
      float w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = wcol[(long)(k + i) * GH];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 hv = *(const float4 *)(hs + b * H + k + i);
          acc[b] = fmaf(w[i + 0], hv.x, acc[b]);
          acc[b] = fmaf(w[i + 1], hv.y, acc[b]);
          acc[b] = fmaf(w[i + 2], hv.z, acc[b]);
          acc[b] = fmaf(w[i + 3], hv.w, acc[b]);
        }
      }
    }
    for (; k < H; k += 4) {
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
      const int b = idx / H, u = idx - b * H;
      const int bg = b0 + b;
      if (bg >= B) continue;
      const int vlen = valid_len ? valid_len[bg] : T;
      if (s >= vlen) continue;
      const int ti = dir ? (vlen - 1 - s) : s;
      const float *g = gi + ((long)bg * T + ti) * ldgi + dir * GH;
      const float *q = gh + b * GH;
      float hn;
      if (G == 3) {
        const float r = sigmoidf_(g[u] + q[u]);
        const float z = sigmoidf_(g[H + u] + q[H + u]);
        const float n = tanhf(g[2 * H + u] + r * q[2 * H + u]);
        hn = (1.f - z) * n + z * hs[idx];
        if (save) {
          float *sv = save + ((long)dir * B * T + (long)bg * T + ti) * (4 * H);
          sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = q[2 * H + u];
        }
      } else {
        const float ig = sigmoidf_(g[u] + q[u]);
        const float fg = sigmoidf_(g[H + u] + q[H + u]);
        const float gg = tanhf(g[2 * H + u] + q[2 * H + u]);
        const float og = sigmoidf_(g[3 * H + u] + q[3 * H + u]);
        const float c2 = fg * cs[idx] + ig * gg;
        cs[idx] = c2;
        hn = og * tanhf(c2);
        if (save) {
          float *sv = save + ((long)dir * B * T + (long)bg * T + ti) * (5 * H);
          sv[u] = ig; sv[H + u] = fg; sv[2 * H + u] = gg; sv[3 * H + u] = og; sv[4 * H + u] = c2;
        }
      }
      hs[idx] = hn;
      seq[((long)bg * T + ti) * ldo + dir * H + u] = hn;
    }
    __syncthreads();
  }
  for (int idx = j; idx < NB * H; idx += GH) {
    const int b = idx / H, u = idx - b * H;
    if (b0 + b >= B) continue;
    if (h_last) h_last[((long)dir * B + b0 + b) * H + u] = hs[idx];
    if (c_last && G == 4) c_last[((long)dir * B + b0 + b) * H + u] = cs[idx];
  }
}

// The same recurrence for the blocks whose weight columns do not fit the registers (G*H > 512: GRU / LSTM with H = 256, the
// captioner's encoder), one batch row per workgroup.  What bounded the kernel above there, per step: the L1 fill of the streamed
// weights (160 of 256 k-values x 768 rows x 4 B = 480 KB at 64 B/clk), the LDS pipe (every lane reading the same h values:
// 64 ds_read_b128 per wave) and the latency of the step's gi loads behind the barrier.  Here
//   * the k-values of a thread's column are split three ways: KR in registers, KL in LDS ([k/4][row][4]: conflict-free 16-byte
//     reads; as much as the 160 KB hold), the rest streamed with 16 loads in flight;
//   * h is not read per FMA: a lane reads 16 bytes per 16 k-values (lane l holds h[k0 + 4 (l mod 4) + e], e = 0..3) and the FMAs
//     take their h operand through DPP quad_perm:[j,j,j,j] (v_fmac_f32_dpp: lane j of the quad, broadcast to the quad) -
//     4 x fewer LDS instructions, no extra VALU work;
//   * the step's gi values are requested before the dot product.
// The dot product still runs over k in ascending order with one accumulator: results are bit-identical to the kernel above.
template <int G, int KR, int KL, int MAXT>
__global__ __launch_bounds__(MAXT) void rnn_recurrent_big_kernel(
    const float *__restrict__ gi, int ldgi, const float *__restrict__ whT, const float *__restrict__ bh,
    const int32_t *__restrict__ valid_len, float *__restrict__ seq, int ldo, float *__restrict__ h_last,
    float *__restrict__ c_last, float *__restrict__ save, int B, int T, int H) {
  static_assert(KR % 16 == 0 && KL % 16 == 0, "whole 16-wide h chunks");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int GH = G * H;
  float *hs = lds;                 // [H]
  float *gh = hs + H;              // [GH]
  float *cs = gh + GH;             // [H] (LSTM)
  float *wl = cs + H;              // [KL/4][GH][4]
  const int j = threadIdx.x, lane4 = (j & 3) * 4;
  const int dir = blockIdx.y, bg = blockIdx.x;
  const float *wcol = whT + (long)dir * H * GH + j;
  const float bj = bh[dir * GH + j];
  float wr[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) wr[k] = wcol[(long)k * GH];
  rnn_dot_fill_lds<KR, KL>(wl, GH, j, wcol, GH);
  if (j < H) {
    hs[j] = 0.f;
    if (G == 4) cs[j] = 0.f;
  }
  const int vlen = valid_len ? valid_len[bg] : T;
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    // this step's input pre-activations of the units the thread finishes below (requested now, used after the barrier)
    const bool live = j < H && s < vlen;
    const int ti = dir ? (vlen - 1 - s) : s;
    float gq[G];
    if (live) {
      const float *g = gi + ((long)bg * T + ti) * ldgi + dir * GH;
#pragma unroll
      for (int e = 0; e < G; ++e) gq[e] = g[e * H + j];
    }
    const float acc = rnn_dot_big<KR, KL>(bj, wr, wl, GH, j, wcol, GH, hs, H, lane4);
    gh[j] = acc;
    __syncthreads();
    if (live) {
      const int u = j;
      float hn;
      if (G == 3) {
        const float r = sigmoidf_(gq[0] + gh[u]);
        const float z = sigmoidf_(gq[1] + gh[H + u]);
        const float n = tanhf(gq[2] + r * gh[2 * H + u]);
        hn = (1.f - z) * n + z * hs[u];
        if (save) {
          float *sv = save + ((long)dir * B * T + (long)bg * T + ti) * (4 * H);
          sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = gh[2 * H + u];
        }
      } else {
        const float ig = sigmoidf_(gq[0] + gh[u]);
        const float fg = sigmoidf_(gq[1] + gh[H + u]);
        const float gg = tanhf(gq[2] + gh[2 * H + u]);
        const float og = sigmoidf_(gq[G - 1] + gh[3 * H + u]);
        const float c2 = fg * cs[u] + ig * gg;
        cs[u] = c2;
        hn = og * tanhf(c2);
        if (save) {
          float *sv = save + ((long)dir * B * T + (long)bg * T + ti) * (5 * H);
          sv[u] = ig; sv[H + u] = fg; sv[2 * H + u] = gg; sv[3 * H + u] = og; sv[4 * H + u] = c2;
        }
      }
      hs[u] = hn;
      seq[((long)bg * T + ti) * ldo + dir * H + u] = hn;
    }
    __syncthreads();
  }
  if (j < H) {
    if (h_last) h_last[((long)dir * B + bg) * H + j] = hs[j];
    if (c_last && G == 4) c_last[((long)dir * B + bg) * H + j] = cs[j];
  }
}

__global__ void temporal_pool_kernel(const float *__restrict__ x, int B, int T, int F, int kind,
                                     float *__restrict__ y) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)B * F) return;
  const int b = (int)(id / F), f = (int)(id % F);
  const float *p = x + (long)b * T * F + f;
  float acc = kind == TN_POOL_MAX ? -INFINITY : 0.f;
  for (int t = 0; t < T; ++t) {
    const float v = p[(long)t * F];
    acc = kind == TN_POOL_MAX ? fmaxf(acc, v) : acc + v;
  }
  y[id] = kind == TN_POOL_MAX ? acc : acc / (float)T;
}

__global__ void prf1_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels, int rows,
                            int classes, unsigned long long *__restrict__ mat) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float *p = logits + (long)r * classes;
  int best = 0;
  float bv = p[0];
  for (int c = 1; c < classes; ++c)
    if (p[c] > bv) { bv = p[c]; best = c; }   // first maximum, like ndarray.argmax
  const int lab = labels[r];
  if (lab >= 0 && lab < classes) atomicAdd(mat + (long)lab * classes + best, 1ULL);
}

}  // namespace

int launch_rnn_recurrent(int gates, const float *gi, int ldgi, const float *whT, const float *bh,
                         const int32_t *valid_len, float *seq, int ldo, float *h_last, float *c_last, int B, int T,
                         int H, int dirs, hipStream_t s, float *save) {
  TN_REQUIRE(gates == 3 || gates == 4, "rnn: gates must be 3 or 4");
  TN_REQUIRE(gates * H <= 1024 && H % 4 == 0, "rnn: gates*hidden must be <= 1024 and hidden % 4 == 0");
  const int threads = gates * H;
  // rows per workgroup: 4 when that still gives every CU a workgroup, else 1 (latency-bound small batches)
  const int nb = ((B + 3) / 4) * dirs >= 256 ? 4 : 1;
  // register-resident prefix of each weight column, bounded by the VGPR budget the block size leaves
  const int kr = (threads <= 512 && H >= 128) ? 128 : (threads <= 768 && H >= 96) ? 96 : H >= 64 ? 64 : 0;
  const dim3 grid((B + nb - 1) / nb, dirs), block(threads);
  const size_t lds = (size_t)(nb * H * 2 + nb * gates * H) * sizeof(float);
#define TN_RNN_LAUNCH(G_, NB_, KR_, MT_)                                                                               \
  hipLaunchKernelGGL((rnn_recurrent_kernel<G_, NB_, KR_, MT_>), grid, block, lds, s, gi, ldgi, whT, bh, valid_len, seq, \
                     ldo, h_last, c_last, save, B, T, H)
#define TN_RNN_PICK(G_, NB_)                                   \
  do {                                                         \
    if (kr == 128) TN_RNN_LAUNCH(G_, NB_, 128, 512);           \
    else if (kr == 96) TN_RNN_LAUNCH(G_, NB_, 96, 768);        \
    else if (kr == 64) TN_RNN_LAUNCH(G_, NB_, 64, 1024);       \
    else TN_RNN_LAUNCH(G_, NB_, 0, 1024);                      \
  } while (0)
  // one row per workgroup and a column that does not fit the registers (H = 256): registers + LDS + stream, h through DPP
  if (nb == 1 && H == 256) {
    constexpr int KR3 = 112, KL3 = 48, KR4 = 64, KL4 = 32;
    const int kl = gates == 3 ? KL3 : KL4;
    const size_t lds2 = (size_t)(2 * H + gates * H + kl * gates * H) * sizeof(float);
    if (gates == 3) {
      TN_SET_ATTR_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)rnn_recurrent_big_kernel<3, KR3, KL3, 768>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((rnn_recurrent_big_kernel<3, KR3, KL3, 768>), grid, block, lds2, s, gi, ldgi, whT, bh, valid_len, seq, ldo, h_last,
                         c_last, save, B, T, H);
    } else {
      TN_SET_ATTR_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)rnn_recurrent_big_kernel<4, KR4, KL4, 1024>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((rnn_recurrent_big_kernel<4, KR4, KL4, 1024>), grid, block, lds2, s, gi, ldgi, whT, bh, valid_len, seq, ldo, h_last,
                         c_last, save, B, T, H);
    }
    TN_HIP_CHECK(hipGetLastError());
    return TN_OK;
  }
  // (register-resident weights only with one row per workgroup: with four the unrolled prefix spills)
  if (gates == 3) { if (nb == 4) TN_RNN_LAUNCH(3, 4, 0, 1024); else TN_RNN_PICK(3, 1); }
  else { if (nb == 4) TN_RNN_LAUNCH(4, 4, 0, 1024); else TN_RNN_PICK(4, 1); }
#undef TN_RNN_PICK
#undef TN_RNN_LAUNCH
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_temporal_pool(const float *x, int B, int T, int F, int kind, float *y, hipStream_t s) {
  const long total = (long)B * F;
  hipLaunchKernelGGL(temporal_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, B, T, F, kind,
                     y);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_prf1(const float *logits, const int32_t *labels, int rows, int classes, int64_t *mat, hipStream_t s) {
  hipLaunchKernelGGL(prf1_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, logits, labels, rows, classes,
                     (unsigned long long *)mat);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

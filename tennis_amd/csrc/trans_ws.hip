// The last two transitions of DenseNet-121 (BatchNorm - ReLU - Conv1x1 - AvgPool 2x2: 512 -> 256 on the 28 x 28 map, 1024 -> 512 on
// the 14 x 14 one; reference call site models/vision/definitions.py:30 -> gluoncv DenseNet _make_transition) as a WARP-SPECIALISED
// kernel (round 6).
//
// conv1x1.hip runs a transition as a tiled GEMM whose waves alternate between two phases behind two barriers per k-tile: staging
// (load, BN + ReLU, average of the four source pixels, hi + lo split, LDS store: ~390 VALU instructions per wave and k-tile) and
// multiplying (64 MFMAs).  The SQ counters of that kernel on the last transition (profiles/r06_d_transitions_sq_pmc.txt): VALU
// 28 %, MFMA 18 %, LDS 6 %, waiting 48 % of a wave's life - the two phases never overlap, and neither wider tiles, nor more waves,
// nor a deeper prefetch, nor the 32x32 MFMA shape changed the 50 us it takes (docs/kernels.md).  Here the two phases are two KINDS
// of wave:
//
// * a workgroup = one tile of a frame's pooled pixels x ALL output channels (TG below: 64 pixels x 512 channels - a 7 x 7 frame is
//   one tile of 49 rows - or 128 x 256 - a 14 x 14 frame is two tiles of 98), eight waves = two per SIMD, one workgroup per CU;
// * waves 0 - 3 STAGE: global -> registers two k-tiles ahead, BN + ReLU + average + hi / lo split exactly as conv1x1.hip does it
//   (same arithmetic, same order: the operand tiles are the same bits), LDS store into one of two tile buffers;
// * waves 4 - 7 MULTIPLY: pixel fragments from the other tile buffer, weight fragments straight from global memory in MFMA
//   operand order (pack_trans_frags: 1 KiB per wave-load, one k-tile ahead in registers; the weights are L2-resident and the LDS
//   holds nothing but the pixel tiles), 64 v_mfma_f32_32x32x16_f16 per k-tile into 128 accumulator registers (the fp32 sums run
//   in a different order than the tiled kernel's 16x16x32 ones: results agree to an fp16 ulp);
// * ONE barrier per k-tile: a stager's VALU instructions issue beside the multiplier wave that shares its SIMD.
//
// The pixel tile is pooled once (conv1x1.hip: once per 256-channel column tile).  Measured at batch 256: 70 -> 63 us and 55 -> 44 us
// alone; knock-out builds of the last transition: 29 us with neither staging arithmetic nor MFMAs - the kernel's memory time, at
// the boxes' practical read rate - so the two kinds of wave still add more than they hide.  Inside the pipelined step the gain is
// not visible (139.8 k against 139.7 k frames/s): there a kernel costs the CU-time it occupies, and this one holds a CU alone.
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

#ifndef TN_TWS_EXP
#define TN_TWS_EXP 0   // timing experiments only (results wrong): bit 0 the stagers store raw values (no BN / average / split), bit 1 no MFMAs, bit 2 no weight-fragment loads inside the loop
#endif

namespace {

constexpr int BK = 64;                    // channels per k-tile
// Two tile shapes, each 128 accumulator registers per multiplier wave: 64 pooled pixels x 512 channels (the last transition) and
// 128 pixels x 256 channels (the second one).  A frame is cut into ceil(P / BM) tiles of equal size PT <= BM (14 x 14 -> 7 x 7:
// one tile of 49; 28 x 28 -> 14 x 14: two of 98; the 32 x 32 / 16 x 16 maps of a 512 x 512 input: eight of 128 / four of 64).
template <int BM, int NB>
struct TG {
  static constexpr int XT = BM * 128;              // one pixel tile (hi or lo) of a k-tile
  static constexpr int CPITCH = NB * 2 + 16;       // epilogue row: NB halves + 8 of padding
  static constexpr int LDS_BYTES = BM * CPITCH > 4 * XT ? BM * CPITCH : 4 * XT;
  static constexpr int SR = BM / 32;               // tile rows a stager thread owns
  static constexpr int NCT = NB / 4 / 32, NPT = BM / 32;      // 32-channel x 32-pixel accumulator tiles of a multiplier wave
  static_assert(NCT * NPT == 8, "128 accumulator registers");
};

// workgroup barrier that orders LDS traffic only (__syncthreads() also drains the global loads in flight)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int V>
using ic = std::integral_constant<int, V>;

// NK: k-tiles known at compile time (K = 1024: 16; 0 = run-time count).  hipcc's wait-count pass puts an s_waitcnt vmcnt(0) at the head
// of a LOOP whose body carries loads across the back edge (seen in the ISA: the stagers then wait for the tile they requested a
// moment ago, once per trip) - fully unrolled, every wait is counted exactly and the newest tile stays in flight.
template <int NK, int BM, int NB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void trans_ws_kernel(Conv1x1Args a, int TPF, int PT) {
  using G = TG<BM, NB>;
  constexpr int XT = G::XT, CPITCH = G::CPITCH, SR = G::SR, NCT = G::NCT, NPT = G::NPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int Wo = a.W >> 1, Ho = a.H >> 1, P = Wo * Ho;      // pooled pixels per frame
  const int frame = (int)blockIdx.x / TPF, p0 = ((int)blockIdx.x % TPF) * PT;      // this tile: pooled pixels p0 .. p0 + nrow - 1 of the frame
  const int nrow = P - p0 < PT ? P - p0 : PT;
  const int K = a.K, nk = NK ? NK : K / BK;

  if (wid < 4) {
    // ================= stagers: tile i -> buffer i & 1 =================
    const int c = t & 7, r0 = t >> 3;               // 16-byte chunk column, tile rows r0 + 32 i
    const f16 *xsrc[SR][4];
#pragma unroll
    for (int i = 0; i < SR; ++i) {
      int p = r0 + 32 * i;
      p = p0 + (p < nrow ? p : nrow - 1);           // rows past the tile repeat its last pixel (computed, never stored)
      const int py = p / Wo, px = p - py * Wo;
      const long base = ((long)(frame * a.H + 2 * py) * a.W + 2 * px);
      xsrc[i][0] = a.x + base * a.ldx;
      xsrc[i][1] = a.x + (base + 1) * a.ldx;
      xsrc[i][2] = a.x + (base + a.W) * a.ldx;
      xsrc[i][3] = a.x + (base + a.W + 1) * a.ldx;
    }
    f16x8 xr[2][SR][4];
    float scb[2][8], shb[2][8];
    // (unconditional loads, k-tiles past the end repeat the last one: a load under a branch makes hipcc drain the queue at the next use)
    auto load = [&](int kt, auto b_tag) {
      constexpr int PB = decltype(b_tag)::value;
      kt = kt < nk ? kt : nk - 1;
      const int kc = kt * BK + c * 8;
      const float4 s0 = *(const float4 *)(a.scale + kc), s1 = *(const float4 *)(a.scale + kc + 4);
      const float4 t0 = *(const float4 *)(a.shift + kc), t1 = *(const float4 *)(a.shift + kc + 4);
      float (&sc)[8] = scb[PB], (&sh)[8] = shb[PB];
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
#pragma unroll
      for (int i = 0; i < SR; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) xr[PB][i][s] = *(const f16x8 *)(xsrc[i][s] + kc);
    };
    auto stage = [&](auto b_tag) {
      constexpr int PB = decltype(b_tag)::value;
      unsigned char *Xs = smem + PB * 2 * XT, *Xl = Xs + XT;
#pragma unroll
      for (int i = 0; i < SR; ++i) {
        f16x8 v, vl;
        if (TN_TWS_EXP & 1) { v = xr[PB][i][0]; vl = xr[PB][i][1]; } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // (conv1x1.hip's store_tile, POOL: same operations in the same order)
          float acc = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) acc += fmaxf(fmaf((float)xr[PB][i][s][j], scb[PB][j], shb[PB][j]), 0.f);
          const float m = 0.25f * acc;
          v[j] = (f16)m;
          vl[j] = (f16)(m - (float)v[j]);
        }
        *(f16x8 *)(Xs + swz<128>(r0 + 32 * i, c)) = v;
        *(f16x8 *)(Xl + swz<128>(r0 + 32 * i, c)) = vl;
      }
    };
    load(0, ic<0>{});
    load(1, ic<1>{});
#pragma unroll
    for (int i0 = 0; i0 < nk; i0 += 2) {      // (nk is even: no conditional around a load)
      stage(ic<0>{});
      load(i0 + 2, ic<0>{});
      lds_barrier();
      stage(ic<1>{});
      load(i0 + 3, ic<1>{});
      lds_barrier();
    }
    lds_barrier();                              // (the multipliers' barrier behind the last tile)
  } else {
    // ================= multipliers: tile i - 1 from buffer (i - 1) & 1 =================
    // v_mfma_f32_32x32x16_f16: A = 32 channels x 16 k of the weights, B = 32 pixels x 16 k (lane l: row l & 31, k = 8 (l >> 5) .. + 7).
    // A 32-cycle MFMA leaves its SIMD ~6 issue slots for the stager wave that shares it, a 16-cycle 16x16x32 about one
    // (scripts/microbench/slotbench.hip): with the 16x16x32 shape this kernel took 45.6 us = the sum of its staging VALU time,
    // its MFMA time and its memory time (knock-out builds, TN_TWS_EXP: 36.0 without the staging arithmetic, 33.3 without the
    // MFMAs, 27.4 without either).
    const int wn = wid - 4;                          // channels (NB / 4) wn .. + NB / 4 - 1
    const int r32 = lane & 31, h32 = lane >> 5;
    f32x16 acc[NCT][NPT];                            // [32-channel tile][32-pixel tile]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
    // weight fragments (k-step g of 16 channels, channel tile NCT wn + ct) at ((g * (NB / 32) + NCT wn + ct) * 64 + lane) * 16 bytes.
    // A ring of four k-steps = one k-tile: the slot of step s is refilled with step s of the NEXT tile right behind its MFMAs,
    // 2 048 MFMA-cycles ahead of its use (an L2 hit is about half of that); no conditional inside the loop body - a load under a
    // branch makes hipcc's wait-count pass drain the queue at the next use
    const f16x8 *wf = (const f16x8 *)a.wfrag + (size_t)(wn * NCT) * 64 + lane;
    f16x8 wa[4][NCT];
    auto load_w = [&](int kt, auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      kt = kt < nk ? kt : nk - 1;        // (past the end: the last tile once more)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) wa[S][ct] = wf[((size_t)(kt * 4 + S) * (NB / 32) + ct) * 64];
    };
    auto mult = [&](int kt, auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      const unsigned char *Xs = smem + (kt & 1) * 2 * XT, *Xl = Xs + XT;
      f16x8 xb[NPT], xl[NPT];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        xb[pt] = *(const f16x8 *)(Xs + swz<128>(pt * 32 + r32, S * 2 + h32));
        xl[pt] = *(const f16x8 *)(Xl + swz<128>(pt * 32 + r32, S * 2 + h32));
      }
      if (TN_TWS_EXP & 2) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[0][pt][0] += (float)xb[pt][0] + (float)xl[pt][0] + (float)wa[S][pt % NCT][0];
      } else {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[S][ct], xb[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[S][ct], xl[pt], acc[ct][pt], 0, 0, 0);
      }
      if (!(TN_TWS_EXP & 4)) load_w(kt + 1, s_tag);
    };
    load_w(0, ic<0>{});
    load_w(0, ic<1>{});
    load_w(0, ic<2>{});
    load_w(0, ic<3>{});
    lds_barrier();                       // tile 0 is staged
#pragma unroll
    for (int kt = 0; kt < nk; ++kt) {
      mult(kt, ic<0>{});
      mult(kt, ic<1>{});
      mult(kt, ic<2>{});
      mult(kt, ic<3>{});
      lds_barrier();
    }
    // every multiplier is behind its last fragment read (the loop's last barrier): the tile buffers become the output tile.
    // D: lane holds pixel 32 pt + (l & 31), channels 32 ct + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = pt * 32 + r32, n = wn * (NB / 4) + ct * 32 + 8 * g + 4 * h32;
          f16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (f16)acc[ct][pt][4 * g + r];
          *(f16x4 *)(smem + m * CPITCH + n * 2) = h;
          if (a.y32 && m < nrow)      // the un-rounded result for the head, 16 B per lane
            *(float4 *)(a.y32 + ((long)frame * P + p0 + m) * a.ld32 + n) =
                make_float4(acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
        }
  }
  __syncthreads();
  // coalesced rows: 64 threads x 16 B per pixel
  for (int id = t; id < nrow * (NB / 8); id += 512) {
    const int row = id / (NB / 8), ch = id % (NB / 8);
    const uint4 v = *(const uint4 *)(smem + row * CPITCH + ch * 16);
    *(uint4 *)(a.y + ((long)frame * P + p0 + row) * a.ldy + a.yoff + ch * 8) = v;
  }
}

// [N][K] fp16 -> v_mfma_f32_32x32x16_f16 A-operand fragments [K / 16 k-steps][N / 32 channel tiles][64 lanes][8]: lane l holds row l & 31, k = 8 (l >> 5) .. + 7
__global__ void pack_trans_frags_kernel(const f16 *__restrict__ w, int N, int K, f16 *__restrict__ out) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte cell
  if (id >= (long)N * K / 8) return;
  const int l = (int)(id & 63);
  const long f = id >> 6;
  const int ct = (int)(f % (N / 32)), g = (int)(f / (N / 32));
  *(f16x8 *)(out + id * 8) = *(const f16x8 *)(w + (long)(ct * 32 + (l & 31)) * K + g * 16 + 8 * (l >> 5));
}

}  // namespace

// host form of pack_trans_frags_kernel (api.hip packs a model's weights once, at create)
std::vector<f16> pack_trans_frags(const f16 *w, int N, int K) {
  std::vector<f16> out((size_t)N * K);
  for (int g = 0; g < K / 16; ++g)
    for (int ct = 0; ct < N / 32; ++ct)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j)
          out[(((size_t)g * (N / 32) + ct) * 64 + l) * 8 + j] = w[(size_t)(ct * 32 + (l & 31)) * K + g * 16 + 8 * (l >> 5) + j];
  return out;
}

bool trans_ws_supported(const Conv1x1Args &a) {
  return a.pool && !a.exact && !a.bias && (a.N == 512 || a.N == 256) && a.K % (2 * BK) == 0 && a.H % 2 == 0 && a.W % 2 == 0 && a.H >= 2 && a.W >= 2 &&
         a.M % ((a.H / 2) * (a.W / 2)) == 0;
}

int launch_pack_trans_frags(const f16 *w, int N, int K, f16 *out, hipStream_t s) {
  TN_REQUIRE(N % 32 == 0 && K % 16 == 0, "pack_trans_frags: N % 32 or K % 16");
  const long cells = (long)N * K / 8;
  hipLaunchKernelGGL(pack_trans_frags_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, w, N, K, out);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

template <int NK, int BM, int NB>
static int launch_tws(const Conv1x1Args &a, hipStream_t s) {
  using G = TG<BM, NB>;
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)trans_ws_kernel<NK, BM, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)));
  const int P = (a.H / 2) * (a.W / 2), TPF = (P + BM - 1) / BM, PT = (P + TPF - 1) / TPF;
  hipLaunchKernelGGL((trans_ws_kernel<NK, BM, NB>), dim3(a.M / P * TPF), dim3(512), G::LDS_BYTES, s, a, TPF, PT);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

int launch_trans_ws(const Conv1x1Args &a, hipStream_t s) {
  TN_REQUIRE(trans_ws_supported(a) && a.wfrag, "trans_ws: unsupported geometry");
  if (a.N == 512) return a.K == 16 * BK ? launch_tws<16, 64, 512>(a, s) : launch_tws<0, 64, 512>(a, s);
  return a.K == 8 * BK ? launch_tws<8, 128, 256>(a, s) : launch_tws<0, 128, 256>(a, s);
}

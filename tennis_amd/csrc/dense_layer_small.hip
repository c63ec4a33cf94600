// Fused DenseNet dense layer (SURVEY §7 H1, §2c rows K2+K3+K4):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// one launch per layer, replacing gluoncv's BatchNorm-Activation-Conv1x1-BatchNorm-
// Activation-Conv3x3-Concat chain (reference call site models/vision/definitions.py:30).
// The 128-channel bottleneck never goes to HBM.
//
// A workgroup of 4 waves owns a tile of ROUT x WT output pixels of one frame (28x7 in
// the 56^2 and 28^2 blocks, a whole frame in the 14^2 and 7^2 blocks) and keeps the
// bottleneck of the tile plus a one-pixel halo in an LDS tile of (ROUT+2) x (WT+2) pixel
// slots of 256 B.  Tiles are small enough for >= 2 workgroups per CU, so one workgroup's
// load-bound phase A overlaps another's MFMA/LDS-bound phase B (measured: a CU streams
// only ~15-17 B/clk global->LDS, so the two phases must not serialise).
//
// Phase A (bottleneck GEMM, one row per tile slot, N = 128, K): raw activations, the 1x1
//   weights and the BN1 scale/shift slice stream global -> LDS by LDS-DMA
//   (global_load_lds_dwordx4) through an NST-stage ring with counted vmcnt waits and ONE
//   raw s_barrier per 32-channel k-tile.  The DMA destination is lane-linear, so the
//   bank-conflict swizzle is applied to the per-lane SOURCE address and again on the
//   fragment read.  BN1+ReLU (fp32 math, one rounding) is applied to the pixel fragment
//   after the ds_read; a wave owns all 128 output channels of its rows, so every element
//   is transformed once.  v_mfma_f32_16x16x32_f16, weight fragment as the A operand.
// Epilogue A: BN2+ReLU in fp32 -> fp16 into the tile; slots outside the frame get ZERO
//   (the convolution pads after the activation), so no separate padding pass exists.
// Phase B: the 3x3 is a constant-offset walk over the flattened tile with
//   v_mfma_f32_32x32x16_f16 — no bounds logic; K = 1152 is split by channel halves over
//   wave pairs (partials meet in LDS); packed 3x3 weights stream through an LDS ring, one
//   tap at a time, three taps of prefetch in registers.
// Writing the 32 output channels at channel offset K of the same buffer IS the concat.
// blockIdx is remapped so that all tiles of a frame run on one XCD: halo pixels shared by
// neighbouring tiles are then L2 hits instead of second HBM reads.
#include <type_traits>

#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int NW = 4;          // waves per workgroup
constexpr int NT = NW * 64;    // threads
constexpr int BK = 32;         // channels per k-tile (64-byte staged rows)

template <int WT, int ROUT, int BM, int NST, int NRING>
struct DLGeom {
  static constexpr int WP = WT + 2;
  static constexpr int TR = ROUT + 2;
  static constexpr int NSLOT = TR * WP;
  static constexpr int NF = (ROUT * WP + 31) / 32;           // 32-slot output fragments
  static constexpr int NG = NW / 2;                          // fragment groups (wave pairs)
  static constexpr int MAXF = (NF + NG - 1) / NG;            // fragments per group
  static constexpr int TILE_BYTES = NSLOT * 256;
  static constexpr int XS = BM * 64, WS = 128 * 64, TS = 1024, STAGE = XS + WS + TS;
  static constexpr int XPIECES = XS / 1024, PIECES = XPIECES + WS / 1024, PPW = PIECES / NW;
  static constexpr int RING_A = NST * STAGE;
  static constexpr int W3RING = TILE_BYTES;                  // NRING x 8 KiB of 3x3 weights
  static constexpr int RED_BYTES = NG * MAXF * 4 * 1024;
  static constexpr int B0 = TILE_BYTES + NRING * 8192;
  static constexpr int LDS_BYTES = B0 > RING_A ? B0 : RING_A;
  static constexpr int MIW = BM / (16 * NW);                 // 16-row pixel fragments per wave
  static_assert(PIECES % NW == 0, "DMA pieces must divide over the waves");
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups must fit one CU");
  static_assert(RED_BYTES <= TILE_BYTES + NRING * 8192, "reduction buffer does not fit");
  static_assert(BM >= NSLOT && BM % (16 * NW) == 0, "phase A tile too small");
};

// 16-B chunk swizzle inside a 64-byte staged row: chunk ^ g((row>>2)&3), g = {0,2,3,1};
// makes every ds_read_b128 lane group of a 16x32 MFMA operand fragment conflict-free
__device__ __forceinline__ int stage_swz(int row, int chunk) {
  return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

// One LDS-DMA piece: 64 lanes x 16 B, global (per-lane address) -> LDS (wave-uniform base
// + lane*16).  Issued through inline asm on purpose: hipcc treats the builtin as an LDS
// store it must order against every later ds_read and inserts s_waitcnt vmcnt(0) in front
// of the fragment reads, which drains the whole ring each k-tile.  Hidden in asm, the only
// waits are the counted ones below (cdna_hip_programming.md 5.7; M0 saved/restored).
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// CHAIN: see dense_layer_big.hip -- a.nchain consecutive layers on the workgroup's own frame in one launch
template <int WT, int ROUT, int BM, int NST, int NRING, bool CHAIN>
__global__ __launch_bounds__(NT) void dense_layer_kernel(DenseLayerArgs a) {
  using G = DLGeom<WT, ROUT, BM, NST, NRING>;
  constexpr int WP = G::WP, NSLOT = G::NSLOT, NF = G::NF, MAXF = G::MAXF, MIW = G::MIW, PPW = G::PPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem;                   // bottleneck tile (aliases the DMA ring)
  unsigned char *ring = smem + G::W3RING;       // 3x3 weight ring
  unsigned char *red = smem;                    // phase B partial sums (aliases the tile)

  const int t = threadIdx.x;
  const int H = a.H, WI = a.W, ldc = a.ldc;
#define DL_STAMP(i) do { if (a.ts && t == 0) a.ts[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  DL_STAMP(0);
  // ---- which tile: all tiles of a frame share blockIdx % 8 (the XCD) when B % 8 == 0 ----
  const int tcols = WI / WT, tpi = (H / ROUT) * tcols;
  int img, within;
  if ((a.B & 7) == 0) {
    const int j = blockIdx.x >> 3;
    img = (j / tpi) * 8 + (blockIdx.x & 7);
    within = j % tpi;
  } else {
    img = blockIdx.x / tpi;
    within = blockIdx.x % tpi;
  }
  const int r0 = (within / tcols) * ROUT, x0 = (within % tcols) * WT;   // first output pixel
  const f16 *ibase = a.buf + (long)img * H * WI * ldc;

  // ======================= phase A: bottleneck = conv1x1(relu(bn1(x))) =======================
  // row r of the GEMM is tile slot r; slots outside the frame read a clamped address and
  // are zeroed in the epilogue
  auto slot_pixel = [&](int s, bool &inimg) -> int {
    const int tr = s / WP, xp = s - tr * WP;
    const int y = r0 - 1 + tr, x = x0 - 1 + xp;
    inimg = s < NSLOT && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)WI;
    return inimg ? y * WI + x : r0 * WI + x0;
  };
  const int nlayers = CHAIN ? a.nchain : 1;
  const int K0 = a.K;
  for (int layer = 0; layer < nlayers; ++layer) {
  // per-layer copies of the thread coordinates, laundered so that the compiler does not hoist every
  // address computation of the body out of the layer loop (that costs ~80 VGPRs and spills)
  int t_ = threadIdx.x;
  if constexpr (CHAIN) asm volatile("" : "+v"(t_));
  const int t = t_;
  const int lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  if constexpr (CHAIN) {
    const DenseLayerDev d = a.chain[layer];
    a.K = K0 + 32 * layer;
    a.s1 = d.s1; a.t1 = d.t1; a.w1 = d.w1; a.s2 = d.s2; a.t2 = d.t2; a.w3p = d.w3p;
  }
  const int K = a.K;
  const f16 *src[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = wid + NW * j;                  // wave-uniform; pieces interleave over waves
    const int prow = lane >> 2, p = lane & 3;        // row inside the piece, linear chunk position
    if (piece < G::XPIECES) {
      const int row = piece * 16 + prow;
      bool in;
      const int pix = slot_pixel(row, in);
      src[j] = ibase + (long)pix * ldc + stage_swz(row, p) * 8;
    } else {
      const int row = (piece - G::XPIECES) * 16 + prow;
      src[j] = a.w1 + (long)row * K + stage_swz(row, p) * 8;
    }
  }
  // wave 0: BN1 slice, lanes 0..7 scale, 8..15 shift; lanes >= 16 re-load the same words (the
  // piece is issued with a full exec mask so that it counts in vmcnt like every other piece)
  const float *tsrc = ((lane & 8) ? a.t1 : a.s1) + (lane & 7) * 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem);   // LDS byte address of smem
  auto issue = [&](int st) {
    const unsigned sb = lds0 + st * G::STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      dma16(src[j], sb + (wid + NW * j) * 1024);
      src[j] += BK;
    }
    if (wid == 0) {
      dma16(tsrc, sb + G::XS + G::WS);
      tsrc += BK;
    }
  };
  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s);

  // request the first three taps of the 3x3 weights already now (two 16-B pieces per thread per
  // tap): their latency hides behind the whole K loop instead of stalling epilogue A
  const f16x8 *w3 = (const f16x8 *)a.w3p + t;
  f16x8 wq[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    wq[i][0] = w3[i * 512];
    wq[i][1] = w3[i * 512 + 256];
  }

  const int frow = lane & 15, fch = lane >> 4;
  f32x4 acc[8][MIW];
#pragma unroll
  for (int ni = 0; ni < 8; ++ni)
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int st = 0;
  unsigned long long tseg[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define DL_SEG(i) do { if (a.ts) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tseg[i] += n_ - tprev; tprev = n_; } } while (0)
  // one k-tile: wait until stage kt has landed (YOUNGER = stages issued after it that may still
  // be in flight), barrier, refill the slot everyone just finished with, compute
  auto ktile = [&](int kt, auto younger_tag) {
    constexpr int YOUNGER = decltype(younger_tag)::value;
    if (wid == 0) wait_vmcnt<YOUNGER * (PPW + 1)>(); else wait_vmcnt<YOUNGER * PPW>();
    DL_SEG(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DL_SEG(1);
    if (kt == 0) DL_STAMP(1);
    if (kt + NST - 1 < nk) issue(st == 0 ? NST - 1 : st - 1);   // slot (kt+NST-1) % NST
    DL_SEG(2);
    const unsigned char *Xs = smem + st * G::STAGE;
    const unsigned char *Ws = Xs + G::XS;
    const float *tb = (const float *)(Ws + G::WS);
    const float4 s0 = *(const float4 *)(tb + fch * 8), s1 = *(const float4 *)(tb + fch * 8 + 4);
    const float4 t0 = *(const float4 *)(tb + 32 + fch * 8), t1 = *(const float4 *)(tb + 32 + fch * 8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    f16x8 wa[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int row = ni * 16 + frow;
      wa[ni] = *(const f16x8 *)(Ws + row * 64 + (stage_swz(row, fch) << 4));
    }
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
      const int row = wid * (BM / NW) + mi * 16 + frow;
      const f16x8 xraw = *(const f16x8 *)(Xs + row * 64 + (stage_swz(row, fch) << 4));
      const f16x8 xb = bn_relu8(xraw, sc, sh);
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb, acc[ni][mi], 0, 0, 0);
    }
    st = st == NST - 1 ? 0 : st + 1;
    DL_SEG(3);
  };
  {
    int kt = 0;
    for (; kt + (NST - 2) < nk; ++kt) ktile(kt, std::integral_constant<int, NST - 2>{});   // steady state
    if constexpr (NST == 3)
      for (; kt < nk; ++kt) ktile(kt, std::integral_constant<int, 0>{});                   // drain
  }
  if (a.ts && t == 0) { a.ts[(long)gridDim.x * 8 + (long)blockIdx.x * 4 + 0] = tseg[0]; a.ts[(long)gridDim.x * 8 + (long)blockIdx.x * 4 + 1] = tseg[1];
                        a.ts[(long)gridDim.x * 8 + (long)blockIdx.x * 4 + 2] = tseg[2]; a.ts[(long)gridDim.x * 8 + (long)blockIdx.x * 4 + 3] = tseg[3]; }
  __syncthreads();   // every wave is done reading the DMA ring; the tile may now be written

  DL_STAMP(2);
  DL_STAMP(3);
  // ---- epilogue A: BN2 + ReLU -> fp16 into the tile; out-of-frame slots get zero ----
  // D[i=n][j=m]: lane holds channels n = ni*16 + fch*4 + r of tile slot s = .. + frow
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi) {
    const int s = wid * (BM / NW) + mi * 16 + frow;
    bool in;
    (void)slot_pixel(s, in);
    unsigned char *dst = tile + s * 256 + (fch & 1) * 8;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const float4 sv = *(const float4 *)(a.s2 + ni * 16 + fch * 4);
      const float4 tv = *(const float4 *)(a.t2 + ni * 16 + fch * 4);
      f16x4 hv;
      hv[0] = (f16)fmaxf(fmaf(acc[ni][mi][0], sv.x, tv.x), 0.f);
      hv[1] = (f16)fmaxf(fmaf(acc[ni][mi][1], sv.y, tv.y), 0.f);
      hv[2] = (f16)fmaxf(fmaf(acc[ni][mi][2], sv.z, tv.z), 0.f);
      hv[3] = (f16)fmaxf(fmaf(acc[ni][mi][3], sv.w, tv.w), 0.f);
      if (!in) hv = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
      const int chunk = ni * 2 + (fch >> 1);              // channels n>>3
      if (s < NSLOT) *(f16x4 *)(dst + ((chunk ^ (s & 15)) << 4)) = hv;
    }
  }
  if constexpr (NRING == 2) {
    *(f16x8 *)(ring + t * 16) = wq[0][0];   // tap 0 -> ring[0]
    *(f16x8 *)(ring + 4096 + t * 16) = wq[0][1];
    wq[0][0] = w3[3 * 512];                 // request tap 3
    wq[0][1] = w3[3 * 512 + 256];
  }
  __syncthreads();
  DL_STAMP(4);

  // ======================= phase B: y = conv3x3(tile) ========================================
  const int g = wid >> 1;            // fragment group
  const int hh = wid & 1;            // channel half: channels [64*hh, 64*hh+64)
  const int f0 = (g * NF) / G::NG, f1 = ((g + 1) * NF) / G::NG;   // this group's fragments [f0, f1)
  const int px = lane & 31, khalf = lane >> 5;
  f32x16 bacc[MAXF];
#pragma unroll
  for (int j = 0; j < MAXF; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) bacc[j][r] = 0.f;

  // NFR = fragments this wave really owns (wave-uniform); the loop body is branch-free so the
  // compiler can run the LDS reads ahead of the MFMAs
  auto phase_b = [&](auto nfr_tag) {
    constexpr int NFR = decltype(nfr_tag)::value;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // taps are fully unrolled, so the three weight registers rotate with static indices
      // (prefetch distance 3 taps)
      if constexpr (NRING == 2) {
        if (tap + 1 < 9) {
          *(f16x8 *)(ring + ((tap + 1) & 1) * 8192 + t * 16) = wq[(tap + 1) % 3][0];
          *(f16x8 *)(ring + ((tap + 1) & 1) * 8192 + 4096 + t * 16) = wq[(tap + 1) % 3][1];
          if (tap + 4 < 9) {
            wq[(tap + 1) % 3][0] = w3[(tap + 4) * 512];
            wq[(tap + 1) % 3][1] = w3[(tap + 4) * 512 + 256];
          }
        }
      } else {
        if (tap > 0) __syncthreads();      // everyone is done reading the previous tap's weights
        *(f16x8 *)(ring + t * 16) = wq[tap % 3][0];
        *(f16x8 *)(ring + 4096 + t * 16) = wq[tap % 3][1];
        if (tap + 3 < 9) {
          wq[tap % 3][0] = w3[(tap + 3) * 512];
          wq[tap % 3][1] = w3[(tap + 3) * 512 + 256];
        }
        __syncthreads();
      }
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int off = WP + dy * WP + dx + px + 32 * f0;   // slot of this lane's pixel, fragment f0
      const unsigned char *wring = ring + (NRING == 2 ? (tap & 1) * 8192 : 0) + (hh * 4) * 1024 + lane * 16;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f16x8 wf = *(const f16x8 *)(wring + kk * 1024);
        const int chunk = ((hh * 4 + kk) << 1) + khalf;
        f16x8 xf[NFR];
#pragma unroll
        for (int j = 0; j < NFR; ++j) {
          const int slot = off + 32 * j;
          xf[j] = *(const f16x8 *)(tile + slot * 256 + ((chunk ^ (slot & 15)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NFR; ++j)
          bacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf[j], bacc[j], 0, 0, 0);
      }
      if constexpr (NRING == 2) __syncthreads();
    }
    if constexpr (NRING != 2) __syncthreads();
  };
  if (f1 - f0 == MAXF) phase_b(std::integral_constant<int, MAXF>{});
  else phase_b(std::integral_constant<int, (MAXF > 1 ? MAXF - 1 : 1)>{});

  DL_STAMP(5);
  // ---- combine the two channel halves through LDS, store 32 channels per pixel ----
  if (hh == 1) {
#pragma unroll
    for (int j = 0; j < MAXF; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4 *)(red + ((g * MAXF + j) * 4 + q) * 1024 + lane * 16) =
            (f32x4){bacc[j][4 * q], bacc[j][4 * q + 1], bacc[j][4 * q + 2], bacc[j][4 * q + 3]};
  }
  __syncthreads();
  if (hh == 0) {
    f16 *ybase = (f16 *)ibase + K;
#pragma unroll
    for (int j = 0; j < MAXF; ++j) {
      if (f0 + j < f1) {
        const int s = WP + 32 * (f0 + j) + px;       // tile slot of this lane's output pixel
        const int tr = s / WP, xp = s - tr * WP;
        const bool ok = xp >= 1 && xp <= WT && tr >= 1 && tr <= ROUT;
        f16 *dst = ybase + ((long)(r0 + tr - 1) * WI + (x0 + xp - 1)) * ldc + 4 * khalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = *(const f32x4 *)(red + ((g * MAXF + j) * 4 + q) * 1024 + lane * 16);
          f16x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = (f16)(bacc[j][4 * q + r] + o[r]);
          if (ok) *(f16x4 *)(dst + 8 * q) = hv;
        }
      }
    }
  }
  DL_STAMP(6);
  if constexpr (CHAIN) {
    wait_vmcnt<0>();      // this layer's stores have completed ...
    __syncthreads();      // ... for every wave, before the next layer's loads of the same frame
  }
  }   // layer
}

template <int WT, int ROUT, int BM, int NST, int NRING, bool CHAIN = false>
int launch_geom(const DenseLayerArgs &a, hipStream_t s) {
  using G = DLGeom<WT, ROUT, BM, NST, NRING>;
  static bool attr_set = false;
  if (!attr_set) {
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_layer_kernel<WT, ROUT, BM, NST, NRING, CHAIN>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  const dim3 grid(a.B * (a.H / ROUT) * (a.W / WT)), block(NT);
  hipLaunchKernelGGL((dense_layer_kernel<WT, ROUT, BM, NST, NRING, CHAIN>), grid, block, G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

}  // namespace

bool dense_layer_small_supported(int H, int W) {
  return (H % 7 == 0 && W % 28 == 0 && W <= 56) || (H == 14 && W == 14) || (H == 7 && W == 7);
}

int launch_dense_layer_small(const DenseLayerArgs &a, hipStream_t s) {
  const int klast = a.K + 32 * (a.nchain > 0 ? a.nchain - 1 : 0);
  TN_REQUIRE(a.K % 32 == 0 && klast <= 1024 && a.ldc % 8 == 0 && klast + 32 <= a.ldc, "dense_layer: bad channel geometry");
  if (a.nchain > 0) {
    TN_REQUIRE(a.H == 7 && a.W == 7 && a.chain, "dense_layer: layer chaining needs whole-frame tiles (7x7)");
    return launch_geom<7, 7, 128, 3, 2, true>(a, s);
  }
  if (a.H == 14 && a.W == 14) return launch_geom<14, 14, 256, 3, 1>(a, s);
  if (a.H == 7 && a.W == 7) return launch_geom<7, 7, 128, 3, 2>(a, s);
  if (a.H % 7 == 0 && a.W % 28 == 0) return launch_geom<28, 7, 320, 2, 1>(a, s);
  TN_REQUIRE(false, "dense_layer: unsupported spatial size");
}

// Fused DenseNet dense layer (SURVEY §7 H1, §2c rows K2+K3+K4):
//
//   y[.., K:K+32] = conv3x3( relu(bn2( conv1x1( relu(bn1( x[.., 0:K] )) ) )) )
//
// one launch per layer, replacing gluoncv's BatchNorm-Activation-Conv1x1-BatchNorm-
// Activation-Conv3x3-Concat chain (reference call site models/vision/definitions.py:30).
// The 128-channel bottleneck never goes to HBM: a workgroup (8 waves) owns ROUT full
// image rows, computes the bottleneck of those rows plus one halo row above/below
// (phase A: MFMA GEMM over K, BN1+ReLU applied while staging global->regs->LDS,
// BN2+ReLU in the epilogue) into an LDS tile laid out as (ROUT+2) x (W+2) pixel
// slots of 256 B with zero padding materialised, so that the 3x3 (phase B) is a pure
// constant-offset walk over the flattened tile — no bounds logic in the hot loop.
// Phase B splits K=1152 by channel halves over wave pairs (partials meet in LDS), and
// streams the packed 3x3 weights through a 2 x 8 KiB LDS ring, one tap per barrier.
// The staging buffers of phase A alias the (not yet written) bottleneck tile.
// Writing the 32 output channels at channel offset K of the same buffer IS the concat.
#include <type_traits>

#include "common.h"

namespace {

template <int W, int ROUT, int MI>
struct DLGeom {
  static constexpr int WP = W + 2;
  static constexpr int TR = ROUT + 2;
  static constexpr int NSLOT = TR * WP;
  static constexpr int NF = (ROUT * WP + 31) / 32;           // 32-slot output fragments
  static constexpr int MAXF = (NF + 3) / 4;                  // fragments per wave group
  static constexpr int RSLOT = WP + 32 * NF + WP + 2;        // highest slot phase B touches + 1
  static constexpr int TSLOT = NSLOT > RSLOT ? NSLOT : RSLOT;
  static constexpr int TILE_BYTES = TSLOT * 256;
  static constexpr int BM = 64 * MI;
  static constexpr int STAGE_BYTES = BM * 128 + 128 * 128;
  static constexpr int RING_BYTES = 2 * 8192;
  static constexpr int RED_BYTES = 4 * MAXF * 4 * 1024;
  static constexpr int A0 = TILE_BYTES + RING_BYTES;
  static constexpr int A1 = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static constexpr int LDS_BYTES = A0 > A1 ? A0 : A1;
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
  static_assert(BM >= TR * W, "phase A tile too small");
};

template <int W, int ROUT, int MI>
__global__ __launch_bounds__(512) void dense_layer_kernel(DenseLayerArgs a) {
  using G = DLGeom<W, ROUT, MI>;
  constexpr int WP = G::WP, TR = G::TR, NF = G::NF, MAXF = G::MAXF, BM = G::BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Xs = smem;                     // phase A staging (aliases the tile)
  unsigned char *Ws = smem + BM * 128;
  unsigned char *tile = smem;                   // bottleneck tile
  unsigned char *ring = smem + G::TILE_BYTES;   // 3x3 weight ring
  unsigned char *red = smem;                    // phase B partial sums (aliases the tile)

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int H = a.H, K = a.K, ldc = a.ldc;
#define DL_STAMP(i) do { if (a.ts && t == 0) a.ts[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  DL_STAMP(0);
  constexpr int TPI = 0;  // (unused) tiles per image computed at run time
  (void)TPI;
  const int tiles_per_img = H / ROUT;
  const int img = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - img * tiles_per_img) * ROUT;     // first output row
  const int rlo = r0 > 0 ? r0 - 1 : 0;                           // first computed bottleneck row
  const int rhi = (r0 + ROUT < H) ? r0 + ROUT + 1 : H;           // one past the last
  const int MA = (rhi - rlo) * W;                                // real phase-A rows
  const int top_pad = (r0 == 0) ? 1 : 0;                         // tile row 0 lies above the image
  const f16 *xbase = a.buf + ((long)img * H * W + (long)rlo * W) * ldc;

  // ======================= phase A: bottleneck = conv1x1(relu(bn1(x))) =======================
  const int c = t & 7;       // 16-byte chunk column (8 channels), fixed per thread
  const int sr = t >> 3;     // 0..63
  const f16 *xsrc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = sr + 64 * i;
    if (m >= MA) m = MA - 1;
    xsrc[i] = xbase + (long)m * ldc;
  }
  const f16 *wsrc = a.w1 + (long)sr * K;

  f16x8 xr[MI], wr[2];
  float sc[8], sh[8];
  auto load_tile = [&](int kt) {
    const int kc = kt * 64 + c * 8;
    if (kc < K) {
      const float4 s0 = *(const float4 *)(a.s1 + kc), s1 = *(const float4 *)(a.s1 + kc + 4);
      const float4 t0 = *(const float4 *)(a.t1 + kc), t1 = *(const float4 *)(a.t1 + kc + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
#pragma unroll
      for (int i = 0; i < MI; ++i) xr[i] = *(const f16x8 *)(xsrc[i] + kc);
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[i] = *(const f16x8 *)(wsrc + (long)(64 * i) * K + kc);
    }
  };
  auto store_tile = [&](int kt) {
    const bool kv = kt * 64 + c * 8 < K;
    f16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (f16)0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      *(f16x8 *)(Xs + swz<128>(sr + 64 * i, c)) = kv ? bn_relu8(xr[i], sc, sh) : z;
#pragma unroll
    for (int i = 0; i < 2; ++i) *(f16x8 *)(Ws + swz<128>(sr + 64 * i, c)) = kv ? wr[i] : z;
  };

  const int wm = wid >> 1, wn = wid & 1;
  const int frow = lane & 15, fch = lane >> 4;
  f32x4 acc[4][MI];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + 63) / 64;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile(kt);
    __syncthreads();
    if (kt == 0) DL_STAMP(1);
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kt * 64 + ks * 32 < K) {
        f16x8 wa[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          wa[ni] = *(const f16x8 *)(Ws + swz<128>(wn * 64 + ni * 16 + frow, ks * 4 + fch));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const f16x8 xb = *(const f16x8 *)(Xs + swz<128>(wm * 16 * MI + mi * 16 + frow, ks * 4 + fch));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], xb, acc[ni][mi], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  DL_STAMP(2);
  // request the first two taps of the 3x3 weights now (one 16-B piece per thread per tap)
  const f16x8 *w3 = (const f16x8 *)a.w3p + t;
  f16x8 wq0 = w3[0], wq1 = w3[512];

  // BN2 scale/shift for this lane's 16 output channels (latency hides behind the zero fill)
  float s2[4][4], t2[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const float4 sv = *(const float4 *)(a.s2 + wn * 64 + ni * 16 + fch * 4);
    const float4 tv = *(const float4 *)(a.t2 + wn * 64 + ni * 16 + fch * 4);
    s2[ni][0] = sv.x; s2[ni][1] = sv.y; s2[ni][2] = sv.z; s2[ni][3] = sv.w;
    t2[ni][0] = tv.x; t2[ni][1] = tv.y; t2[ni][2] = tv.z; t2[ni][3] = tv.w;
  }
  // ---- zero padding of the tile: the two pad columns of every row, out-of-image halo rows ----
  {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    if (t < TR * 32) {
      const int tr = t >> 5, side = (t >> 4) & 1, ch = t & 15;
      *(uint4 *)(tile + (tr * WP + side * (WP - 1)) * 256 + ch * 16) = z4;
    }
    if (top_pad)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + idx * 16) = z4;
    if (r0 + ROUT >= H)
      for (int idx = t; idx < WP * 16; idx += 512) *(uint4 *)(tile + (TR - 1) * WP * 256 + idx * 16) = z4;
  }
  DL_STAMP(3);
  // ---- epilogue A: BN2 + ReLU, fp16, scatter into the tile ----
  {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = wm * 16 * MI + mi * 16 + frow;
      if (m < MA) {
        const int rr = m / W, x = m - rr * W;
        const int slot = (rr + top_pad) * WP + x + 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f16x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = (f16)fmaxf(fmaf(acc[ni][mi][r], s2[ni][r], t2[ni][r]), 0.f);
          const int n = wn * 64 + ni * 16 + fch * 4;           // 4 consecutive channels
          *(f16x4 *)(tile + slot * 256 + (((n >> 3) ^ (slot & 15)) << 4) + (n & 7) * 2) = hv;
        }
      }
    }
  }
  *(f16x8 *)(ring + t * 16) = wq0;   // tap 0 -> ring[0]
  wq0 = w3[2 * 512];                 // request tap 2
  __syncthreads();
  DL_STAMP(4);

  // ======================= phase B: y = conv3x3(tile) ========================================
  const int g = wid >> 1;            // fragment group 0..3
  const int hh = wid & 1;            // channel half: channels [64*hh, 64*hh+64)
  const int f0 = (g * NF) >> 2, f1 = ((g + 1) * NF) >> 2;   // this group's fragments [f0, f1)
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
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      // stage the next tap's weights while this tap computes
      if (tap + 1 < 9) *(f16x8 *)(ring + ((tap + 1) & 1) * 8192 + t * 16) = (tap & 1) ? wq0 : wq1;
      if (tap + 3 < 9) {
        if (tap & 1) wq0 = w3[(tap + 3) * 512]; else wq1 = w3[(tap + 3) * 512];
      }
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int off = WP + dy * WP + dx + px + 32 * f0;   // slot of this lane's pixel, fragment f0
      const unsigned char *wring = ring + (tap & 1) * 8192 + (hh * 4) * 1024 + lane * 16;
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
      __syncthreads();
    }
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
    f16 *ybase = a.buf + ((long)img * H * W) * ldc + K;
#pragma unroll
    for (int j = 0; j < MAXF; ++j) {
      if (f0 + j < f1) {
        const int s = WP + 32 * (f0 + j) + px;       // tile slot of this lane's output pixel
        const int tr = s / WP, xp = s - tr * WP;
        const bool ok = xp >= 1 && xp <= W && tr >= 1 && tr <= ROUT;
        f16 *dst = ybase + ((long)(r0 + tr - 1) * W + (xp - 1)) * ldc + 4 * khalf;
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
}

template <int W, int ROUT, int MI>
int launch_geom(const DenseLayerArgs &a, hipStream_t s) {
  using G = DLGeom<W, ROUT, MI>;
  static bool attr_set = false;
  if (!attr_set) {
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)dense_layer_kernel<W, ROUT, MI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set = true;
  }
  const dim3 grid(a.B * (a.H / ROUT)), block(512);
  hipLaunchKernelGGL((dense_layer_kernel<W, ROUT, MI>), grid, block, G::LDS_BYTES, s, a);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

}  // namespace

bool dense_layer_supported(int H, int W) { return (H == 56 && W == 56) || (H == 28 && W == 28) || (H == 14 && W == 14); }

int launch_dense_layer(const DenseLayerArgs &a, hipStream_t s) {
  TN_REQUIRE(a.K % 32 == 0 && a.ldc % 8 == 0 && a.K + 32 <= a.ldc, "dense_layer: bad channel geometry");
  if (a.H == 56 && a.W == 56) return launch_geom<56, 7, 8>(a, s);
  if (a.H == 28 && a.W == 28) return launch_geom<28, 14, 7>(a, s);
  if (a.H == 14 && a.W == 14) return launch_geom<14, 14, 4>(a, s);
  TN_REQUIRE(false, "dense_layer: unsupported spatial size");
}

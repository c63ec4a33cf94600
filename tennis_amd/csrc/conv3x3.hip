// K3 — fused BN+ReLU -> 3x3 convolution (128 -> 32 channels, pad 1) as an
// implicit GEMM on MFMA, writing its 32 channels in place at a channel offset
// of the dense block's concat buffer (that strided write IS the reference's
// HybridConcurrent concat).  Replaces the per-layer
// BatchNorm -> Activation -> Convolution(3x3) -> Concat chain of gluoncv's
// DenseNet (reference call site models/vision/definitions.py:30).
//
// One workgroup (4 waves) produces 128 consecutive pixels (NHWC linear order)
// x 32 channels.  The BN+ReLU'd input pixels [m0-W-1, m0+128+W+1) are staged
// once into LDS (fp16, 256 B per pixel, XOR-swizzled 16-B chunks); zero padding
// is decided per (output pixel, tap) and served from a zero slot, because the
// convolution pads AFTER the activation.  K = 9 taps x 128 ch = 72 k-steps of
// v_mfma_f32_32x32x16_f16; the k-steps are split over the 4 waves (18 each) so
// that each weight fragment (read straight from the pre-packed, L2-resident
// weight image) is reused by 4 pixel fragments; partial sums meet in LDS.
#include "common.h"

namespace {

constexpr int TILE_PX = 128;
constexpr int RED_BYTES = 4 * TILE_PX * 32 * 4;  // 4 waves x 128 px x 32 n fp32 = 64 KiB

// EX (exact-weights mode, round 6: the un-fused layers of map sizes no fused kernel tiles): w = hi + lo as two packed images,
// the lo image 2 x 72 x 64 fragments further on (api.hip packs both MFMA layouts of an image back to back, then the next
// image); every pixel fragment is multiplied with both.
template <int V, bool EX = false>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3x3Args a, int lds_px) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int W = a.W, H = a.H, M = a.M;
  const int m0 = blockIdx.x * TILE_PX;
  const int sbase = m0 - W - 1;  // linear pixel held by LDS slot 0 (may be negative)
  const int zero_off = lds_px * 256;  // 16 B of zeros live here

  // V>=1: all 18 weight fragments of this wave are requested before anything else, so
  // their L2 latency hides behind the staging phase instead of stalling every k-step
  const f16x8 *wp = (const f16x8 *)a.wp + (long)(wid * 18) * 64 + lane;
  f16x8 wpre[V >= 1 ? 18 : 1];
  if constexpr (V >= 1) {
#pragma unroll
    for (int i = 0; i < 18; ++i) wpre[i] = wp[(long)i * 64];
  }

  // ---- stage BN+ReLU'd input pixels -------------------------------------
  {
    const int ch = t & 15;  // 8-channel chunk, fixed per thread
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = a.scale[ch * 8 + j];
      sh[j] = a.shift[ch * 8 + j];
    }
    if constexpr (V >= 1) {
      // batches of 8 independent 16-B loads per thread, then transform + LDS write
      for (int p0 = t >> 4; p0 < lds_px; p0 += 128) {
        f16x8 raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          int gp = sbase + p0 + 16 * i;
          gp = gp < 0 ? 0 : (gp >= M ? M - 1 : gp);
          raw[i] = *(const f16x8 *)(a.x + (long)gp * 128 + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = p0 + 16 * i, gp = sbase + p;
          f16x8 v = bn_relu8(raw[i], sc, sh);
          if (gp < 0 || gp >= M) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
          }
          if (p < lds_px) *(f16x8 *)(smem + swz<256>(p, ch)) = v;
        }
      }
    } else {
      for (int p = t >> 4; p < lds_px; p += 16) {
        const int gp = sbase + p;
        f16x8 v;
        if (gp >= 0 && gp < M) {
          v = bn_relu8(*(const f16x8 *)(a.x + (long)gp * 128 + ch * 8), sc, sh);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
        }
        *(f16x8 *)(smem + swz<256>(p, ch)) = v;
      }
    }
    if (t == 0) *(uint4 *)(smem + zero_off) = make_uint4(0, 0, 0, 0);
  }

  // ---- per-lane pixel coordinates and tap validity ------------------------
  const int prow = lane & 31;
  const int khalf = lane >> 5;
  int pbase[4];   // LDS slot of this lane's pixel for tap (0,0), per M-fragment
  int vmask[4];   // bit tap = 1 if the tap lands inside the image
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int q = f * 32 + prow;
    int m = m0 + q;
    const bool inb = m < M;
    if (!inb) m = M - 1;
    const int x = m % W;
    const int y = (m / W) % H;
    pbase[f] = q + W + 1;
    int vm = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool ok = inb && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
      vm |= ok ? (1 << tap) : 0;
    }
    vmask[f] = vm;
  }
  __syncthreads();

  // ---- main loop: this wave's 18 k-steps over all 4 pixel fragments -------
  f32x16 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int s = wid * 18 + i;
    const int tap = s >> 3;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int chunk = ((s & 7) << 1) + khalf;
    const int doff = dy * W + dx;
    f16x8 wb;
    if constexpr (V >= 1) wb = wpre[i]; else wb = wp[(long)i * 64];
    [[maybe_unused]] f16x8 wl;
    if constexpr (EX) wl = wp[(long)(2 * 72 + i) * 64];
    f16x8 xa[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int slot = pbase[f] + doff;
      const int addr = ((vmask[f] >> tap) & 1) ? swz<256>(slot, chunk) : zero_off;
      xa[f] = *(const f16x8 *)(smem + addr);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[f], wb, acc[f], 0, 0, 0);
    if constexpr (EX) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[f], wl, acc[f], 0, 0, 0);
    }
  }
  __syncthreads();  // all waves done reading the staged pixels

  // ---- cross-wave reduction through LDS -----------------------------------
  // D[i=pixel][j=n]: lane holds n = lane&31, pixel rows (r&3)+8*(r>>2)+4*(lane>>5)
  float *red = (float *)smem;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      red[(wid * TILE_PX + q) * 32 + prow] = acc[f][r];
    }
  __syncthreads();
  {
    const int q = t >> 1;          // pixel within tile
    const int nh = (t & 1) * 16;   // channel half
    const int m = m0 + q;
    if (m < M) {
      float sum[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) sum[j] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 *src = (const float4 *)(red + (w * TILE_PX + q) * 32 + nh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = src[j];
          sum[4 * j + 0] += v.x; sum[4 * j + 1] += v.y; sum[4 * j + 2] += v.z; sum[4 * j + 3] += v.w;
        }
      }
      f16x8 o0, o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o0[j] = (f16)sum[j];
        o1[j] = (f16)sum[8 + j];
      }
      f16 *dst = a.y + (long)m * a.ldy + a.yoff + nh;
      *(f16x8 *)dst = o0;
      *(f16x8 *)(dst + 8) = o1;
    }
  }
}

}  // namespace

size_t conv3x3_lds_bytes(int W) {
  const size_t tile = (size_t)(TILE_PX + 2 * W + 2) * 256 + 16;
  return tile > (size_t)RED_BYTES ? tile : (size_t)RED_BYTES;
}

int launch_conv3x3(const Conv3x3Args &a, hipStream_t s) {
  TN_REQUIRE(a.ldy % 8 == 0 && a.yoff % 8 == 0, "conv3x3: strides must be multiples of 8");
  TN_REQUIRE(a.W <= 240, "conv3x3: W too large for the LDS tile");
  const int lds_px = TILE_PX + 2 * a.W + 2;
  const size_t lds = conv3x3_lds_bytes(a.W);
  TN_SET_ATTR_ONCE_PER_DEVICE(TN_HIP_CHECK(hipFuncSetAttribute((const void *)conv3x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)conv3x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    TN_HIP_CHECK(hipFuncSetAttribute((const void *)conv3x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024)));
  const dim3 grid((a.M + TILE_PX - 1) / TILE_PX), block(256);
  if (a.exact)
    hipLaunchKernelGGL((conv3x3_kernel<1, true>), grid, block, lds, s, a, lds_px);
  else if (a.variant == 9)
    hipLaunchKernelGGL(conv3x3_kernel<0>, grid, block, lds, s, a, lds_px);
  else
    hipLaunchKernelGGL(conv3x3_kernel<1>, grid, block, lds, s, a, lds_px);
  TN_HIP_CHECK(hipGetLastError());
  return TN_OK;
}

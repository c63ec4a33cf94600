// Shared device/host helpers for libtennis_hip (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/tennis_hip.h"
#include "../../include/tennis_hip_debug.h"

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing -------------------------------------------------------
void tn_set_error(const std::string &msg);
#define TN_HIP_CHECK(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      tn_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return TN_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)
#define TN_REQUIRE(cond, msg)                                                          \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      tn_set_error(std::string("invalid argument: ") + (msg));                         \
      return TN_ERR_INVALID;                                                           \
    }                                                                                  \
  } while (0)

// Makes `dev` the calling thread's current HIP device for the rest of the scope and puts the previous one back
// on exit: the library never leaves the caller's (torch's) current device changed behind its back.
struct TnDeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit TnDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev == dev) prev = -1;
    else ok = hipSetDevice(dev) == hipSuccess;
  }
  ~TnDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  TnDeviceGuard(const TnDeviceGuard &) = delete;
  TnDeviceGuard &operator=(const TnDeviceGuard &) = delete;
};
#define TN_DG_CAT2(a, b) a##b
#define TN_DG_CAT(a, b) TN_DG_CAT2(a, b)
#define TN_ON_DEVICE(dev)                                                              \
  TnDeviceGuard TN_DG_CAT(tn_dg_, __LINE__)(dev);                                      \
  if (!TN_DG_CAT(tn_dg_, __LINE__).ok) {                                               \
    tn_set_error("hipSetDevice failed");                                               \
    return TN_ERR_HIP;                                                                 \
  }

struct tn_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
};

// ---- device helpers -------------------------------------------------------
// BN(inference)+ReLU on 8 fp16 values with fp32 scale/shift, rounded once to fp16.
__device__ __forceinline__ f16x8 bn_relu8(f16x8 v, const float *__restrict__ s, const float *__restrict__ t) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = fmaf((float)v[j], s[j], t[j]);
    r[j] = (f16)fmaxf(f, 0.0f);
  }
  return r;
}

// The dense layers' BN1 + ReLU in its rounding-free form (calib_host.hip::bn_relu_clamp_fold): the operand is clamp(x, lo, hi),
// lo / hi fp16 numbers handed over as floats.
__device__ __forceinline__ f16x8 clamp8(f16x8 v, const float *__restrict__ lo, const float *__restrict__ hi) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (f16)fminf(fmaxf((float)v[j], lo[j]), hi[j]);
  return r;
}
// two channels (a packed dword): the constants as floats -> packed halves (exact: they are fp16 numbers), v_pk_max_f16 + v_pk_min_f16.
// The result usually goes straight into an MFMA as its B operand, and hipcc pads ONE wait state behind an asm statement where a
// VALU-written MFMA operand needs two (cdna_hip_programming.md 5.7 item 2; seen in round 5 as one stale dword of the first
// fragment on the waves whose schedule put the MFMA right behind the statement): the statement carries the other one itself.
__device__ __forceinline__ unsigned clamp2_pk(unsigned in, float lo0, float lo1, float hi0, float hi1) {
  unsigned l, h, o;
  asm("v_cvt_pk_f16_f32 %1, %4, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_pk_max_f16 %0, %3, %1\n\tv_pk_min_f16 %0, %0, %2\n\ts_nop 0"
      : "=&v"(o), "=&v"(l), "=&v"(h) : "v"(in), "v"(lo0), "v"(lo1), "v"(hi0), "v"(hi1));
  return o;
}
__device__ __forceinline__ f16x8 clamp8_pk(f16x8 v, const float *__restrict__ lo, const float *__restrict__ hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 in = __builtin_bit_cast(u32x4, v);
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = clamp2_pk(in[j], lo[2 * j], lo[2 * j + 1], hi[2 * j], hi[2 * j + 1]);
  return __builtin_bit_cast(f16x8, out);
}

// relu(acc*s+t) for 4 fp32 accumulators -> 4 fp16 (fp32 fma, one rounding, packed ReLU): 4 v_fma_f32 + 2 v_cvt_pk_f16_f32 +
// 2 v_pk_max_f16 = 12.9 ns against 18.3 ns for 4 v_fma_mixlo/hi_f16 + 2 v_pk_max_f16 (inline asm: left to itself the
// compiler fuses fma + conversion into the slow mixlo / mixhi forms, or SLP-packs the fmas and pays for it in moves).
// `v` is an MFMA result, and hipcc pads no MFMA -> VALU hazard in front of an asm statement (it does not know what the
// statement reads; garbage was seen in round 2 when the scheduler happened to put the last MFMA right in front of it): the
// statement therefore OPENS with the wait states an 8-pass MFMA result needs (12, cdna_hip_programming.md 5.7 item 2) - one
// statement, so nothing can be scheduled between the wait and the reads.  Correctness no longer depends on where the
// compiler places the MFMAs.
__device__ __forceinline__ f16x4 bn_relu4_from_f32(f32x4 v, float4 s, float4 t) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  float a0, a1, a2, a3;
  unsigned d0, d1;
  asm("s_nop 11\n\t"
      "v_fma_f32 %2, %6, %10, %14\n\tv_fma_f32 %3, %7, %11, %15\n\tv_fma_f32 %4, %8, %12, %16\n\tv_fma_f32 %5, %9, %13, %17\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\tv_pk_max_f16 %0, %0, 0\n\tv_pk_max_f16 %1, %1, 0"
      : "=&v"(d0), "=&v"(d1), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(s.x), "v"(s.y), "v"(s.z), "v"(s.w), "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
  const u32x2 o = {d0, d1};
  return __builtin_bit_cast(f16x4, o);
}

// Byte offset of 16-byte chunk `chunk` of row `row` in an LDS tile whose rows are
// ROWB bytes (ROWB/16 chunks, power of two), XOR-swizzled so that a ds_read_b128
// lane group reading 16 different rows at the same chunk is conflict-free.
template <int ROWB>
__device__ __forceinline__ int swz(int row, int chunk) {
  constexpr int NCH = ROWB / 16;
  return row * ROWB + ((chunk ^ (row & (NCH - 1))) << 4);
}

// ---- launchers (defined in the kernel .hip files) --------------------------
struct Conv1x1Args {
  const f16 *x;       // [rows_in][ldx] NHWC activations (concat buffer)
  int ldx;            // channel stride of x
  int K;              // input channels used (multiple of 32)
  const float *scale; // [K] folded BN scale
  const float *shift; // [K]
  const f16 *w;       // [N][K] fp16, K contiguous
  int N;              // output channels (multiple of 128)
  f16 *y;             // [M][ldy], written at column yoff
  int ldy, yoff;
  int M;              // output pixels (B*Ho*Wo)
  int pool;           // 0: Ho=H ; 1: 2x2 average of BN+ReLU'd input before the GEMM
  int H, W;           // input spatial size (used when pool)
  int variant = 0;    // tuning hook: 0 = default kernel choice
  int exact = 0;      // weights as hi + lo fp16 pairs: w [N][2 Kp] = [hi | lo], Kp = K rounded up to 64 (pool: K % 64 == 0)
  const float *bias = nullptr;   // [N] added to the fp32 result before the rounding to fp16 (no pooling)
  int clamp = 0;      // the dense layers' BN1 form: scale / shift hold lo / hi, the operand is clamp(x, lo, hi) (no arithmetic, no rounding)
  float *y32 = nullptr;   // [M][ld32] fp32: the result once more, un-rounded (the last transition: what the head reads), columns [0, N)
  int ld32 = 0;
  const f16 *wfrag = nullptr;   // w once more in MFMA operand order (pack_trans_frags): enables the warp-specialised kernel of trans_ws.hip
};
int launch_conv1x1(const Conv1x1Args &a, hipStream_t s);
// trans_ws.hip: the last transition (N = 512, one frame of <= 64 pooled pixels per workgroup)
bool trans_ws_supported(const Conv1x1Args &a);
int launch_trans_ws(const Conv1x1Args &a, hipStream_t s);
std::vector<f16> pack_trans_frags(const f16 *w, int N, int K);
int launch_pack_trans_frags(const f16 *w, int N, int K, f16 *out, hipStream_t s);

struct Conv3x3Args {
  const f16 *x;       // [M][128] bottleneck, dense
  const float *scale; // [128]
  const float *shift; // [128]
  const f16 *wp;      // packed B fragments [72][64 lanes][8]
  f16 *y;             // [M][ldy] at column yoff, 32 channels
  int ldy, yoff;
  int M, H, W;        // M = B*H*W
  int variant = 0;    // tuning hook: 0 = default kernel choice
  int exact = 0;      // weights as hi + lo: wp holds the hi image (both MFMA layouts, pack_conv3x3) and then the lo image
};
int launch_conv3x3(const Conv3x3Args &a, hipStream_t s);
size_t conv3x3_lds_bytes(int W);

struct DenseLayerDev {   // one layer's parameters as the chained kernels read them from device memory
  const float *s1, *t1;
  const f16 *w1;
  const float *s2, *t2;
  const f16 *w3p;
};
struct DenseLayerArgs {
  f16 *buf;            // concat buffer [B][H][W][ldc]: reads channels [0,K), writes [K,K+32)
  int ldc, K;
  const float *s1, *t1;  // [K]   folded BN1
  const f16 *w1;         // [128][K]
  const float *s2, *t2;  // [128] folded BN2
  const f16 *w3p;        // packed 3x3 fragments [72][64][8]
  int B, H, W;
  unsigned long long *ts = nullptr;  // tuning hook: 8 s_memtime stamps per workgroup
  int variant = 0;                   // tuning hook of the K loop: bit 3 = refill up front
  const DenseLayerDev *chain = nullptr;  // device array: run nchain consecutive layers (K, K+32, ...) in one launch
  int nchain = 0;                        // (whole-frame tiles only: 14x14 and 7x7)
  int exact = 0;                         // weights as hi + lo fp16 pairs: w1 [128][2 Kp] = [hi | lo], w3p = hi image then lo image
};
bool dense_layer_supported(int H, int W);
int dense_layer_kmax(int W);       // most input channels a fused layer of that (supported) map width takes
int launch_dense_layer(const DenseLayerArgs &a, hipStream_t s);

// The strip-streaming fused dense layer (dense_strip_impl.h): one workgroup per frame, weights resident in LDS.
struct DenseStripArgs {
  f16 *buf;              // concat buffer [B][H][W][ldc]: reads channels [0,K), writes [K,K+32)
  int ldc, K;
  const float *s1, *t1;  // [K]   folded BN1
  const f16 *w1s;        // 1x1 weights (BN2 scale folded in) + BN2 shift as A fragments (pack_w1_strip)
  const f16 *w3s;        // 3x3 weights as A fragments (pack_w3_strip)
  int B, H, W;
  unsigned long long *ts = nullptr;   // tuning hook: s_memtime stamps of wave 0 of every workgroup (128 per workgroup)
};
bool dense_strip_supported(int H, int W, int K);
int launch_dense_strip(const DenseStripArgs &a, hipStream_t s);
std::vector<f16> pack_w1_strip(const float *w /*[128][K], BN2 scale folded in*/, int K, const float *shift /*[128] BN2 shift*/);
std::vector<f16> pack_w3_strip(const float *w /*(32,128,3,3)*/);

// The 7x7 dense block with the frame's concat buffer resident in LDS (dense_block7.hip): one launch, one workgroup per frame.
struct DenseBlock7Args {
  f16 *buf;              // concat buffer [B][49][ldc]: reads channels [0,K0), appends [K0, K0 + 32 nl)
  int ldc, K0, nl, B;
  const f16 *wa, *wb;    // per-wave weight streams (pack_block7)
  const float *tab;      // per layer s1[1024] | t1[1024] | t2[128]
  unsigned a_off[4], b_off[4];   // start of each wave's streams, in 16-byte units
  unsigned long long *ts = nullptr;
  float *side = nullptr;         // [B][49][ldc] fp32: the appended channels once more, un-rounded, for the head
};
struct Block7Layer { const float *w1f /*[128][K], BN2 scale folded in*/, *w3 /*(32,128,3,3)*/, *s1, *t1 /*[K]*/, *t2 /*[128]*/; };
struct Block7Image {
  std::vector<f16> wa, wb;
  std::vector<float> tab;
  unsigned a_off[4], b_off[4];
};
bool dense_block7_supported(int H, int W, int K0, int nl);
Block7Image pack_block7(const std::vector<Block7Layer> &layers, int K0);
int launch_dense_block7(const DenseBlock7Args &a, hipStream_t s);

// The 14x14 dense block as one launch of pixel-owning waves with all weights streamed through an LDS ring (dense_block14.hip).
struct DenseBlock14Args {
  f16 *buf;                      // concat buffer [B][196][ldc]: reads channels [0,K0), appends [K0, K0 + 32 nl)
  int ldc, K0, nl, B;
  const unsigned char *stream;   // the block's weight stream (pack_block14)
  int total_units;               // dense_block14_units(K0, nl)
  f16 *scratch = nullptr;        // B x dense_block14_scratch_halfs(): the kernel's k-step-major working copy of the frames
  unsigned long long *ts = nullptr;   // tuning hook: s_memtime per layer (64 per workgroup)
};
struct Block14Layer { const float *w1f /*[128][K], BN2 scale folded in*/, *w3 /*(32,128,3,3)*/, *s1, *t1 /*[K]*/, *t2 /*[128]*/; };
bool dense_block14_supported(int H, int W, int K0, int nl);
int dense_block14_units(int K0, int nl);
size_t dense_block14_scratch_halfs();   // per frame
std::vector<unsigned char> pack_block14(const std::vector<Block14Layer> &layers, int K0);
int launch_dense_block14(const DenseBlock14Args &a, hipStream_t s);

// The 28x28 dense block as one launch: dense_block14.hip's recipe with the frame walked in four passes of eight rows (dense_block28.hip).
struct DenseBlock28Args {
  f16 *buf;                      // concat buffer [B][784][ldc]: reads channels [0,K0), appends [K0, K0 + 32 nl)
  int ldc, K0, nl, B;
  const unsigned char *stream;   // the block's weight stream (pack_block28)
  int total_units;               // dense_block28_units(K0, nl)
  f16 *scratch = nullptr;        // B x dense_block28_scratch_halfs(): the kernel's k-step-major working copy of the frames (zeroed once)
  unsigned long long *ts = nullptr;   // tuning hook: s_memtime per layer (64 per workgroup)
};
bool dense_block28_supported(int H, int W, int K0, int nl);
int dense_block28_units(int K0, int nl);
size_t dense_block28_scratch_halfs();   // per frame
std::vector<unsigned char> pack_block28(const std::vector<Block14Layer> &layers, int K0);
int launch_dense_block28(const DenseBlock28Args &a, hipStream_t s);

// ---- the stem's operand (round 5) -------------------------------------------------------------------------------
// The reference hands the network ToTensor + Normalize of a decoded frame, v = (x / 255 - mean_c) / std_c (evaluate.py:96-97).
// Rounding v to fp16 as the MFMA operand was the largest single error on flat frames (scripts/round_study.py: 8e-4 .. 1.3e-3 on
// the features, the same error in every pixel of a flat region).  The stem therefore works on (x - 255 mean_c): the factor
// 1 / (255 std_c) is folded into the conv0 weights BEFORE they are rounded (weights.as_fp16_model defines the fp16 model that
// way; kStemWScale keeps the small weights out of fp16's subnormals and comes out again through the BatchNorm scale).  For
// uint8 frames the staged operand is the INTEGER x - q_c (q_c = 255 mean_c rounded: exact in fp16, two packed-half
// instructions per two values), out-of-frame taps are staged as 255 mean_c - q_c (= the normalised zero), and the constant
// sum w (255 mean_c - q_c) sits in the BatchNorm shift (StemArgs::shift_u8).  Normalised inputs (fp32 NCHW / fp16 NHWC) are
// staged as v * 255 std_c, rounded once, padding 0.
constexpr double kStemMean[3] = {0.485, 0.456, 0.406}, kStemStd[3] = {0.229, 0.224, 0.225};
constexpr float kStemQ[3] = {124.f, 116.f, 104.f};
constexpr double kStemWScale = 64.0;
constexpr float stem_unscale(int c) { return (float)(255.0 * kStemStd[c]); }                     // v -> x - 255 mean_c
constexpr float stem_wfactor(int c) { return (float)(kStemWScale / (255.0 * kStemStd[c])); }     // conv0 weight of input channel c
constexpr double stem_pad(int c) { return 255.0 * kStemMean[c] - (double)kStemQ[c]; }            // staged value of an out-of-frame tap (u8)

struct StemArgs {
  const void *x;
  int layout;         // tn_layout
  int B, H, W;        // input size
  const f16 *wp;      // packed A fragments [7 ky][4 nfrag][64 lanes][8]
  const f16 *wp_zf;   // the same with the zero x-tap FIRST (k slot = (kx + 1) * 4 + c): fused stem + maxpool kernel
  const float *scale; // [64] folded BN scale
  const float *shift; // [64]
  f16 *y;             // [B][Ho][Wo][64]
  int Ho, Wo;
  const float *shift_u8 = nullptr;   // the shift for TN_LAYOUT_NHWC_U8 input (carries the constant of the integer staging, see above)
  const f16 *wp_zf_lo = nullptr;     // exact-weights mode: the lo halves of the weights (w = hi + lo), same fragment layout as wp_zf
  // Round 6: the pooled map is stored CENTRED, relu(bn(conv)) - m_c, with m_c the channel's mean as the consuming BatchNorms
  // know it (api.hip "centred stem output").  The kernels get `shift` / `shift_u8` with m_c already subtracted and the ReLU's
  // floor -m_c here (max pool and ReLU commute with the subtraction: max(v - m, -m) = relu(v) - m); nullptr: floor 0.
  const float *floor = nullptr;      // [64]
};
int launch_stem(const StemArgs &a, hipStream_t s);
// fused stem + maxpool: writes the pooled map (Hp x Wp x 64) at row stride ldy
int launch_stem_pool(const StemArgs &a, f16 *out, int ldy, int Hp, int Wp, hipStream_t s);

int launch_maxpool3x3s2(const f16 *x, int B, int H, int W, int C, f16 *y, int ldy, int Ho, int Wo, hipStream_t s);
// hipFuncAttributeMaxDynamicSharedMemorySize is set per DEVICE: a process-wide `static bool` left the second GPU of a process
// with the default limit.  One bit per device; the attribute call is idempotent, so two threads racing only repeat it.
struct PerDeviceFlag {
  std::atomic<unsigned long long> done{0};
  bool is_set(int *dev_out) {
    int d = 0;
    (void)hipGetDevice(&d);
    *dev_out = d;
    return (done.load(std::memory_order_acquire) >> (d & 63)) & 1ull;
  }
  void set(int d) { done.fetch_or(1ull << (d & 63), std::memory_order_release); }
};
#define TN_SET_ATTR_ONCE_PER_DEVICE(...)                    \
  do {                                                      \
    static PerDeviceFlag tn_flag_;                          \
    int tn_dev_;                                            \
    if (!tn_flag_.is_set(&tn_dev_)) {                       \
      __VA_ARGS__;                                          \
      tn_flag_.set(tn_dev_);                                \
    }                                                       \
  } while (0)

// BN1 + ReLU of the dense layers without a rounding: relu(s x + t) = sw clamp(x, lo, hi) + tc, lo / hi fp16 numbers, sw folded into
// the 1x1 weights, sum_k w[n][k] tc[k] into BN2's shift (csrc/calib_host.hip)
void bn_scale_shift(const float *gamma, const float *beta, const float *mean, const float *var, int n, float eps, float *scale, float *shift);
void bn_relu_clamp_fold(const float *scale, const float *shift, int n, float *lo, float *hi, float *sw, float *tc);
int launch_channel_mean(const f16 *x, int ld, int K, const float *scale, const float *shift, long rows, double *scratch /* 32 * K doubles */,
                        float *out, hipStream_t s, int clamp = 0 /* scale / shift are lo / hi of the clamp form */);
int launch_head(const f16 *x, int B, int H, int W, int C, const float *scale, const float *shift,
                float *feat, int PH, int PW, hipStream_t s, const float *x32 = nullptr /* the same map un-rounded: read instead of x */);

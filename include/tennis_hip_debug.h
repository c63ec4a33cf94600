/* libtennis_hip.so - instrumentation and test hooks (NOT part of the drop-in surface of include/tennis_hip.h).
 *
 * A maintainer binding the reference's path needs tennis_hip.h only.  What is declared here is exported by the same library for
 * this repo's own tests/ (single kernels pinned against the oracle at ragged sizes), bench.py (per-kernel HIP-event timing for
 * the roofline object) and scripts/ (tuning): no reference call site stands behind any of it. */
#ifndef TENNIS_HIP_DEBUG_H
#define TENNIS_HIP_DEBUG_H
#include "tennis_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel-family timing returned by tn_densenet121_profile. */
typedef struct {
  char name[32];      /* kernel family, e.g. "conv3x3_bnrelu" */
  int launches;       /* launches of this family in one forward */
  double ms;          /* summed HIP-event time of those launches */
  double flops;       /* algorithmic FLOPs of those launches (2*MACs of the convs) */
  double bytes;       /* algorithmic bytes moved (activations+weights, once each) */
} tn_kernel_stat;


/* Same forward, but every launch is bracketed by HIP events on the ctx stream;
 * fills up to max_stats families and syncs.  For bench.py's roofline. */
int tn_densenet121_profile(tn_encoder *enc, const void *x, tn_layout layout, int batch, float *feat,
                           tn_kernel_stat *stats, int max_stats, int *n_stats);

/* Test hook: copy an internal NHWC fp16 activation of the LAST forward to a host
 * fp32 buffer.  tap in {"stem","pool0","stage1".."stage4","trans1".."trans3",
 * "stage<k>_l0_bottleneck"}; returns the element count through *numel. */
int tn_densenet121_read_tap(tn_encoder *enc, const char *tap, int batch, float *out_host,
                            size_t capacity, size_t *numel);

int tn_jpeg_sync_passes(const tn_jpeg *j);   /* diagnostic: decoder passes the last call needed to synchronise */

/* ---- test hooks (used by tests/ only) -------------------------------------- */
/* Run ONE encoder kernel on caller-provided device activations (fp16 NHWC) with
 * host fp32 weights in Gluon layout, folded/packed exactly as
 * tn_densenet121_create does; synchronous.  They let the parity tests pin each
 * kernel against the oracle at ragged sizes and channel offsets. */
int tn_dbg_conv1x1(tn_ctx *ctx, const void *x_f16, int ldx, int K, const float *scale_host,
                   const float *shift_host, const float *w_host, int N, void *y_f16, int ldy, int yoff,
                   int M, int pool, int H, int W);
int tn_dbg_conv3x3(tn_ctx *ctx, const void *x_f16, const float *scale_host, const float *shift_host,
                   const float *w_host, void *y_f16, int ldy, int yoff, int B, int H, int W);
/* Tuning hooks: asynchronous single launches on device-resident, pre-converted
 * operands (fp16 [N][K] 1x1 weights; tn_dbg_pack_conv3x3 image for the 3x3:
 * 2 x 72*64*8 halves, the 32x32x16 and the 16x16x32 MFMA operand layouts). */
int tn_dbg_pack_conv3x3(const float *w_host, uint16_t *out_host);
int tn_dbg_conv1x1_dev(tn_ctx *ctx, const void *x_f16, int ldx, int K, const float *scale, const float *shift,
                       const void *w_f16, int N, void *y_f16, int ldy, int yoff, int M, int pool, int H, int W,
                       int variant);
int tn_dbg_conv3x3_dev(tn_ctx *ctx, const void *x_f16, const float *scale, const float *shift,
                       const void *wp_f16, void *y_f16, int ldy, int yoff, int B, int H, int W, int variant);
int tn_dbg_dense_layer_dev(tn_ctx *ctx, void *buf_f16, int ldc, int K, const float *s1, const float *t1,
                           const void *w1_f16, const float *s2, const float *t2, const void *w3p_f16, int B,
                           int H, int W, unsigned long long *ts /* NULL or stamps */, int variant /* 0 auto, 1 big, 2 small */);
int tn_dbg_linear(tn_ctx *ctx, const float *x, const float *w, const float *bias, float *y, int M, int N,
                  int K);
/* The strip-streaming fused dense layer (csrc/dense_strip.hip; 56x56 / 28x28 blocks, K <= 320): fp32 (128,K) 1x1 weights
 * with the folded scale / shift (128 each) of the BatchNorm behind them, and (32,128,3,3) 3x3 weights -> the MFMA
 * A-fragment images the kernel keeps resident in LDS ((K+16)*128 and 36864 halves; either output may be NULL), and one
 * asynchronous launch on device-resident packed operands. */
int tn_dbg_pack_strip(const float *w1_host, int K, const float *s2_host, const float *t2_host, uint16_t *w1s_out,
                      const float *w3_host, uint16_t *w3s_out);
int tn_dbg_dense_strip_dev(tn_ctx *ctx, void *buf_f16, int ldc, int K, const float *s1, const float *t1,
                           const void *w1s_f16, const void *w3s_f16, int B, int H, int W,
                           unsigned long long *ts /* NULL or 128 s_memtime stamps per frame */);

/* The LDS-resident 7x7 dense block (csrc/dense_block7.hip): nl layers from K0 input channels in ONE launch on a device
 * concat buffer (B,7,7,ldc) fp16.  w1_all: the (128, K_l) 1x1 weights one after the other (K_l = K0 + 32 l); s1_all / t1_all
 * folded BN1 scale / shift (K_l each); s2_all / t2_all folded BN2 scale / shift (128 per layer); w3_all nl x (32,128,3,3). */
int tn_dbg_block7_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                         const float *s2_all, const float *t2_all, const float *w3_all, void **out);
int tn_dbg_block7_run(void *handle, void *buf_f16, int ldc, int B);
int tn_dbg_block7_run_ts(void *handle, void *buf_f16, int ldc, int B,
                         unsigned long long *ts /* NULL or 128 per frame: s_memtime stamps of wave 0 (start, then 5 per layer) */);
void tn_dbg_block7_destroy(void *handle);

/* The streamed 14x14 dense block (csrc/dense_block14.hip; reference call site models/vision/definitions.py:30, the third dense
 * block of gluoncv's DenseNet-121): nl layers from K0 >= 256 input channels in ONE launch on a device concat buffer (B,14,14,ldc)
 * fp16; operands as tn_dbg_block7_create.  The handle owns the packed weight stream and the kernel's working copy of the frames. */
int tn_dbg_block14_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                          const float *s2_all, const float *t2_all, const float *w3_all, void **out);
int tn_dbg_block14_run(void *handle, void *buf_f16, int ldc, int B);
int tn_dbg_block14_run_ts(void *handle, void *buf_f16, int ldc, int B,
                          unsigned long long *ts /* NULL or 64 per frame (160 in a -DTN_B14_STAMPS build): s_memtime stamps of wave 0, one per layer */);
void tn_dbg_block14_destroy(void *handle);
/* the streamed 28x28 dense block (csrc/dense_block28.hip), same operand convention */
int tn_dbg_block28_create(tn_ctx *ctx, int K0, int nl, const float *w1_all, const float *s1_all, const float *t1_all,
                          const float *s2_all, const float *t2_all, const float *w3_all, void **out);
int tn_dbg_block28_run(void *handle, void *buf_f16, int ldc, int B);
int tn_dbg_block28_run_ts(void *handle, void *buf_f16, int ldc, int B, unsigned long long *ts);
void tn_dbg_block28_destroy(void *handle);

#ifdef __cplusplus
}
#endif
#endif /* TENNIS_HIP_DEBUG_H */

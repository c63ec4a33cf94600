// Launchers of the temporal-head training kernels (train.hip): bi-GRU forward with saved gates, max-over-time with
// argmax, softmax cross-entropy, the backward passes and the SGD update.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int launch_pool_max_arg(const float *x, int B, int T, int F, float *y, int32_t *arg, hipStream_t s);
int launch_softmax_ce(const float *logits, const int32_t *labels, int B, int C, float *loss, float *dlogits,
                      hipStream_t s);
int launch_dense_bwd(const float *dlogits, const float *pooled, const float *wd, int B, int C, int K, float *dwd,
                     float *dbd, float *dpooled, hipStream_t s);
int launch_scatter_pool_grad(const float *dpooled, const int32_t *arg, int B, int T, int F, float *dseq, hipStream_t s);
int launch_gru_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                         float *dgh, float *hprev, int B, int T, int H, hipStream_t s, int dirs = 2,
                         const int32_t *valid_len = nullptr, const float *dh_last = nullptr);
int launch_lstm_train_bwd(const float *seq, const float *gates, const float *dseq, const float *wh, float *dgi,
                          float *hprev, int B, int T, int H, hipStream_t s, int dirs = 2,
                          const int32_t *valid_len = nullptr, const float *dh_last = nullptr, const float *dc_last = nullptr);
int launch_gemm_tn_f32(const float *A, int lda, const float *Bm, int ldb, float *Cm, int ldc, int M, int N, int K,
                       hipStream_t s, float *workspace = nullptr, long workspace_floats = 0);   // workspace: enables split-K
// ... with B -> relu(B * bsc[n] + bsh[n]) applied while the operand is staged
int launch_gemm_tn_f32_bnrelu(const float *A, int lda, const float *Bm, int ldb, const float *bsc, const float *bsh, float *Cm, int ldc,
                              int M, int N, int K, hipStream_t s, float *workspace = nullptr, long workspace_floats = 0);
int launch_colsum_f32(const float *A, int lda, int rows, int cols, float *out, hipStream_t s);
int launch_sgd_momentum(float *w, const float *g, float *mom, long n, float lr, float momentum, float wd,
                        float rescale, hipStream_t s);
int launch_transpose_f32(const float *src, int rows, int cols, float *dst, hipStream_t s);

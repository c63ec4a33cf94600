// Launchers of the recurrent / pooling / metric kernels (rnn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int launch_rnn_recurrent(int gates, const float *gi, int ldgi, const float *whT, const float *bh,
                         const int32_t *valid_len, float *seq, int ldo, float *h_last, float *c_last, int B, int T,
                         int H, int dirs, hipStream_t s, float *save = nullptr);
// save (training, train.hip's BPTT): per (dir, row) the gate activations of every step, [dirs][B*T][(G+1)*H] =
// GRU r | z | n | (W_hn h + b_hn), LSTM i | f | g | o | c_t
int launch_temporal_pool(const float *x, int B, int T, int F, int kind, float *y, hipStream_t s);
int launch_prf1(const float *logits, const int32_t *labels, int rows, int classes, int64_t *mat, hipStream_t s);

// fp32 linear launcher shared by dense / rnn / gnmt translation units.
#pragma once
#include <hip/hip_runtime.h>
int launch_linear_f32(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy,
                      int M, int N, int K, int accumulate, hipStream_t s);

// fp32 linear launcher shared by dense / rnn / gnmt translation units.
#pragma once
#include <hip/hip_runtime.h>
int launch_linear_f32(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy,
                      int M, int N, int K, int accumulate, hipStream_t s);
// Latency-optimised form for skinny problems (few rows, K >= 128, float4-aligned operands; falls back to launch_linear_f32
// otherwise): 16 x 16 output tiles, K split over the four waves of a workgroup, every operand requested up front.
int launch_linear_f32_lat(const float *X, int ldx, const float *Wt, int ldw, const float *bias, float *Y, int ldy, int M, int N, int K,
                          hipStream_t s);
// Y = relu(X * asc[k] + ash[k]) W^T (+ bias): a BatchNorm + ReLU in front of the GEMM applied to the X operand while it is staged
int launch_linear_f32_bnrelu(const float *X, int ldx, const float *asc, const float *ash, const float *Wt, int ldw, const float *bias,
                             float *Y, int ldy, int M, int N, int K, int accumulate, hipStream_t s);

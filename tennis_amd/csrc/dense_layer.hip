// Fused DenseNet dense layer of the tile kernels (dense_layer_big.hip: 8 waves, ROUT full image rows, one workgroup
// per CU; whole-frame layer chains at 14x14 and 7x7).  The 56x56 / 28x28 layers with K <= 320 go to the strip kernel (dense_strip_impl.h) at
// batch >= 64 (api.hip).  Round 1's 4-wave geometry (28x7 tiles, two workgroups per CU) lost at every block size and
// is gone; spatial sizes none of these kernels tile run un-fused (conv1x1.hip + conv3x3.hip).
#include "common.h"

bool dense_layer_big_supported(int H, int W);
int dense_layer_big_kmax(int W);
int launch_dense_layer_big(const DenseLayerArgs &a, hipStream_t s);

bool dense_layer_supported(int H, int W) { return dense_layer_big_supported(H, W); }
int dense_layer_kmax(int W) { return dense_layer_big_kmax(W); }

int launch_dense_layer(const DenseLayerArgs &a, hipStream_t s) {
  TN_REQUIRE(dense_layer_big_supported(a.H, a.W), "dense_layer: unsupported spatial size");
  return launch_dense_layer_big(a, s);
}

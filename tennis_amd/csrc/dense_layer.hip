// Fused DenseNet dense layer: dispatch between the two workgroup geometries
// (dense_layer_big.hip: 8 waves, ROUT full image rows, one workgroup per CU;
//  dense_layer_small.hip: 4 waves, 28x7 / whole-frame tiles, >= 2 workgroups per CU).
// The choice per block is empirical (scripts/kbench.py on MI355X, batch 256).
#include "common.h"

bool dense_layer_big_supported(int H, int W);
bool dense_layer_small_supported(int H, int W);
int launch_dense_layer_big(const DenseLayerArgs &a, hipStream_t s);
int launch_dense_layer_small(const DenseLayerArgs &a, hipStream_t s);

bool dense_layer_supported(int H, int W) { return dense_layer_big_supported(H, W) || dense_layer_small_supported(H, W); }

int launch_dense_layer(const DenseLayerArgs &a, hipStream_t s) {
  if (a.nchain > 0) return launch_dense_layer_big(a, s);   // whole-frame chains (14x14, 7x7)
  const bool big_ok = dense_layer_big_supported(a.H, a.W), small_ok = dense_layer_small_supported(a.H, a.W);
  bool use_big = big_ok;                       // measured: the 8-wave geometry wins at every block size
  if ((a.variant & 3) == 1) use_big = true;
  if ((a.variant & 3) == 2) use_big = false;
  if (use_big && big_ok) return launch_dense_layer_big(a, s);
  if (small_ok) return launch_dense_layer_small(a, s);
  if (big_ok) return launch_dense_layer_big(a, s);
  TN_REQUIRE(false, "dense_layer: unsupported spatial size");
}

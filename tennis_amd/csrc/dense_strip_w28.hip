// dense_strip kernels of the 28 x 28 maps (one translation unit per map width: the nine K instantiations of a width take
// about a minute of hipcc each, and make builds the widths in parallel)
#include "dense_strip_impl.h"

int launch_dense_strip_w28(const DenseStripArgs &a, hipStream_t s) { return launch_strip_w<28>(a, s); }

// Tuning probe: effective shader clock under a light, latency-bound launch stream (32 workgroups, back to back)
// versus a chip-filling one.  Dependent v_fma chain: 4 clocks per instruction per wave64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void chain(float *out, long long *ticks, int n) {
  float a = threadIdx.x * 1e-6f, b = 1.000001f, c = 1e-7f;
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) a = fmaf(a, b, c);
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
  if (a == 123.f) out[0] = a;
}
int main() {
  float *out; long long *ticks;
  hipMalloc(&out, 4); hipMalloc(&ticks, 8);
  const int n = 8192;
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int grid = cfg == 0 ? 32 : cfg == 1 ? 1 : 2048, block = cfg == 2 ? 256 : 1024;
    std::vector<double> us;
    for (int rep = 0; rep < 3000; ++rep) {
      hipLaunchKernelGGL(chain, dim3(grid), dim3(block), 0, 0, out, ticks, n);
      if (rep % 100 == 99) {
        long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        us.push_back(t / 100.0);
      }
    }
    hipDeviceSynchronize();
    std::sort(us.begin(), us.end());
    // one wave per SIMD (1024 threads = 16 waves = 4 per SIMD): n FMAs * 4 clk * waves-per-SIMD
    const int wps = block == 1024 ? 4 : 1 * (grid > 256 ? 2 : 1);
    printf("grid %d block %d: chain of %d fma: median %.1f us (min %.1f max %.1f) -> %.2f GHz if %d waves/SIMD share the pipe\n",
           grid, block, n, us[us.size() / 2], us.front(), us.back(), n * 4.0 * wps / us[us.size() / 2] / 1e3, wps);
  }
  return 0;
}

// Read bandwidth of ONE workgroup (one CU) from an L2-resident buffer: 512 threads, 8-byte loads, 16 lanes per 128-byte
// line (the access shape of the Cholesky's L operands), U independent loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U, int W>
__global__ __launch_bounds__(512) void rd(const double* __restrict__ a, int n_lines, int iters, long long* out, double* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int line = ((it * U + u) * 8 + wave) * 4 + (lane >> 4);      // 4 lines per wave-load
      line %= n_lines;
      if (W == 8) v[u] = a[(size_t)line * 16 + (lane & 15)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = s;
}
int main() {
  const int n_lines = 8192;            // 1 MB
  double* a; long long* o; double* sink;
  hipMalloc(&a, n_lines * 128); hipMalloc(&o, 64); hipMalloc(&sink, 4096);
  hipMemset(a, 0, n_lines * 128);
  long long r;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((rd<4, 8>), dim3(1), dim3(512), 0, 0, a, n_lines, 256, o, sink);
    hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
    printf("U=4 : %.1f B/clk\n", 256.0 * 4 * 512 * 8 / r);
    hipLaunchKernelGGL((rd<12, 8>), dim3(1), dim3(512), 0, 0, a, n_lines, 128, o, sink);
    hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
    printf("U=12: %.1f B/clk\n", 128.0 * 12 * 512 * 8 / r);
    hipLaunchKernelGGL((rd<24, 8>), dim3(1), dim3(512), 0, 0, a, n_lines, 64, o, sink);
    hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
    printf("U=24: %.1f B/clk\n", 64.0 * 24 * 512 * 8 / r);
  }
  return 0;
}

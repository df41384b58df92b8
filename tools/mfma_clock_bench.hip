// f64 4x4x4 MFMA rate and in-kernel clock on constant-like vs random operands (DVFS give-back check).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(512) void k(double* out, long long* clk, int iters, int random) {
  int lane = threadIdx.x & 63;
  unsigned long long h = (blockIdx.x * 512ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    double ra = (double)(h & 0xfffffffffffffull) / 4503599627370496.0 * 2.0 - 1.0;
    h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
    double rb = (double)(h & 0xfffffffffffffull) / 4503599627370496.0 * 2.0 - 1.0;
    a[i] = random ? ra : 0.5;
    b[i] = random ? rb : 0.25;
  }
  double acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = 0;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int run(const char* name, int random, int iters) {
  const int grid = 256;
  double* out; long long* clk;
  CHECK(hipMalloc(&out, (size_t)grid * 512 * 8)); CHECK(hipMalloc(&clk, grid * 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<16>), dim3(grid), dim3(512), 0, 0, out, clk, 1000, random);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<16>), dim3(grid), dim3(512), 0, 0, out, clk, iters, random);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  static long long h[512]; CHECK(hipMemcpy(h, clk, grid * 16, hipMemcpyDeviceToHost));
  double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= grid;
  double ninst = (double)grid * 8 * iters * 16;
  printf("%-28s %8.3f ms  %6.2f TFLOP/s  in-kernel clock %.2f GHz  (%.1f cycles/inst/SIMD at that clock)\n", name, ms,
         ninst * 512 / (ms * 1e-3) / 1e12, ghz, ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * 2));
  return 0;
}
int main() {
  run("constant operands, 0.6 ms", 0, 2000);
  run("random operands,   0.6 ms", 1, 2000);
  run("constant operands, 30 ms", 0, 100000);
  run("random operands,   30 ms", 1, 100000);
  run("random operands,   300 ms", 1, 1000000);
  return 0;
}

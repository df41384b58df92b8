// Does anything overlap with the FP64 MFMA stream on gfx950?  Each wave runs 10 independent
// v_mfma_f64_4x4x4_4b_f64 per iteration plus K extra instructions of one kind:
//   kind 0: v_add_u32 (integer VALU)   1: v_mul_f64 (FP64 VALU)   2: ds_read_b64 (LDS, fixed address VGPR)
//   kind 3: s_add_u32 (SALU)           4: v_fma_f32 (FP32 VALU)
// Grid = 256 CUs x 2 workgroups x 4 waves (2 waves per SIMD), clock64 per wave, cycles per iteration reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND, int K>
__global__ __launch_bounds__(256, 2) void k(double* out, long long* cyc, int iters) {
  __shared__ double sm[512];
  sm[threadIdx.x] = threadIdx.x * 1e-3; sm[threadIdx.x + 256] = 0.5;
  __syncthreads();
  double acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-6, b = 0.5;
  int iv[8]; double dv[8]; float fv[8]; double lv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { iv[i] = threadIdx.x + i; dv[i] = 1.0 + 1e-9 * i; fv[i] = 1.0f + i; lv[i] = 0; }
  unsigned sacc = 0;
  const int addr = (threadIdx.x & 63) * 8;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      if (i < K) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(iv[i % 8]));
        if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dv[i % 8]) : "v"(b));
        if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(lv[i % 8]) : "v"(addr));
        if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fv[i % 8]));
      }
    }
    if (K > 10) {
#pragma unroll
      for (int i = 10; i < K; ++i) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(iv[i % 8]));
        if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dv[i % 8]) : "v"(b));
        if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(lv[i % 8]) : "v"(addr));
        if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fv[i % 8]));
      }
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += iv[i] + dv[i] + fv[i] + lv[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + sacc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K>
int run(const char* name) {
  const int grid = 512, iters = 2000;
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, grid * 256 * 8)); CHECK(hipMalloc(&cyc, grid * 4 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<KIND, K>), dim3(grid), dim3(256), 0, 0, out, cyc, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<KIND, K>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  static long long h[2048]; CHECK(hipMemcpy(h, cyc, grid * 4 * 8, hipMemcpyDeviceToHost));
  double m = 0; for (int i = 0; i < grid * 4; ++i) m += h[i]; m /= grid * 4;
  // clock64 = 100 MHz s_memtime? report both wall-based cycles at 2.4 GHz nominal and counter ticks
  printf("%-10s K=%2d  %.3f ms  wall ns/iter %.1f  (=%.1f cyc @2.1GHz per 10 MFMA + K)  ticks/iter %.1f\n", name, K, ms,
         ms * 1e6 / iters, ms * 1e6 / iters * 2.1, m / iters);
  hipFree(out); hipFree(cyc);
  return 0;
}

int main() {
  run<0, 0>("none");
  run<0, 5>("v_add_u32"); run<0, 10>("v_add_u32"); run<0, 20>("v_add_u32");
  run<1, 5>("v_mul_f64"); run<1, 10>("v_mul_f64"); run<1, 20>("v_mul_f64");
  run<2, 5>("ds_read"); run<2, 10>("ds_read"); run<2, 20>("ds_read");
  run<3, 10>("s_add"); run<3, 20>("s_add");
  run<4, 5>("v_fma_f32"); run<4, 10>("v_fma_f32"); run<4, 20>("v_fma_f32");
  return 0;
}

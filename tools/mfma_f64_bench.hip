// Microbenchmark: issue rate of the f64 matrix / vector instructions on gfx950, one wave per
// SIMD (256-thread workgroups, one per CU) and two waves per SIMD.  asm barriers keep the
// compiler from hoisting or simplifying the loop-invariant operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  int lane = threadIdx.x & 63;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed * (lane + i) + 0.5; b[i] = seed * (lane - i) + 0.25; }
  double s = 0;
  if (MODE == 0) {          // v_mfma_f64_16x16x4_f64
    double4_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = (double4_t){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  } else if (MODE == 1) {   // v_mfma_f64_4x4x4_4b_f64 (4 blocks): 1 f64 acc per lane
    double acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t];
  } else {                  // v_fma_f64
    double acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = t;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_fma(a[t & 3], b[(t >> 2) & 3], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
int run(const char* name, int grid, double flops_per_inst) {
  double* out;
  CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = MODE == 0 ? 4000 : 40000;
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, 100, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-3);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ninst = (double)grid * 4 * iters * NACC;
  double waves_per_simd = grid / 256.0;
  printf("%-34s grid=%4d  %8.3f ms  %7.2f TFLOP/s  (%.1f cycles/inst/SIMD @2.4GHz)\n", name, grid, ms,
         ninst * flops_per_inst / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * waves_per_simd));
  CHECK(hipFree(out));
  return 0;
}

int main() {
  run<28, 0>("mfma_f64_16x16x4, 28 acc", 256, 2048);
  run<28, 0>("mfma_f64_16x16x4, 28 acc, 2w/SIMD", 512, 2048);
  run<8, 0>("mfma_f64_16x16x4, 8 acc", 256, 2048);
  run<16, 1>("mfma_f64_4x4x4_4b, 16 acc", 256, 512);
  run<16, 1>("mfma_f64_4x4x4_4b, 16 acc, 2w/SIMD", 512, 512);
  run<16, 2>("v_fma_f64, 16 acc", 256, 128);
  run<16, 2>("v_fma_f64, 16 acc, 2w/SIMD", 512, 128);
  run<16, 2>("v_fma_f64, 16 acc, 4w/SIMD", 1024, 128);
  return 0;
}

// Shader clock seen by a small launch: a chain of s_nop 15 (16 cycles each) timed with the 100 MHz wall clock.
// Usage: clock_probe [workgroups]   (1 workgroup = the single-workgroup latency kernels: Cholesky, QP, inverse)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void nop_chain(long long* out, int reps) {
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("s_nop 15");
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// LDS round trip: dependent ds_read chain
__global__ void lds_chain(long long* out, int reps) {
  __shared__ int a[256];
  a[threadIdx.x] = (threadIdx.x + 1) & 255;
  __syncthreads();
  int p = threadIdx.x;
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) p = a[p];
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
// barrier cost: 256 threads
__global__ void bar_chain(long long* out, int reps) {
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) __syncthreads();
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// dependent f64 division chain
__global__ void div_chain(long long* out, int reps, double x) {
  long long t0 = wall_clock64();
  double v = x;
  for (int r = 0; r < reps; ++r) v = 1.0 / (v + 1.0);
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)v; }
}
// dependent v_mfma_f64_4x4x4_4b_f64 chain: 4 passes = 16 cycles each
__global__ void mfma_chain(long long* out, int reps, double x) {
  double acc = 0.0;
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, acc, 0, 0, 0);
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}
// dependent v_add_u32 chain
__global__ void add_chain(long long* out, int reps, int x) {
  int v = threadIdx.x;
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(x));
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
int main(int argc, char** argv) {
  int wgs = argc > 1 ? atoi(argv[1]) : 1;
  long long* d; hipMalloc(&d, 64); long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    const int reps = 20000;
    nop_chain<<<wgs, 64>>>(d, reps); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    double ticks = (double)h[0]; double cyc = (double)reps * 64 * 16;
    printf("wgs %d: nop chain %.0f ticks(100MHz) for %.0f cycles -> %.2f GHz\n", wgs, ticks, cyc, cyc / (ticks * 10.0) );
  }
  double ghz;
  { nop_chain<<<wgs, 64>>>(d, 20000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); ghz = 20000.0 * 64 * 16 / (h[0] * 10.0); }
  lds_chain<<<wgs, 256>>>(d, 100000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("dependent ds_read: %.1f ns = %.0f cycles\n", h[0] * 10.0 / 100000, h[0] * 10.0 / 100000 * ghz);
  bar_chain<<<wgs, 256>>>(d, 100000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("s_barrier (256 threads): %.1f ns = %.0f cycles\n", h[0] * 10.0 / 100000, h[0] * 10.0 / 100000 * ghz);
  div_chain<<<wgs, 256>>>(d, 100000, 0.3); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("dependent f64 add+div: %.1f ns = %.0f cycles\n", h[0] * 10.0 / 100000, h[0] * 10.0 / 100000 * ghz);
  mfma_chain<<<wgs, 64>>>(d, 20000, 0.5); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("dependent v_mfma_f64_4x4x4: %.2f ns each (16 cycles -> %.2f GHz)\n", h[0] * 10.0 / (20000.0 * 16), 16.0 / (h[0] * 10.0 / (20000.0 * 16)));
  add_chain<<<wgs, 64>>>(d, 20000, 3); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("dependent v_add_u32: %.2f ns each\n", h[0] * 10.0 / (20000.0 * 64));
  return 0;
}

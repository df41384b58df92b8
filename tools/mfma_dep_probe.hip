// v_mfma_f64_4x4x4: cycles per instruction as a function of the number of INDEPENDENT accumulators in the loop (1 wave):
// with NACC accumulators every instruction depends on the one NACC positions earlier.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters) {
  double a = threadIdx.x * 0.001 + 1.0, b = 1.0 - threadIdx.x * 0.002;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run(double* o, long long* c) {
  hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(64), 0, 0, o, c, 2000);
  long long r; (void)hipMemcpy(&r, c, 8, hipMemcpyDeviceToHost);
  printf("independent accumulators %2d: %.1f cycles per MFMA\n", NACC, r / (2000.0 * 16));
}
int main() {
  double* o; long long* c; (void)hipMalloc(&o, 512); (void)hipMalloc(&c, 64);
  run<1>(o, c); run<2>(o, c); run<4>(o, c); run<8>(o, c); run<16>(o, c);
  run<1>(o, c); run<4>(o, c); run<16>(o, c);
  return 0;
}

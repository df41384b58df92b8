// Microbenchmark of the Gram kernel's MFMA phase: NACC accumulators, per-tile runtime LDS
// offsets, software prefetch depth PF, one barrier per KT=8 snapshots (2 k-steps).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC, int PF, int KSTEPS, bool BARRIER>
__global__ __launch_bounds__(256, 1) void k(double* out, int iters, const unsigned* desc, int Wp) {
  extern __shared__ double sm[];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 2 * 4 * KSTEPS * Wp; i += 256) sm[i] = 1e-3 * (i % 17);
  __syncthreads();
  const int lane_off = (lane >> 4) * Wp + (lane & 15);
  int ao[NACC], bo[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    unsigned d = desc[wave * NACC + t];
    ao[t] = lane_off + (int)(d & 0xffffu);
    bo[t] = lane_off + (int)(d >> 16);
  }
  double4_t acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (double4_t){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const double* P = sm + (it & 1) * 2 * 4 * KSTEPS * Wp;
    constexpr int NM = KSTEPS * NACC;
    double av[NM], bv[NM];
#pragma unroll
    for (int i = 0; i < PF && i < NM; ++i) {
      const double* Pk = P + (i / NACC) * 4 * Wp;
      av[i] = Pk[ao[i % NACC]];
      bv[i] = Pk[bo[i % NACC]];
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (i + PF < NM) {
        const double* Pk = P + ((i + PF) / NACC) * 4 * Wp;
        av[i + PF] = Pk[ao[(i + PF) % NACC]];
        bv[i + PF] = Pk[bo[(i + PF) % NACC]];
      }
      acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc[i % NACC], 0, 0, 0);
    }
    if (BARRIER) __syncthreads();
  }
  double s = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int PF, int KSTEPS, bool BARRIER>
int run(const char* name) {
  const int grid = 256, Wp = 336;
  double* out; unsigned* desc;
  CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
  unsigned h[4 * 64];
  for (int w = 0; w < 4; ++w) for (int t = 0; t < 64; ++t) { unsigned a = ((w * 5 + t) % 21) * 16, b = 4 * KSTEPS * Wp * ((t >> 1) & 1) + ((w * 7 + 3 * t) % 21) * 16; h[w * NACC + t % NACC] = a | (b << 16); }
  CHECK(hipMalloc(&desc, sizeof(h))); CHECK(hipMemcpy(desc, h, sizeof(h), hipMemcpyHostToDevice));
  size_t lds = (size_t)2 * 2 * 4 * KSTEPS * Wp * 8;
  CHECK(hipFuncSetAttribute((const void*)k<NACC, PF, KSTEPS, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 20000 / KSTEPS;
  hipLaunchKernelGGL((k<NACC, PF, KSTEPS, BARRIER>), dim3(grid), dim3(256), lds, 0, out, 10, desc, Wp);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, PF, KSTEPS, BARRIER>), dim3(grid), dim3(256), lds, 0, out, iters, desc, Wp);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double nm = (double)grid * 4 * iters * NACC * KSTEPS;
  printf("%-40s %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA @2.4GHz)\n", name, ms, nm * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * KSTEPS));
  CHECK(hipFree(out)); CHECK(hipFree(desc));
  return 0;
}

int main() {
  run<28, 3, 2, true>("nacc28 pf3 k2 barrier");
  run<28, 3, 2, false>("nacc28 pf3 k2 nobarrier");
  run<28, 6, 2, true>("nacc28 pf6 k2 barrier");
  run<28, 10, 2, true>("nacc28 pf10 k2 barrier");
  run<28, 6, 4, true>("nacc28 pf6 k4 barrier");
  run<28, 1, 2, true>("nacc28 pf1 k2 barrier");
  return 0;
}

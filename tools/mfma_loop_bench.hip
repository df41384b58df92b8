// Microbenchmark of the Gram kernel's MFMA phase (4x4x4 f64, 32 tiles x 4 accumulators per wave,
// A fragment shared + rotated, B fragments from LDS with prefetch, one barrier per 2 k-steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int CTRL>
__device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// MODE 0: B from LDS (prefetch PF), barrier; 1: B from LDS, no barrier; 2: B = register (no LDS in loop), no barrier
template <int NACC, int PF, int MODE, int NT>
__global__ __launch_bounds__(NT) void k(double* out, int iters, const unsigned* desc, int Wp) {
  extern __shared__ double sm[];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 2 * 8 * Wp; i += NT) sm[i] = 1e-3 * (i % 17) + 0.1;
  __syncthreads();
  const int lane_off = (lane >> 4) * Wp + (lane & 15);
  int bo[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) bo[t] = lane_off + (int)desc[(wave & 3) * 64 + t];
  const int ao = lane_off + 16 * (wave & 7);
  double acc[NACC][4];
#pragma unroll
  for (int t = 0; t < NACC; ++t) for (int s = 0; s < 4; ++s) acc[t][s] = 0;
  constexpr int NSTEP = 2 * NACC;
  double breg = 0.3 + lane;
  for (int it = 0; it < iters; ++it) {
    const double* P = sm + (it & 1) * 2 * 8 * Wp;
    double bvs[NSTEP];
    double af[2][4];
    for (int kk = 0; kk < 2; ++kk) af[kk][0] = P[kk * 4 * Wp + ao];
    if (MODE != 2) {
#pragma unroll
      for (int i = 0; i < PF && i < NSTEP; ++i) bvs[i] = P[(i / NACC) * 4 * Wp + bo[i % NACC]];
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      af[kk][1] = row_ror<0x124>(af[kk][0]); af[kk][2] = row_ror<0x128>(af[kk][0]); af[kk][3] = row_ror<0x12c>(af[kk][0]);
    }
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int kk = step / NACC, q = step % NACC;
      double bv;
      if (MODE != 2) {
        if (step + PF < NSTEP) bvs[step + PF] = P[((step + PF) / NACC) * 4 * Wp + bo[(step + PF) % NACC]];
        bv = bvs[step];
      } else {
        asm volatile("" : "+v"(breg));
        bv = breg;
      }
      acc[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][0], bv, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][1], bv, acc[q][1], 0, 0, 0);
      acc[q][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][2], bv, acc[q][2], 0, 0, 0);
      acc[q][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][3], bv, acc[q][3], 0, 0, 0);
    }
    if (MODE == 0) __syncthreads();
  }
  double s = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int NACC, int PF, int MODE, int NT = 256>
int run(const char* name, int grid = 256) {
  const int Wp = 336;
  double* out; unsigned* desc;
  CHECK(hipMalloc(&out, (size_t)grid * NT * 8));
  unsigned h[4 * 64];
  for (int w = 0; w < 4; ++w) for (int t = 0; t < 64; ++t) h[w * 64 + t] = 8 * Wp * ((t >> 1) & 1) + ((w * 7 + 3 * t) % 21) * 16;
  CHECK(hipMalloc(&desc, sizeof(h))); CHECK(hipMemcpy(desc, h, sizeof(h), hipMemcpyHostToDevice));
  size_t lds = (size_t)2 * 2 * 8 * Wp * 8;
  CHECK(hipFuncSetAttribute((const void*)k<NACC, PF, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 3000;
  hipLaunchKernelGGL((k<NACC, PF, MODE, NT>), dim3(grid), dim3(NT), lds, 0, out, 10, desc, Wp);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, PF, MODE, NT>), dim3(grid), dim3(NT), lds, 0, out, iters, desc, Wp);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double nm = (double)grid * (NT / 64) * iters * NACC * 8;
  printf("%-44s %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA @2.4GHz)\n", name, ms, nm * 512 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * 8 * (NT / 256)));
  CHECK(hipFree(out)); CHECK(hipFree(desc));
  return 0;
}

int main() {
  run<32, 4, 0>("nacc32 pf4 lds barrier, 1 wave/SIMD");
  run<16, 4, 0, 512>("nacc16 pf4 lds barrier, 2 waves/SIMD");
  run<16, 4, 1, 512>("nacc16 pf4 lds nobarrier, 2 waves/SIMD");
  run<16, 4, 2, 512>("nacc16 regB nobarrier, 2 waves/SIMD");
  run<8, 4, 0, 1024>("nacc8 pf4 lds barrier, 4 waves/SIMD");
  return 0;
}

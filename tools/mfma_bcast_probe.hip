// Probe (round 6): do the broadcast controls CBSZ / ABID of the MAI encoding act on v_mfma_f64_4x4x4_4b_f64 on gfx950, and
// at which rate?  With them one register holding four DIFFERENT 4x4 blocks can be used as "block j in all four blocks"
// without a second LDS read or a lane permutation - what the Kronecker Gram kernel needs to keep four different WEIGHTS in
// the four blocks of the other operand (3 weight multiplies per A group and k-step instead of 9).
//
// Part 1: semantics.  Random small-integer operands; for every (cbsz, abid) the host finds, per output block b, the source
// block s(b) of the FIRST operand and t(b) of the SECOND with D_b = A_s x B_t (exact in integers), using the lane maps
// measured by mfma444_probe: operand lane = k*16 + blk*4 + index, D lane = i*16 + blk*4 + j.
// Part 2: rate of a stream of independent accumulators with and without the controls.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CBSZ, int ABID>
__global__ void pairs(unsigned long long* T) {
  const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) T[la * 64 + lb] = m;
}

template <int CBSZ, int ABID>
static int semantics(const double*, const double*, double*, const double*, const double*) {
  // the whole bilinear map: for every pair of one-hot operand lanes (la, lb) the set of output lanes that become non-zero
  unsigned long long* dT;
  static unsigned long long T[64 * 64];
  CHECK(hipMalloc(&dT, sizeof(T)));
  hipLaunchKernelGGL((pairs<CBSZ, ABID>), dim3(64, 64), dim3(64), 0, 0, dT);
  CHECK(hipMemcpy(T, dT, sizeof(T), hipMemcpyDeviceToHost));
  CHECK(hipFree(dT));
  printf("cbsz=%d abid=%d: output lane d = i*16 + blk*4 + j  <-  (first-operand lane, second-operand lane) pairs\n", CBSZ, ABID);
  for (int d = 0; d < 64; ++d) {
    if ((d & 3) != 1 || (d >> 4) != 2) continue;          // i = 2, j = 1 of every block
    printf("  d=%2d:", d);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (T[la * 64 + lb] >> d & 1) printf(" (%d,%d)", la, lb);
    printf("\n");
  }
  return 0;
}

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double seed) {
  const int lane = threadIdx.x & 63;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed * (lane + i) + 0.5; b[i] = seed * (lane - i) + 0.25; }
  double acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
      else if ((t & 3) == 0) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[(t >> 2) & 3], acc[t], 2, 0, 0);
      else if ((t & 3) == 1) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[(t >> 2) & 3], acc[t], 2, 1, 0);
      else if ((t & 3) == 2) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[(t >> 2) & 3], acc[t], 2, 2, 0);
      else acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[(t >> 2) & 3], acc[t], 1, 1, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static int run_rate(const char* name, int grid) {
  constexpr int NACC = 16;
  double* out;
  CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("%-28s grid %4d: %.3f ms, %.1f TFLOP/s\n", name, grid, ms, (double)grid * 4 * iters * NACC * 512.0 / (ms * 1e-3) / 1e12);
  }
  CHECK(hipFree(out));
  return 0;
}

int main() {
  double A[64], B[64], *dA, *dB, *dD;
  srand(3);
  for (int l = 0; l < 64; ++l) { A[l] = rand() % 7 - 3; B[l] = rand() % 5 - 2; }
  CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dD, 512));
  CHECK(hipMemcpy(dA, A, 512, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B, 512, hipMemcpyHostToDevice));
  semantics<0, 0>(dA, dB, dD, A, B);
  semantics<2, 0>(dA, dB, dD, A, B); semantics<2, 1>(dA, dB, dD, A, B); semantics<2, 2>(dA, dB, dD, A, B); semantics<2, 3>(dA, dB, dD, A, B);
  semantics<1, 0>(dA, dB, dD, A, B); semantics<1, 1>(dA, dB, dD, A, B); semantics<1, 2>(dA, dB, dD, A, B); semantics<1, 3>(dA, dB, dD, A, B);
  semantics<3, 0>(dA, dB, dD, A, B); semantics<3, 1>(dA, dB, dD, A, B);
  semantics<0, 1>(dA, dB, dD, A, B);
  for (int grid : {256, 512}) {
    if (run_rate<0>("plain", grid)) return 1;
    if (run_rate<1>("cbsz 2 / 1 broadcast", grid)) return 1;
  }
  return 0;
}

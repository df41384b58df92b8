// Timing and residual of the workgroup SPD inverses (kp_wg_inverse.h) on one workgroup: inv_probe [n [ld [shift [rank]]]]
// H = A A' + shift I with A n x rank: `30 30 1e-3 22` has the conditioning of the MPC Hessians (rank-22 data term + R)
#include "wg_inverse_blocked.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
template <int MODE>
__global__ __launch_bounds__(256) void inv_kernel(const double* H, double* out, long long* ticks, int n, int ld, int reps) {
  extern __shared__ __align__(16) double sm[];
  double* X = sm;
  double* Y = sm + ld * ld;
  double* Cm = sm + 2 * ld * ld;
  long long acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int e = threadIdx.x; e < 3 * ld * ld; e += 256) sm[e] = __builtin_nan("");      // everything outside the matrix is poison
    __syncthreads();
    for (int e = threadIdx.x; e < n * n; e += 256) {
      X[e % n + (e / n) * ld] = H[e];
      if (MODE != 0) Cm[e % n + (e / n) * ld] = H[e];       // (the copy is part of what the checked inverse costs its caller)
    }
    __syncthreads();
    long long t0 = wall_clock64();
    if (MODE == 0) wg_spd_inverse_pp(X, Y, n, ld);
    else if (MODE == 1) wg_spd_inverse_fast(X, Y, Cm, n, ld);
    else if (MODE == 3) wg_spd_inverse_tiles<false>(X, Y, n, ld, 0.0);      // the one-pivot-per-barrier sweep (n even, <= 32)
    else {
      if (n <= 16) wg_spd_inverse_mfma<1>(X, Y, n, ld);
      else if (n <= 32) wg_spd_inverse_mfma<2>(X, Y, n, ld);
      else if (n <= 48) wg_spd_inverse_mfma<3>(X, Y, n, ld);
      else wg_spd_inverse_mfma<4>(X, Y, n, ld);
    }
    long long t1 = wall_clock64();
    acc += t1 - t0;
    __syncthreads();
  }
  for (int e = threadIdx.x; e < n * n; e += 256) out[e] = X[e % n + (e / n) * ld];
  if (threadIdx.x == 0) ticks[0] = acc;
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 30;
  int ld = argc > 2 ? atoi(argv[2]) : n;
  std::vector<double> H(n * n), A(n * n);
  srand(1);
  const double shift = argc > 3 ? atof(argv[3]) : 1.0;   // H = A A' + shift I
  const int rank = argc > 4 ? atoi(argv[4]) : n;
  const double scale = argc > 5 ? atof(argv[5]) : 1.0;
  for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = i == j ? shift : 0.0;
      for (int k = 0; k < rank; ++k) s += A[i + k * n] * A[j + k * n] * scale;
      H[i + j * n] = s;
    }
  double *dH, *dO; long long* dT;
  hipMalloc(&dH, n * n * 8); hipMalloc(&dO, n * n * 8); hipMalloc(&dT, 8);
  hipMemcpy(dH, H.data(), n * n * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  std::vector<double> O0(n * n);
  for (int mode = 0; mode < ((n & 1) || n > 32 ? 3 : 4); ++mode) {
    for (int it = 0; it < 2; ++it) {
      if (mode == 0) inv_kernel<0><<<1, 256, 3 * ld * ld * 8>>>(dH, dO, dT, n, ld, reps);
      else if (mode == 1) inv_kernel<1><<<1, 256, 3 * ld * ld * 8>>>(dH, dO, dT, n, ld, reps);
      else if (mode == 2) inv_kernel<2><<<1, 256, 3 * ld * ld * 8>>>(dH, dO, dT, n, ld, reps);
      else inv_kernel<3><<<1, 256, 3 * ld * ld * 8>>>(dH, dO, dT, n, ld, reps);
    }
    std::vector<double> O(n * n); long long t;
    hipMemcpy(O.data(), dO, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(&t, dT, 8, hipMemcpyDeviceToHost);
    if (mode == 0) O0 = O;
    if (mode == 3) {
      int ndiff = 0;
      for (int e = 0; e < n * n; ++e) ndiff += memcmp(&O[e], &O0[e], 8) != 0;
      printf("elements of the pair sweep that differ from the one-pivot sweep in any bit: %d of %d\n", ndiff, n * n);
    }
    double err = 0, asym = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += H[i + k * n] * O[k + j * n];
        err = fmax(err, fabs(s - (i == j)));
        asym = fmax(asym, fabs(O[i + j * n] - O[j + i * n]));
      }
    printf("%s n %d ld %d: %.2f us per inverse (%.0f ns per scalar pivot), |H Hinv - I| = %.2e, asymmetry %.1e\n",
           mode == 0 ? "scalar sweep (pairs)" : mode == 1 ? "blocked + check" : mode == 2 ? "blocked alone" : "scalar sweep (one pivot per barrier)", n, ld, t * 0.01 / reps, t * 10.0 / reps / n, err, asym);
  }
  return 0;
}

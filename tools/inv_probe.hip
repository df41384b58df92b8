// Timing of the workgroup SPD inverse (kp_wg_inverse.h) on one workgroup: inv_probe [n [ld]]
#include "../koopman-realizations_amd/csrc/kp_wg_inverse.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__global__ __launch_bounds__(256) void inv_kernel(const double* H, double* out, long long* ticks, int n, int ld, int reps) {
  extern __shared__ __align__(16) double sm[];
  double* X = sm;
  double* Y = sm + ld * ld;
  long long acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int e = threadIdx.x; e < n * n; e += 256) X[e % n + (e / n) * ld] = H[e];
    __syncthreads();
    long long t0 = wall_clock64();
    wg_spd_inverse_pp(X, Y, n, ld);
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
  for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = 0; k < n; ++k) s += A[i + k * n] * A[j + k * n];
      H[i + j * n] = s;
    }
  double *dH, *dO; long long* dT;
  hipMalloc(&dH, n * n * 8); hipMalloc(&dO, n * n * 8); hipMalloc(&dT, 8);
  hipMemcpy(dH, H.data(), n * n * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int it = 0; it < 2; ++it) inv_kernel<<<1, 256, 2 * ld * ld * 8>>>(dH, dO, dT, n, ld, reps);
  std::vector<double> O(n * n); long long t;
  hipMemcpy(O.data(), dO, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(&t, dT, 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += H[i + k * n] * O[k + j * n];
      err = fmax(err, fabs(s - (i == j)));
    }
  printf("n %d ld %d: %.2f us per inverse (%.0f ns per pivot), |H Hinv - I| = %.2e\n", n, ld, t * 0.01 / reps, t * 10.0 / reps / n, err);
  return 0;
}

// kp_symm_gemm2 (csrc/kp_symm_gemm.h) alone: parity against a host product on sampled entries (all entries when small)
// and time per launch.  usage: symm_gemm_probe [W nc reps variant]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../koopman-realizations_amd/csrc/kp_symm_gemm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 336, nc = argc > 2 ? atoi(argv[2]) : 42 * 336, reps = argc > 3 ? atoi(argv[3]) : 20, var = argc > 4 ? atoi(argv[4]) : 0;
  std::vector<double> G((size_t)W * W), X((size_t)W * nc), C((size_t)W * nc);
  srand(1);
  for (int i = 0; i < W; ++i)
    for (int j = 0; j <= i; ++j) G[i + (size_t)j * W] = G[j + (size_t)i * W] = rand() / (double)RAND_MAX - 0.5;
  for (auto& v : X) v = rand() / (double)RAND_MAX - 0.5;
  double *dG, *dX, *dC;
  CK(hipMalloc(&dG, G.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dC, C.size() * 8));
  CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, C.size() * 8));
  CK(kp_symm_gemm2(nullptr, dG, dX, W, nc, dC, var));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
  double err = 0.0;
  const size_t total = (size_t)W * nc, stride = total > 400000 ? total / 200000 : 1;
  size_t checked = 0;
  for (size_t e = 0; e < total; e += stride, ++checked) {
    const int i = (int)(e % W), j = (int)(e / W);
    double s = 0.0;
    for (int k = 0; k < W; ++k) s += G[k + (size_t)i * W] * X[k + (size_t)j * W];
    err = fmax(err, fabs(s - C[e]));
  }
  // last row / last column explicitly (tile edges)
  for (int j = 0; j < nc; ++j) {
    double s = 0.0;
    for (int k = 0; k < W; ++k) s += G[k + (size_t)(W - 1) * W] * X[k + (size_t)j * W];
    err = fmax(err, fabs(s - C[(W - 1) + (size_t)j * W]));
  }
  for (int i = 0; i < W; ++i) {
    double s = 0.0;
    for (int k = 0; k < W; ++k) s += G[k + (size_t)i * W] * X[k + (size_t)(nc - 1) * W];
    err = fmax(err, fabs(s - C[i + (size_t)(nc - 1) * W]));
  }
  printf("W %d nc %d RA %d variant %d: max abs err %.3e over %zu sampled entries + last row/column\n", W, nc, sg2_pick_ra(W), var, err, checked);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 5; ++r) CK(kp_symm_gemm2(nullptr, dG, dX, W, nc, dC, var));
  CK(hipEventRecord(e0, nullptr));
  for (int r = 0; r < reps; ++r) CK(kp_symm_gemm2(nullptr, dG, dX, W, nc, dC, var));
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double fl = 2.0 * W * (double)W * nc;
  printf("  %.4f ms per launch, %.2f TFLOP/s (%.1f %% of 78.6), %.1f GB/s of X + C\n", ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6 * 100, 16.0 * W * nc / ms / 1e6);
  return err < 1e-10 ? 0 : 2;
}

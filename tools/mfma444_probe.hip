// Probes the lane->element maps of v_mfma_f64_4x4x4_4b_f64 (4 blocks) with one-hot operands.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* T) {
  int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
  double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) T[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) + 100 * (__popcll(m) - 1) : -1;
}
int main() {
  int* d; hipMalloc(&d, 64 * 64 * 4);
  hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
  static int T[64 * 64]; hipMemcpy(T, d, sizeof(T), hipMemcpyDeviceToHost);
  for (int blk = 0; blk < 2; ++blk) {
    printf("block %d: rows = A lane (la), cols = B lane (lb), entry = output lane\n", blk);
    for (int la = blk * 16; la < blk * 16 + 16; ++la) {
      for (int lb = 0; lb < 64; ++lb) if (T[la * 64 + lb] >= 0) printf("(%d,%d)->%d ", la, lb, T[la * 64 + lb]);
      printf("\n");
    }
  }
  return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x124, 0xf, 0xf, false);        // row_ror:4
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xf, 0xf, false);   // row_ror:8
  out[128 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x12c, 0xf, 0xf, false);  // row_ror:12
}
int main() {
  int* d; hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; ++r) { printf("row_ror:%d  lanes 0..19 receive from: ", 4 * (r + 1)); for (int i = 0; i < 20; ++i) printf("%d ", h[r * 64 + i]); printf("\n"); }
  return 0;
}

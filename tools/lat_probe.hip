// Global-memory round-trip latency seen by one workgroup on one CU: (a) dependent loads over a 1 MB buffer that is
// L2 resident, (b) a store by wave 1, workgroup barrier, load by wave 0 (the pattern between two Cholesky panels).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chase(const int* __restrict__ p, int steps, long long* out) {
  int i = 0;
  for (int k = 0; k < 64; ++k) i = p[i];          // warm
  long long t0 = clock64();
  for (int k = 0; k < steps; ++k) i = p[i];
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void st_ld(double* a, int n, long long* out) {
  __shared__ long long acc[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long tot_ld = 0, tot_bar = 0;
  double s = 0;
  for (int it = 0; it < 64; ++it) {
    long long t0 = clock64();
    if (wave == 1) a[(size_t)it * n + lane] = it + lane;
    __syncthreads();
    long long t1 = clock64();
    double v = wave == 0 ? a[(size_t)it * n + lane] : 0.0;
    s += v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = clock64();
    tot_bar += t1 - t0; tot_ld += t2 - t1;
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2] = tot_bar / 64; out[3] = tot_ld / 64; out[4] = (long long)s; }
  // plain repeated loads of lines nobody wrote in this kernel
  long long tot2 = 0;
  for (int it = 0; it < 64; ++it) {
    long long t1 = clock64();
    double v = a[(size_t)(it + 100) * n + lane];
    s += v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tot2 += clock64() - t1;
  }
  if (threadIdx.x == 0) { out[5] = tot2 / 64; out[6] = (long long)s; }
  long long tot3 = 0;
  for (int it = 0; it < 64; ++it) {      // second touch: L2 (or L1) hits
    long long t1 = clock64();
    double v = a[(size_t)(it + 100) * n + lane];
    s += v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tot3 += clock64() - t1;
  }
  if (threadIdx.x == 0) { out[7] = tot3 / 64; out[8] = (long long)s; }
}
int main() {
  const int N = 1 << 18;  // 1 MB of ints
  int* h = new int[N];
  for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i + 672 + 32) % N);   // stride 2688 + 128 B
  int* d; long long* o; double* a;
  hipMalloc(&d, N * 4); hipMalloc(&o, 128); hipMalloc(&a, 336 * 336 * 8 * 2);
  hipMemset(a, 0, 336 * 336 * 8 * 2);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
  long long r[16];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, 2000, o);
    hipMemcpy(r, o, 128, hipMemcpyDeviceToHost);
    printf("dependent loads (1 MB, L2 resident after warm-up pass): %.1f cycles per load\n", r[0] / 2000.0);
  }
  hipLaunchKernelGGL(st_ld, dim3(1), dim3(512), 0, 0, a, 336, o);
  hipMemcpy(r, o, 128, hipMemcpyDeviceToHost);
  printf("store + __syncthreads: %lld cycles; load of the just-stored line: %lld; first-touch load: %lld; second touch: %lld\n", r[2], r[3], r[5], r[7]);
  return 0;
}

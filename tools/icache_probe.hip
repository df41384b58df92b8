// What a cold instruction cache costs a single workgroup: icache_probe
// A block of straight-line code (NI independent 8-byte VALU instructions, 32 KB at NI = 4096) is executed three times in
// one launch by one wave (and by four): the first pass fetches it from L2 / HBM, the later ones from the instruction cache.
// Second kernel: the same amount of code cut into 64 pieces that are visited in a scrambled order (every piece ends in a
// jump to a far target, as the phases, loops and branches of a large kernel do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
#define REP1024(x) REP4(REP256(x))
__global__ __launch_bounds__(256) void straight(long long* out, double seed) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  long long t[4];
  for (int pass = 0; pass < 3; ++pass) {
    t[pass] = wall_clock64();
    asm volatile(REP1024("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  }
  t[3] = wall_clock64();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 3; ++i) out[i] = t[i + 1] - t[i];
    out[3] = (long long)(a0 + a1 + a2 + a3 == 12345.0);
  }
}
// 64 pieces of 64 instructions (512 B each); piece i jumps to piece (i * 37 + 11) % 64 ... a permutation cycle of length 64
template <int I> __device__ __forceinline__ void piece(double& a0, double& a1, double& a2, double& a3) {
  asm volatile(REP16("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
__global__ __launch_bounds__(256) void scattered(long long* out, double seed, int start) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  long long t[4];
  for (int pass = 0; pass < 3; ++pass) {
    t[pass] = wall_clock64();
    int i = start;
    for (int n = 0; n < 64; ++n) {
      switch (i) {
#define C(k) case k: piece<k>(a0, a1, a2, a3); break;
#define C4(k) C(k) C(k + 1) C(k + 2) C(k + 3)
#define C16(k) C4(k) C4(k + 4) C4(k + 8) C4(k + 12)
        C16(0) C16(16) C16(32) C16(48)
      }
      i = (i * 37 + 11) & 63;
    }
  }
  t[3] = wall_clock64();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 3; ++i) out[i] = t[i + 1] - t[i];
    out[3] = (long long)(a0 + a1 + a2 + a3 == 12345.0);
  }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[4];
  for (int threads = 64; threads <= 256; threads *= 4)
    for (int rep = 0; rep < 3; ++rep) {
      straight<<<1, threads>>>(d, 1.0 + rep);
      hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      printf("straight 32 KB, %3d threads, launch %d: pass 1 %.2f us, pass 2 %.2f us, pass 3 %.2f us\n", threads, rep, h[0] * 0.01, h[1] * 0.01, h[2] * 0.01);
      scattered<<<1, threads>>>(d, 1.0 + rep, rep);
      hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      printf("64 pieces of 512 B in scrambled order, %3d threads, launch %d: pass 1 %.2f us, pass 2 %.2f us, pass 3 %.2f us\n", threads, rep, h[0] * 0.01, h[1] * 0.01, h[2] * 0.01);
    }
  return 0;
}

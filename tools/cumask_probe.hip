// Does hipExtStreamCreateWithCUMask work on this box?  cumask_probe [R]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long* out, int n) {
  long long t0 = wall_clock64();
  double a = threadIdx.x;
  for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) out[blockIdx.x] = (wall_clock64() - t0) + (a < 0);
}
int main(int argc, char** argv) {
  int R = argc > 1 ? atoi(argv[1]) : 24;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32;
  printf("ncu %d\n", ncu); fflush(stdout);
  std::vector<uint32_t> mg(nw, 0u), ms(nw, 0u);
  for (int i = 0; i < ncu; ++i) ((i >= ncu - R) ? ms : mg)[i / 32] |= 1u << (i % 32);
  hipStream_t pre0, pre1; hipStreamCreateWithFlags(&pre0, hipStreamNonBlocking); hipStreamCreateWithFlags(&pre1, hipStreamNonBlocking);
  hipEvent_t early; hipEventCreateWithFlags(&early, hipEventDisableSystemFence);
  hipStream_t sg, s2;
  hipError_t e1 = hipExtStreamCreateWithCUMask(&sg, nw, mg.data());
  printf("create gram stream: %s\n", hipGetErrorString(e1)); fflush(stdout);
  hipError_t e2 = hipExtStreamCreateWithCUMask(&s2, nw, ms.data());
  printf("create solve stream: %s\n", hipGetErrorString(e2)); fflush(stdout);
  hipStreamDestroy(pre1);
  long long* d; hipMalloc(&d, 8 * 4096);
  printf("record early-created event on masked stream: %s\n", hipGetErrorString(hipEventRecord(early, sg))); fflush(stdout);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int pass = 0; pass < 2; ++pass) {
    hipStream_t s = pass ? s2 : sg;
    int wgs = pass ? R * 4 : (ncu - R) * 4;
    hipEventRecord(a, s);
    spin<<<wgs, 256, 0, s>>>(d, 200000);
    hipEventRecord(b, s);
    hipError_t e = hipStreamSynchronize(s);
    float ms_ = 0; hipEventElapsedTime(&ms_, a, b);
    printf("pass %d: %d workgroups: %s, %.3f ms\n", pass, wgs, hipGetErrorString(e), ms_); fflush(stdout);
  }
  // cross-stream dependency between the two masked streams, events without system fence
  hipEvent_t c; hipEventCreateWithFlags(&c, hipEventDisableSystemFence);
  for (int it = 0; it < 3; ++it) {
    spin<<<(ncu - R) * 2, 256, 0, sg>>>(d, 100000);
    hipError_t e = hipEventRecord(c, sg);
    printf("record: %s\n", hipGetErrorString(e)); fflush(stdout);
    e = hipStreamWaitEvent(s2, c, 0);
    printf("wait: %s\n", hipGetErrorString(e)); fflush(stdout);
    spin<<<R, 256, 0, s2>>>(d, 1000);
    e = hipStreamSynchronize(s2);
    printf("sync: %s\n", hipGetErrorString(e)); fflush(stdout);
  }
  // an event first recorded on an ordinary stream, then on a masked one; wait from an ordinary stream on a masked one
  hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableSystemFence);
  spin<<<64, 256, 0, s0>>>(d, 1000);
  printf("record s0: %s\n", hipGetErrorString(hipEventRecord(ev, s0))); fflush(stdout);
  hipStreamSynchronize(s0);
  spin<<<64, 256, 0, sg>>>(d, 1000);
  printf("record sg: %s\n", hipGetErrorString(hipEventRecord(ev, sg))); fflush(stdout);
  printf("wait s2 on ev: %s\n", hipGetErrorString(hipStreamWaitEvent(s2, ev, 0))); fflush(stdout);
  spin<<<8, 256, 0, s2>>>(d, 1000);
  hipEvent_t ev2; hipEventCreateWithFlags(&ev2, hipEventDisableTiming | hipEventDisableSystemFence);
  printf("record s2: %s\n", hipGetErrorString(hipEventRecord(ev2, s2))); fflush(stdout);
  printf("wait sg on ev2: %s\n", hipGetErrorString(hipStreamWaitEvent(sg, ev2, 0))); fflush(stdout);
  spin<<<8, 256, 64 * 1024, sg>>>(d, 1000);
  printf("sync: %s %s\n", hipGetErrorString(hipStreamSynchronize(sg)), hipGetErrorString(hipStreamSynchronize(s2))); fflush(stdout);
  return 0;
}

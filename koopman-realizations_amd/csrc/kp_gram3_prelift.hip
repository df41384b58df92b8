// kp_gram3_prelift.hip - econ lift of a dim_red dictionary, once per snapshot, for the Kronecker Gram kernel (kp_gram3.hip, PRE mode).
#include "kp_gram3_args.h"

#define KT3 8     // snapshots per tile of kp_gram3_kernel
#define NF3 3     // single-variable powers per column

// ---------------------------------------------------------------------------------------------------------------------
// Econ lift of a dim_red dictionary, once per snapshot (round 4).  kp_gram3_kernel<.,.,true> lifts the FULL dictionary of a
// tile and projects it (pcs' psi on the matrix pipe, LDS-bound) in EVERY workgroup of a snapshot split - as much work as the
// N = 34 Gram itself.  Here one thread per snapshot forms, for both sides, [zeta ; pcs' psi_full(zeta) ; 1] (Ksysid.m:1594-1618) and the
// entries the Gram kernel's tile loader wants: [psi_x (4 G4) | psi_y (4 G4) | the 9 weights ut_a ut_b, 3 zeros] x KT3 snapshots per tile; snapshots past Ns
// give zero rows (the tail mask).  The power table of a thread lives in LDS ([entry][thread]: conflict-free); the projection
// matrix is read through wave-uniform addresses ([full column][32 components], zero padded: scalar loads, the multiply-adds
// take it as a scalar operand).  Measured at 1e5 pairs, 84 full columns, 27 components (tools/prelift_time.py under rocprofv3):
// 109 us with one (snapshot, side) per thread and a row per snapshot in memory (8-byte stores scattered over 67 MB); 80 us with
// the tile layout [entry][8 snapshots] (full 64-byte sectors); 89 us with both sides in one thread (half the scalar loads, half
// the waves); 65 us with the column's factors read without branches (id 255 reads an entry of ones), which lets the loads of
// the next column run under the 64 multiply-adds of this one.  Timing-only builds: without the multiply-adds 25 us, with ONE
// matrix row for every column (scalar-cache hits) 43 us - the 21 KB matrix streams through a 16 KB scalar cache; two passes of
// 16 components (10.5 KB each) were slower (117 us: the factors are read twice).  The multiply-adds alone are 14 us at the f64
// vector peak.
// ---------------------------------------------------------------------------------------------------------------------
#define PRE_T 256
__global__ __launch_bounds__(PRE_T) void kp_gram3_pcs_transpose_kernel(const double* __restrict__ pcs, int nfull, int k, double* __restrict__ pcsT) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nfull * 32) {
    const int c = e >> 5, p = e & 31;
    pcsT[e] = p < k ? pcs[c + (size_t)p * nfull] : 0.0;
  }
}

template <int BM>
__global__ __launch_bounds__(PRE_T) void kp_gram3_prelift_kernel(const double* __restrict__ alpha, const double* __restrict__ beta, const double* __restrict__ u,
                                                                  int64_t Ns, int64_t Ns_pad, int nzeta, int D, int nfull, int k_pcs, int N, int G4,
                                                                  const uint32_t* __restrict__ recipes, const double* __restrict__ pcsT,
                                                                  double* __restrict__ out, int rl) {
  extern __shared__ double tab[];                       // [2 sides][nzeta * D][PRE_T], then one entry of ones
  const int tid = threadIdx.x;
  const int64_t snap = (int64_t)blockIdx.x * PRE_T + tid;
  if (snap >= Ns_pad) return;
  const bool valid = snap < Ns;
  const int nid = nzeta * D;
  for (int v = 0; v < nzeta; ++v) {
    const double xa = valid ? alpha[(int64_t)v * Ns + snap] : 0.0, xb = valid ? beta[(int64_t)v * Ns + snap] : 0.0;
    double pa = xa, pb = xb;
    for (int e = 0; e < D; ++e) {
      tab[(v * D + e) * PRE_T + tid] = pa;
      tab[(nid + v * D + e) * PRE_T + tid] = pb;
      pa *= xa;
      pb *= xb;
    }
  }
  tab[2 * nid * PRE_T + tid] = 1.0;                     // "no factor"
  // both sides of the snapshot in one thread: every scalar operand of the projection serves two multiply-adds.  The factors of a
  // column come from the table without a branch (id 255, "no factor", reads the entry of ones behind the powers), so that the
  // loads of the next column run under the multiply-adds of this one
  double ax[32], ay[32];
#pragma unroll
  for (int p = 0; p < 32; ++p) { ax[p] = 0.0; ay[p] = 0.0; }
#pragma unroll 4
  for (int c = 0; c < nfull; ++c) {
    const uint32_t r = recipes[c];
    double px = 1.0, py = 1.0;
#pragma unroll
    for (int f = 0; f < NF3; ++f) {
      int id = (int)((r >> (8 * f)) & 255u);
      id = id == 255 ? 2 * nid : id;
      px *= tab[id * PRE_T + tid];
      py *= tab[(id == 2 * nid ? id : nid + id) * PRE_T + tid];
    }
    const double* __restrict__ row = pcsT + (size_t)c * 32;
#pragma unroll
    for (int p = 0; p < 32; ++p) { const double w = row[p]; ax[p] = fma(px, w, ax[p]); ay[p] = fma(py, w, ay[p]); }
  }
  // tile layout [entry][KT3 snapshots]: the 8 lanes of a tile write one full 64-byte sector per entry (a row-per-snapshot layout
  // scattered 8-byte stores over 67 MB: the kernel took 109 us instead of 80)
  double* ox = out + (snap / KT3) * (int64_t)(KT3 * rl) + (snap % KT3);
  double* oy = ox + (int64_t)4 * G4 * KT3;
  for (int j = 0; j < nzeta; ++j) {
    ox[j * KT3] = tab[(j * D) * PRE_T + tid];           // (zero past Ns: the table holds zeros there)
    oy[j * KT3] = tab[(nid + j * D) * PRE_T + tid];
  }
#pragma unroll
  for (int p = 0; p < 32; ++p)
    if (p < k_pcs) { ox[(nzeta + p) * KT3] = valid ? ax[p] : 0.0; oy[(nzeta + p) * KT3] = valid ? ay[p] : 0.0; }
  ox[(N - 1) * KT3] = valid ? 1.0 : 0.0;
  oy[(N - 1) * KT3] = valid ? 1.0 : 0.0;
  for (int j = N; j < 4 * G4; ++j) { ox[j * KT3] = 0.0; oy[j * KT3] = 0.0; }
  {                                                     // the weights ut_x ut_y, (x <= y) order without (0, 0); ut = [1, u]
    double ut[BM + 1];
    ut[0] = 1.0;
#pragma unroll
    for (int i = 0; i < BM; ++i) ut[1 + i] = valid ? u[(int64_t)i * Ns + snap] : 0.0;
    double* w = ox + (int64_t)8 * G4 * KT3;
    int cnt = 0;
#pragma unroll
    for (int x = 0; x <= BM; ++x)
#pragma unroll
      for (int y = x; y <= BM; ++y) {
        if (cnt > 0) w[(cnt - 1) * KT3] = valid ? ut[x] * ut[y] : 0.0;
        ++cnt;
      }
    for (int j = cnt - 1; j < 12; ++j) w[j * KT3] = 0.0;
  }
}

hipError_t kp_gram3_pcs_transpose_launch(const double* pcs, int nfull, int k, double* pcsT, hipStream_t st) {
  hipLaunchKernelGGL(kp_gram3_pcs_transpose_kernel, dim3((nfull * 32 + PRE_T - 1) / PRE_T), dim3(PRE_T), 0, st, pcs, nfull, k, pcsT);
  return hipGetLastError();
}

hipError_t kp_gram3_prelift_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int D, int nfull, int k_pcs,
                                   int N, int G4, const uint32_t* recipes, const double* pcsT, double* out, int rl, hipStream_t st) {
  const size_t plds = (size_t)(2 * nzeta * D + 1) * PRE_T * 8;
  const dim3 pg((unsigned)((Ns_pad + PRE_T - 1) / PRE_T));
#define KP_PRELIFT(M) hipLaunchKernelGGL(kp_gram3_prelift_kernel<M>, pg, dim3(PRE_T), plds, st, alpha, beta, u, Ns, Ns_pad, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl)
  if (BM == 1) KP_PRELIFT(1);
  else if (BM == 2) KP_PRELIFT(2);
  else KP_PRELIFT(3);
#undef KP_PRELIFT
  return hipGetLastError();
}

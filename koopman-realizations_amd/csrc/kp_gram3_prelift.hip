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
// vector peak.  Counters (profiles/r04_new_kernels_pmc_summary.json): 6 650 VALU instructions per wave in 138 k cycles, 1.5 waves per
// SIMD (1e5 snapshots are only 1 564 waves) - latency-bound.  Tried on top and no better: only the raw values in LDS, powers formed
// where they are used (25 instead of 74 KB per workgroup: 64 us - there are not enough workgroups for the occupancy to matter); a
// quarter of the components per thread, four times the waves (80 us: the factors are formed four times); the matrix through the
// vector L1 instead of scalar loads (all lanes one address: 88 us).
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

// ---------------------------------------------------------------------------------------------------------------------
// The same row buffer for dictionaries with fourier / gaussian blocks and no projection (def_fourierLift Ksysid.m:694-731,
// def_gaussianLift :790-817; the EXT recipes of kp_context.hip).  kp_gram3_kernel<.,.,false,true> builds their table entries -
// sincospi, exp(-|zeta - c|^2) with the centres in LDS - in every workgroup of a snapshot split and next to the MFMA loop, which
// costs it 40 registers (plans capped at 4 quads per wave).  Here: one thread per snapshot, both sides; table entries per side
// [variable v: x^1..x^Dp, cos / sin(2 pi j x) j = 1..df][gaussian centre c], a column = product of <= 3 entries (id < 128:
// variable entry, 128 + c: gaussian, 255: none), written straight to the tile layout.
// ---------------------------------------------------------------------------------------------------------------------
#define PREX_T 128
template <int BM>
__global__ __launch_bounds__(PREX_T) void kp_gram3_prelift_ext_kernel(const double* __restrict__ alpha, const double* __restrict__ beta, const double* __restrict__ u,
                                                                      int64_t Ns, int64_t Ns_pad, int nzeta, int Dp, int df, int ng, int nfull, int G4,
                                                                      const uint32_t* __restrict__ recipes, const double* __restrict__ centres,
                                                                      double* __restrict__ out, int rl) {
  extern __shared__ double tab[];                       // [nzeta * D + ng][PREX_T], then one entry of ones
  const int tid = threadIdx.x;
  const int64_t snap = (int64_t)blockIdx.x * PREX_T + tid;
  if (snap >= Ns_pad) return;
  const bool valid = snap < Ns;
  const int D = Dp + 2 * df, nv = nzeta * D, ne = nv + ng;
  // raw values of both sides first (all loads in flight together), then the entries from registers
  constexpr int NZMAX = 16;
  double xa[NZMAX], xb[NZMAX];
#pragma unroll
  for (int v = 0; v < NZMAX; ++v) {
    xa[v] = (valid && v < nzeta) ? alpha[(int64_t)v * Ns + snap] : 0.0;
    xb[v] = (valid && v < nzeta) ? beta[(int64_t)v * Ns + snap] : 0.0;
  }
  // one side at a time through ONE table (27 entries x 128 threads = 27 KB for 20 gaussians on 6 states: five workgroups per CU;
  // a table per side left two)
  double* t0 = tab + tid;
  double* ox = out + (snap / KT3) * (int64_t)(KT3 * rl) + (snap % KT3);
  double* oy = ox + (int64_t)4 * G4 * KT3;
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const double* xs = side ? xb : xa;
#pragma unroll
    for (int v = 0; v < NZMAX; ++v) {
      if (v < nzeta) {
        double pw = xs[v];
        for (int e = 0; e < Dp; ++e) {
          t0[(v * D + e) * PREX_T] = pw;
          pw *= xs[v];
        }
        if (df > 0) {                                   // harmonics by the angle-addition recurrence, as kp_gram3_kernel's loader
          double s1, c1;
          sincospi(2.0 * xs[v], &s1, &c1);
          double cj = c1, sj = s1, cm = 1.0, sm1 = 0.0;
          for (int h = 0; h < df; ++h) {
            t0[(v * D + Dp + 2 * h) * PREX_T] = cj;
            t0[(v * D + Dp + 2 * h + 1) * PREX_T] = sj;
            const double cn = 2.0 * c1 * cj - cm, sn = 2.0 * c1 * sj - sm1;
            cm = cj; sm1 = sj; cj = cn; sj = sn;
          }
        }
      }
    }
    for (int c = 0; c < ng; ++c) {                      // exp(-|zeta - centre|^2), the centres through wave-uniform addresses
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < NZMAX; ++i)
        if (i < nzeta) {
          const double dlt = xs[i] - centres[c * nzeta + i];
          r2 = fma(dlt, dlt, r2);
        }
      t0[(nv + c) * PREX_T] = exp(-r2);
    }
    t0[(size_t)ne * PREX_T] = 1.0;
    double* o = side ? oy : ox;
#pragma unroll 2
    for (int c = 0; c < nfull; ++c) {
      const uint32_t r = recipes[c];
      double ps = 1.0;
#pragma unroll
      for (int f = 0; f < NF3; ++f) {
        const int id = (int)((r >> (8 * f)) & 255u);
        ps *= t0[(id == 255 ? ne : id >= 128 ? nv + (id - 128) : id) * PREX_T];
      }
      o[c * KT3] = valid ? ps : 0.0;
    }
  }
  for (int j = nfull; j < 4 * G4; ++j) { ox[j * KT3] = 0.0; oy[j * KT3] = 0.0; }
  {
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

hipError_t kp_gram3_prelift_ext_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int Dp, int df, int ng,
                                       int nfull, int G4, const uint32_t* recipes, const double* centres, double* out, int rl, hipStream_t st) {
  const size_t plds = ((size_t)(nzeta * (Dp + 2 * df) + ng) + 1) * PREX_T * 8;
  const dim3 pg((unsigned)((Ns_pad + PREX_T - 1) / PREX_T));
#define KP_PRELIFTX(M) hipLaunchKernelGGL(kp_gram3_prelift_ext_kernel<M>, pg, dim3(PREX_T), plds, st, alpha, beta, u, Ns, Ns_pad, nzeta, Dp, df, ng, nfull, G4, recipes, centres, out, rl)
  if (BM == 1) KP_PRELIFTX(1);
  else if (BM == 2) KP_PRELIFTX(2);
  else KP_PRELIFTX(3);
#undef KP_PRELIFTX
  return hipGetLastError();
}

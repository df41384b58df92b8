// Rank-revealing solve of the normal equations for RANK-DEFICIENT dictionaries.
//
// MATLAB's `K = Px \ Py` (Ksysid.m:1069) on a rank-deficient Px (the arm's marker coordinates without dim_red: poly-3
// bilinear has rank 252 of 336) is a QR solve with column pivoting: it warns, reports the rank and
// returns a BASIC solution - non-zero only in the rows of the r columns the pivoting selected.  The same greedy choice
// in Gram space is Cholesky with diagonal pivoting (pivot = largest remaining diagonal = largest remaining column norm).
// Every basic solution over a column subset that spans range(Px) has the same residual Px K - Py, which is what the
// parity test compares with LAPACK's pivoted QR.
//
// Round 5: BLOCKED and over many workgroups (rounds 2-4: one workgroup, left-looking, every step re-read all earlier
// columns of L from L2 - 85 MB through one CU at W = 336, 1.9 ms).  Now, per block of nb pivots (nb x W doubles of LDS):
//   kp_pivchol_panel_kernel  ONE workgroup, the serial part: pivot search over the remaining diagonal (one barrier per
//       step: every wave's candidate goes to LDS, everybody reduces the candidates), the pivot column from the TRAILING
//       matrix (one coalesced read; what earlier BLOCKS contribute is already in it) minus the columns of THIS block (LDS);
//   kp_pivchol_update_kernel  every CU: trailing matrix -= panel panel' (columns already chosen are skipped).
// The launches of all blocks are queued without a host round trip; once the remaining diagonal is rounding noise the panel
// kernel sets a flag and the launches behind it return at once.  The factor it leaves - rows in pivot order - IS the Cholesky
// factor of the selected r x r system, so the solve needs no second factorisation: gather, kp_factor_substitute_dev.
// Any width the library fits (the panel shrinks with W: 32 pivots per block up to W = 560, 6 at W = 2 940).
#include <algorithm>
#include <vector>

#include "kp_internal.h"

namespace {

struct PivState {        // device words shared by the launches of one factorisation
  int rank;              // valid once done != 0
  int done;
  int pad[2];
  double d1;             // the first pivot
};

constexpr int PC_RPT_MAX = 4;   // rows per thread of the panel kernel: W <= 4096

}  // namespace

// A := G (working copy: the trailing matrix), dg := diag(G), ipos := -1 (row not chosen), state := 0
__global__ __launch_bounds__(256) void kp_pivchol_init_kernel(const double* __restrict__ G, int W, double* __restrict__ A, double* __restrict__ dg,
                                                              int* __restrict__ ipos, PivState* __restrict__ stt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (int64_t)W * W) A[e] = G[e];
  if (e < W) {
    dg[e] = G[e * (int64_t)W + e];
    ipos[e] = -1;
  }
  if (e == 0) {
    stt->rank = 0;
    stt->done = 0;
    stt->d1 = 0.0;
  }
}

// Steps k0 .. k0 + nb - 1.  L[k][i] (column k of the factor contiguous over the rows i), Pn[c][i] the same block in LDS.
template <int NT>
__global__ __launch_bounds__(NT) void kp_pivchol_panel_kernel(const double* __restrict__ A, int W, int k0, int nb, double rel_tol, double* __restrict__ L,
                                                              double* __restrict__ dg, int* __restrict__ ipos, int* __restrict__ perm,
                                                              PivState* __restrict__ stt) {
  extern __shared__ double Pn[];         // [nb][W]
  __shared__ double red_v[2][NT / 64];
  __shared__ int red_i[2][NT / 64];
  if (stt->done) return;
  const int tid = threadIdx.x;
  double d[PC_RPT_MAX];
  bool used[PC_RPT_MAX], on[PC_RPT_MAX];
#pragma unroll
  for (int r = 0; r < PC_RPT_MAX; ++r) {
    const int i = tid + r * NT;
    on[r] = i < W;
    d[r] = on[r] ? dg[i] : -1.0;
    used[r] = on[r] ? ipos[i] >= 0 : true;
  }
  double d1 = stt->d1;
  int k = k0, stop = 0;
  for (int j = 0; j < nb && k < W; ++j, ++k) {
    // arg max of the remaining diagonal (ties: lowest index)
    double v = -1.0;
    int vi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < PC_RPT_MAX; ++r) {
      const int i = tid + r * NT;
      if (!used[r] && (d[r] > v || (d[r] == v && i < vi))) { v = d[r]; vi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(vi, o, 64);
      if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    if ((tid & 63) == 0) { red_v[j & 1][tid >> 6] = v; red_i[j & 1][tid >> 6] = vi; }
    __syncthreads();                     // (also: the column written in step j - 1 is visible to everybody)
    double pv = red_v[j & 1][0];
    int p = red_i[j & 1][0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
      const double ov = red_v[j & 1][w];
      const int oi = red_i[j & 1][w];
      if (ov > pv || (ov == pv && oi < p)) { pv = ov; p = oi; }
    }
    if (k == 0) d1 = pv;
    if (!(pv > rel_tol * d1) || !(pv > 0.0)) { stop = 1; break; }   // what is left is rounding noise of the first pivots: rank = k
    const double sq = sqrt(pv), rinv = 1.0 / sq;
#pragma unroll
    for (int r = 0; r < PC_RPT_MAX; ++r) {
      const int i = tid + r * NT;
      if (!on[r]) continue;
      double s = A[i + (size_t)p * W];
      for (int c = 0; c < j; ++c) s -= Pn[c * W + i] * Pn[c * W + p];
      double l = used[r] ? 0.0 : s * rinv;                  // rows already chosen have a zero below their own pivot (exactly)
      if (!used[r]) d[r] -= l * l;
      if (i == p) {
        l = sq;
        used[r] = true;
        perm[k] = p;
        ipos[i] = k;
      }
      Pn[j * W + i] = l;
      L[(size_t)k * W + i] = l;
    }
  }
#pragma unroll
  for (int r = 0; r < PC_RPT_MAX; ++r)
    if (on[r]) dg[tid + r * NT] = d[r];
  if (tid == 0) {
    stt->d1 = d1;
    if (stop || k >= W) {
      stt->rank = k;
      stt->done = 1;
    }
  }
}

// Trailing matrix -= L_blk' L_blk over the steps [k0, k0 + nb) for the columns not chosen yet; tile 64 rows x 64 columns,
// thread = one row x 16 columns
__global__ __launch_bounds__(256) void kp_pivchol_update_kernel(double* __restrict__ A, int W, int k0, int nb, const double* __restrict__ L,
                                                                const int* __restrict__ ipos, const PivState* __restrict__ stt) {
  __shared__ double Li[32][64], Lc[32][64];
  if (stt->done) return;
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int e = tid; e < nb * 64; e += 256) {
    const int kk = e >> 6, x = e & 63;
    Li[kk][x] = i0 + x < W ? L[(size_t)(k0 + kk) * W + i0 + x] : 0.0;
    Lc[kk][x] = c0 + x < W ? L[(size_t)(k0 + kk) * W + c0 + x] : 0.0;
  }
  __syncthreads();
  const int x = tid & 63, q0 = (tid >> 6) * 16;
  const int i = i0 + x;
  if (i >= W) return;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  for (int kk = 0; kk < nb; ++kk) {
    const double li = Li[kk][x];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += li * Lc[kk][q0 + q];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = c0 + q0 + q;
    if (c < W && ipos[c] < 0) A[i + (size_t)c * W] -= acc[q];
  }
}

// Lp (n x n, n = r padded to 16): lower triangle = factor rows in pivot order, identity padding;  Cp (n x ncp) = C[perm, :]
__global__ void kp_pivchol_gather_kernel(const double* __restrict__ L, const double* __restrict__ C, int W, int ncols, const int* __restrict__ perm,
                                         int r, int n, int ncp, double* __restrict__ Lp, double* __restrict__ Cp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nG = (int64_t)n * n;
  if (e < nG) {
    const int a = (int)(e % n), b = (int)(e / n);       // row a, column b
    double v = a == b ? 1.0 : 0.0;
    if (a < r && b < r) v = a >= b ? L[(size_t)b * W + perm[a]] : 0.0;
    Lp[e] = v;
  } else if (e < nG + (int64_t)n * ncp) {
    const int64_t f = e - nG;
    const int a = (int)(f % n), c = (int)(f / n);
    Cp[f] = (a < r && c < ncols) ? C[perm[a] + (size_t)c * W] : 0.0;
  }
}

// K (W x ncols, zeroed) [perm[a], c] = Ks[a, c]  (Ks with leading dimension n)
__global__ void kp_scatter_rows_kernel(const double* __restrict__ Ks, int W, int ncols, const int* __restrict__ perm, int r, int n,
                                       double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)r * ncols) return;
  const int a = (int)(e % r), c = (int)(e / r);
  K[perm[a] + (size_t)c * W] = Ks[a + (size_t)c * n];
}

// Basic solution of G K = C over the column subset chosen by diagonal pivoting; *rank receives its size.  The stream is
// synchronised (the rank decides the size of the second stage).
int kp_pivchol_solve_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double* K_dev, int* rank) {
  if (W > 1024 * PC_RPT_MAX) return ctx->fail(KP_ERR_ARG, "rank-revealing solve: W <= 4096");
  hipStream_t s = ctx->stream;
  const int n_max = (W + 15) / 16 * 16, ncp = (ncols + 15) / 16 * 16;
  const size_t bA = (size_t)W * W * 8, bLp = (size_t)n_max * n_max * 8, bCp = (size_t)n_max * ncp * 8, bD = (size_t)(n_max / 16) * 256 * 8;
  char* ws = (char*)ctx->workspace(9, 2 * bA + bLp + bCp + bD + (size_t)W * 16 + 256);
  if (!ws) return ctx->fail(KP_ERR_HIP, "rank-revealing solve: out of device memory");
  double* A = (double*)ws;
  double* L = (double*)(ws + bA);
  double* Lp = (double*)(ws + 2 * bA);
  double* Cp = (double*)(ws + 2 * bA + bLp);
  double* Dinv = (double*)(ws + 2 * bA + bLp + bCp);
  double* dg = (double*)(ws + 2 * bA + bLp + bCp + bD);
  int* perm = (int*)(dg + W);
  int* ipos = perm + W;
  PivState* stt = (PivState*)(((uintptr_t)(ipos + W) + 63) & ~(uintptr_t)63);
  // pivots below W * 64 eps of the first one are rounding noise of a Gram matrix (its entries carry eps * d_1)
  const double rel_tol = (double)W * 64.0 * 2.220446049250313e-16;
  hipLaunchKernelGGL(kp_pivchol_init_kernel, dim3((unsigned)(((int64_t)W * W + 255) / 256)), dim3(256), 0, s, G_dev, W, A, dg, ipos, stt);
  KP_HIP(ctx, hipGetLastError());
  const int nt = W <= 512 ? 512 : 1024;
  const int nb = std::max(1, std::min(32, (int)((size_t)140 * 1024 / ((size_t)8 * W))));
  const size_t lds = (size_t)nb * W * 8;
  static KpLdsCache lds512, lds1024;
  KP_HIP(ctx, nt == 512 ? kp_ensure_lds(lds512, (const void*)kp_pivchol_panel_kernel<512>, lds)
                        : kp_ensure_lds(lds1024, (const void*)kp_pivchol_panel_kernel<1024>, lds));
  const dim3 ugrid((W + 63) / 64, (W + 63) / 64);
  for (int k0 = 0; k0 < W; k0 += nb) {
    const int nbk = std::min(nb, W - k0);
    if (nt == 512)
      hipLaunchKernelGGL(kp_pivchol_panel_kernel<512>, dim3(1), dim3(512), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt);
    else
      hipLaunchKernelGGL(kp_pivchol_panel_kernel<1024>, dim3(1), dim3(1024), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt);
    if (k0 + nbk < W) hipLaunchKernelGGL(kp_pivchol_update_kernel, ugrid, dim3(256), 0, s, A, W, k0, nbk, (const double*)L, (const int*)ipos, (const PivState*)stt);
    KP_HIP(ctx, hipGetLastError());
  }
  PivState h{};
  KP_HIP(ctx, hipMemcpyAsync(&h, stt, sizeof(PivState), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  const int r = h.done ? h.rank : W;
  if (rank) *rank = r;
  KP_HIP(ctx, hipMemsetAsync(K_dev, 0, (size_t)W * ncols * 8, s));
  if (r == 0) return KP_OK;
  const int n = (r + 15) / 16 * 16;
  const int64_t tot = (int64_t)n * n + (int64_t)n * ncp;
  hipLaunchKernelGGL(kp_pivchol_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const double*)L, C_dev, W, ncols, (const int*)perm, r, n, ncp,
                     Lp, Cp);
  KP_HIP(ctx, hipGetLastError());
  int rc = kp_factor_substitute_dev(ctx, Lp, n, Cp, ncp, Dinv, s);
  if (rc) return rc;
  hipLaunchKernelGGL(kp_scatter_rows_kernel, dim3((unsigned)(((int64_t)r * ncols + 255) / 256)), dim3(256), 0, s, (const double*)Cp, W, ncols, (const int*)perm, r, n,
                     K_dev);
  KP_HIP(ctx, hipGetLastError());
  return KP_OK;
}

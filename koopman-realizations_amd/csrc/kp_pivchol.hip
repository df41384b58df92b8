// Rank-revealing solve of the normal equations for RANK-DEFICIENT dictionaries.
//
// MATLAB's `K = Px \ Py` (Ksysid.m:1069) on a rank-deficient Px (the arm's marker coordinates without dim_red: poly-3
// bilinear has rank 252 of 336) is a QR solve with column pivoting: it warns, reports the rank and
// returns a BASIC solution - non-zero only in the rows of the r columns the pivoting selected.  The same greedy choice
// in Gram space is Cholesky with diagonal pivoting (pivot = largest remaining diagonal = largest remaining column norm).
// One workgroup, left-looking (column k of L is formed on demand from G's pivot column and the k columns before it, the
// trailing matrix is never updated): it only has to SELECT the column subset and the rank; the selected r x r system is
// well conditioned by construction and goes through the regular Cholesky / TRSM kernels.
// Every basic solution over a column subset that spans range(Px) has the same residual Px K - Py, which is what the
// parity test compares with LAPACK's pivoted QR.
#include <algorithm>
#include <vector>

#include "kp_internal.h"

#define PC_NT 512
// L is stored [k][i] (column k contiguous over the rows i): thread i reads its entries coalesced
__global__ __launch_bounds__(PC_NT) void kp_pivchol_kernel(const double* __restrict__ G, int W, double rel_tol, double* __restrict__ L,
                                                           int* __restrict__ perm, int* __restrict__ rank_out) {
  __shared__ double red_v[PC_NT / 64];
  __shared__ int red_i[PC_NT / 64];
  __shared__ double piv_v;
  __shared__ int piv_i;
  __shared__ double lrow[512];           // L[p][0..k): the pivot row, broadcast to every thread
  const int i = threadIdx.x;
  const bool on = i < W;
  double d = on ? G[i + (size_t)i * W] : -1.0;       // remaining diagonal of row i
  bool used = !on;
  double d1 = 0.0;
  int k = 0;
  for (; k < W; ++k) {
    // arg max of the remaining diagonal (ties: lowest index)
    double v = used ? -1.0 : d;
    int vi = i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(vi, o, 64);
      if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    if ((i & 63) == 0) { red_v[i >> 6] = v; red_i[i >> 6] = vi; }
    __syncthreads();
    if (i == 0) {
      double bv = red_v[0];
      int bi = red_i[0];
      for (int w = 1; w < PC_NT / 64; ++w)
        if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
      piv_v = bv;
      piv_i = bi;
    }
    __syncthreads();
    const double pv = piv_v;
    const int p = piv_i;
    if (k == 0) d1 = pv;
    if (!(pv > rel_tol * d1) || !(pv > 0.0)) break;   // what is left is rounding noise of the first pivots: rank = k
    // pivot row of L so far
    for (int j = i; j < k; j += PC_NT) lrow[j] = L[(size_t)j * W + p];
    __syncthreads();
    const double rinv = 1.0 / sqrt(pv);
    if (on) {
      double s = G[i + (size_t)p * W];
      for (int j = 0; j < k; ++j) s -= L[(size_t)j * W + i] * lrow[j];
      const double l = used && i != p ? 0.0 : s * rinv;     // rows already chosen have a zero below their own pivot (exactly)
      L[(size_t)k * W + i] = (i == p) ? sqrt(pv) : l;
      if (!used) d -= l * l;
      if (i == p) { used = true; perm[k] = p; }
    }
    __syncthreads();
  }
  if (i == 0) *rank_out = k;
}

// Gs (r x r) = G[perm, perm], Cs (r x ncols) = C[perm, :]
__global__ void kp_gather_sys_kernel(const double* __restrict__ G, const double* __restrict__ C, int W, int ncols, const int* __restrict__ perm,
                                     int r, double* __restrict__ Gs, double* __restrict__ Cs) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nG = (int64_t)r * r;
  if (e < nG) {
    const int a = (int)(e % r), b = (int)(e / r);
    Gs[e] = G[perm[a] + (size_t)perm[b] * W];
  } else if (e < nG + (int64_t)r * ncols) {
    const int64_t f = e - nG;
    const int a = (int)(f % r), c = (int)(f / r);
    Cs[f] = C[perm[a] + (size_t)c * W];
  }
}

// K (W x ncols, zeroed) [perm[a], c] = Ks[a, c]
__global__ void kp_scatter_rows_kernel(const double* __restrict__ Ks, int W, int ncols, const int* __restrict__ perm, int r,
                                       double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)r * ncols) return;
  const int a = (int)(e % r), c = (int)(e / r);
  K[perm[a] + (size_t)c * W] = Ks[e];
}

// Basic solution of G K = C over the column subset chosen by diagonal pivoting; *rank receives its size.  The stream is
// synchronised (the rank decides the size of the second stage).
int kp_pivchol_solve_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double* K_dev, int* rank) {
  if (W > 512) return ctx->fail(KP_ERR_ARG, "rank-revealing solve: W <= 512");
  hipStream_t s = ctx->stream;
  const size_t bL = (size_t)W * W * 8, bC = (size_t)W * ncols * 8;
  char* ws = (char*)ctx->workspace(9, 2 * bL + 2 * bC + (size_t)W * 4 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "rank-revealing solve: out of device memory");
  double* L = (double*)ws;
  double* Gs = (double*)(ws + bL);
  double* Cs = (double*)(ws + 2 * bL);
  double* Ks = (double*)(ws + 2 * bL + bC);
  int* perm = (int*)(ws + 2 * bL + 2 * bC);
  int* rk = perm + W;
  // pivots below W * 64 eps of the first one are rounding noise of a Gram matrix (its entries carry eps * d_1)
  const double rel_tol = (double)W * 64.0 * 2.220446049250313e-16;
  hipLaunchKernelGGL(kp_pivchol_kernel, dim3(1), dim3(PC_NT), 0, s, G_dev, W, rel_tol, L, perm, rk);
  KP_HIP(ctx, hipGetLastError());
  int r = 0;
  KP_HIP(ctx, hipMemcpyAsync(&r, rk, sizeof(int), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  if (rank) *rank = r;
  KP_HIP(ctx, hipMemsetAsync(K_dev, 0, bC, s));
  if (r == 0) return KP_OK;
  const int64_t tot = (int64_t)r * r + (int64_t)r * ncols;
  hipLaunchKernelGGL(kp_gather_sys_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, G_dev, C_dev, W, ncols, perm, r, Gs, Cs);
  KP_HIP(ctx, hipGetLastError());
  int rc = kp_chol_solve_dev(ctx, Gs, Cs, r, ncols, Ks);
  if (rc) return rc;
  hipLaunchKernelGGL(kp_scatter_rows_kernel, dim3((unsigned)(((int64_t)r * ncols + 255) / 256)), dim3(256), 0, s, Ks, W, ncols, perm, r, K_dev);
  KP_HIP(ctx, hipGetLastError());
  return KP_OK;
}

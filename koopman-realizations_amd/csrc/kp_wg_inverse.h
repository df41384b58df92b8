// Workgroup-level SPD inverse shared by the MPC step kernel and tools/inv_probe.hip (device code only).
#pragma once
#include <hip/hip_runtime.h>

// 1/p: v_rcp_f64 and two Newton steps (the IEEE division sequence is ~3x longer and sits on every pivot's critical path)
__device__ __forceinline__ double wg_recip(double p) {
  double ip = __builtin_amdgcn_rcp(p);
  ip = __builtin_fma(__builtin_fma(-p, ip, 1.0), ip, ip);
  ip = __builtin_fma(__builtin_fma(-p, ip, 1.0), ip, ip);
  return ip;
}

// Gauss-Jordan inverse of an SPD n x n matrix (leading dimension ld) in LDS by a whole 256-thread workgroup, ping-pong
// between two buffers: pivot step k reads X and writes Y, then the two swap, so ONE barrier per pivot is enough (an
// in-place sweep needs the pivot row and column saved first: two barriers and an extra LDS round trip).
//   Y[i][j] = (i == k || j == k ? 0 : X[i][j]) - cc rr,   cc = (i == k) ? -1 : X[i][k],   rr = (j == k) ? 1/p : X[k][j] / p
// covers the pivot row, the pivot column and the pivot itself without branches.
//
// Tile version (n even): a thread owns 2 x 2 tiles, so a pivot step is 5 LDS reads per tile (ds_read2_b64 pairs down the
// columns), 6 f64 operations, the selects and 2 writes, with almost no index arithmetic.  The per-element version this
// replaces spent ~35 VALU instructions per element per pivot (760 ns per pivot for n = 30, tools/inv_probe; now 320).
// MULTI = false: n <= 32, one tile per thread.  MULTI = true: tile T = tid + 256 t, t = 0 .. ceil(nt^2 / 256) - 1, whose
// (row, column) advance by (256 % nt, 256 / nt) with one carry; the tile loop is NOT unrolled (unrolled, the 4-tile
// version needed 218 VGPRs and cost the batched MPC kernel a third of its occupancy).
template <bool MULTI>
__device__ __forceinline__ int wg_spd_inverse_tiles(double* X, double* Y, int n, int ld, double thr) {
  const int tid = threadIdx.x;
  const int nt = n >> 1;
  int bad = 0;
  const int ti0 = tid % nt, tj0 = tid / nt;
  const int dti = 256 % nt, dtj = 256 / nt;
  const int ntile = MULTI ? (nt * nt + 255) >> 8 : 1;
  // (forming the reciprocal of pivot k + 1 during step k, after the stores, was tried: slower, 383 against 319 ns per
  //  pivot at n = 30 - the step is bound by instruction issue of the single wave per SIMD, not by the pivot chain)
  for (int k = 0; k < n; ++k) {
    const double piv = X[k + k * ld];
    const double* Xk = X + k * ld;
    if (!(piv > thr)) bad = 1;
    double ip = 0.0;
    int ti = ti0, tj = tj0;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
      const bool valid = tj < nt;
      const int r2 = 2 * ti, c2 = valid ? 2 * tj : 0;
      const double* Xo = X + r2 + c2 * ld;
      const double cc0 = Xk[r2], cc1 = Xk[r2 + 1];
      const double rr0 = X[k + c2 * ld], rr1 = X[k + c2 * ld + ld];
      const double o00 = Xo[0], o10 = Xo[1], o01 = Xo[ld], o11 = Xo[ld + 1];
      if (t == 0) ip = wg_recip(piv);
      const bool i0 = r2 == k, i1 = r2 + 1 == k, j0 = c2 == k, j1 = c2 + 1 == k;
      const double a0 = i0 ? -1.0 : cc0, a1 = i1 ? -1.0 : cc1;
      const double b0 = (j0 ? 1.0 : rr0) * ip, b1 = (j1 ? 1.0 : rr1) * ip;
      const double v00 = ((i0 || j0) ? 0.0 : o00) - a0 * b0;
      const double v10 = ((i1 || j0) ? 0.0 : o10) - a1 * b0;
      const double v01 = ((i0 || j1) ? 0.0 : o01) - a0 * b1;
      const double v11 = ((i1 || j1) ? 0.0 : o11) - a1 * b1;
      if (valid) {
        double* Yo = Y + r2 + c2 * ld;
        Yo[0] = v00;
        Yo[1] = v10;
        Yo[ld] = v01;
        Yo[ld + 1] = v11;
      }
      if (MULTI) {
        ti += dti;
        tj += dtj;
        if (ti >= nt) {
          ti -= nt;
          ++tj;
        }
      }
    }
    __syncthreads();
    double* T_ = X;
    X = Y;
    Y = T_;
  }
  return bad;
}

// Per-element version for odd n without room for a padding row (ld == n).  Thread tid owns elements e = tid + 256 t;
// their (row, column) pairs advance by (256 % n, 256 / n) with one carry: no integer division in the pivot loop and no
// per-thread index table (a fully unrolled 16-entry table cost 248 VGPRs and halved the occupancy of the batched kernel).
__device__ __forceinline__ int wg_spd_inverse_elems(double* X, double* Y, int n, int ld, double thr) {
  const int tid = threadIdx.x;
  int bad = 0;
  const int di = 256 % n, dj = 256 / n, i0 = tid % n, j0 = tid / n;
  const int ne = (n * n + 255) >> 8;
  for (int k = 0; k < n; ++k) {
    const double piv = X[k + k * ld];
    if (!(piv > thr)) bad = 1;
    const double ip = wg_recip(piv);
    const double* Xk = X + k * ld;
    int i = i0, j = j0;
    for (int t0 = 0; t0 < ne; t0 += 4) {
      double v[4];
      int ix[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = j < n;
        const int jc = in ? j : n - 1;
        ix[u] = in ? i + j * ld : -1;
        const double xc = Xk[i], xr = X[k + jc * ld], xo = X[i + jc * ld];
        const double cc = i == k ? -1.0 : xc;
        const double rr = j == k ? 1.0 : xr;
        const double old = (i != k && j != k) ? xo : 0.0;
        v[u] = old - cc * (rr * ip);
        i += di;
        j += dj;
        if (i >= n) {
          i -= n;
          ++j;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ix[u] >= 0) Y[ix[u]] = v[u];
    }
    __syncthreads();
    double* T_ = X;
    X = Y;
    Y = T_;
  }
  return bad;
}

// Inverse of the SPD n x n matrix in X (n <= 64, leading dimension ld), result in X; Y is a second buffer of the same
// shape (contents destroyed).  All 256 threads must call; returns non-zero if a pivot is not positive.
// Odd n with ld > n: the matrix is bordered by a unit row and column (inverse of diag(S, 1) = diag(S^-1, 1)), so the tile
// version applies and row / column n of X are overwritten.
__device__ __forceinline__ int wg_spd_inverse_pp(double* X, double* Y, int n, int ld) {
  const int tid = threadIdx.x;
  // "not positive definite" means a pivot (a diagonal entry of a Schur complement, >= lambda_min) that is not above
  // 1e-13 of the largest diagonal entry: a matrix of condition > 1e13 has no meaningful inverse in doubles.  A bare
  // `pivot > 0` lets the rounding decide for numerically singular matrices (the inverse Schur complement of an active
  // set taken over from a primal-degenerate vertex - nearly dependent rows - is one), and the caller then iterates on noise.
  double dmax = 0.0;
  for (int k = 0; k < n; k += 8) {                 // eight independent reads per round trip (clamped index: repeats are harmless)
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = min(k + j, n - 1);
      d[j] = X[kk + kk * ld];
    }
    dmax = fmax(dmax, fmax(fmax(fmax(d[0], d[1]), fmax(d[2], d[3])), fmax(fmax(d[4], d[5]), fmax(d[6], d[7]))));
  }
  const double thr = 1e-13 * dmax;
  if ((n & 1) && ld == n) {
    const int bad = wg_spd_inverse_elems(X, Y, n, ld, thr);     // odd number of steps: the result is in Y
    for (int e = tid; e < n * n; e += 256) X[e] = Y[e];
    __syncthreads();
    return bad;
  }
  if (n & 1) {
    for (int e = tid; e <= n; e += 256) {
      X[e + n * ld] = e == n ? 1.0 : 0.0;
      X[n + e * ld] = e == n ? 1.0 : 0.0;
    }
    __syncthreads();
    ++n;
  }
  if (n <= 32) return wg_spd_inverse_tiles<false>(X, Y, n, ld, thr);
  return wg_spd_inverse_tiles<true>(X, Y, n, ld, thr);
}

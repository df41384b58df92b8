// Workgroup-level SPD inverse shared by the MPC step kernel and tools/inv_probe.hip (device code only).
#pragma once
#include <hip/hip_runtime.h>

// Barrier between phases that exchange data through LDS only.  The __syncthreads() of the ROCm 7.2 hipcc also waits for every
// global store of the wave to be acknowledged (it emits s_waitcnt vmcnt(0) in front of s_barrier; other builds of the compiler
// lower the same call to s_waitcnt lgkmcnt(0) only - code that NEEDS vmcnt(0) at a barrier, e.g. behind LDS-DMA loads, writes the
// wait out: kp_mpc.hip); in a single-workgroup kernel that has just written results, exports or time stamps - some of them into
// host memory - that is microseconds of waiting for nothing.
__device__ __forceinline__ void wg_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
    wg_lds_barrier();
    double* T_ = X;
    X = Y;
    Y = T_;
  }
  return bad;
}

// Pair version (n even, n <= 32): TWO pivots per barrier with the 2 x 2 tiles held in registers.  A pivot step of the tile
// version is a barrier, an LDS round trip for nine operands, the reciprocal chain and four stores (~770 cycles, most of it
// latency: the workgroup has one wave per SIMD and nothing to overlap with).  Here a thread keeps its tile in registers for
// the whole sweep and only the two pivot rows and columns of the next pair go through LDS (written by the 2 nt - 1 threads
// that own them, into the other buffer: one barrier per pair).  Pivot k + 1 is applied from values every thread forms itself
// with the arithmetic the owner uses - Y[i][k+1], Y[k+1][j] and Y[k+1][k+1] of the intermediate matrix - so every element
// sees exactly the two scalar updates of the one-pivot sweep, in the same order and with the same roundings (tools/inv_probe
// compares the two bit by bit): this is NOT elimination with an inverted 2 x 2 pivot block.
// The matrix is read from R0 (X itself, or Y when the caller's copy lies there: both buffers are exchange space from the first
// pair on); the inverse is written to X.
__device__ __forceinline__ int wg_spd_inverse_pairs(double* X, double* Y, int n, int ld, double thr, double* R0) {
  const int tid = threadIdx.x;
  const int nt = n >> 1;
  const int ti = tid % nt, tj = tid / nt;
  const bool valid = tj < nt;
  const int r2 = 2 * ti, c2 = valid ? 2 * tj : 0;
  double o00 = R0[r2 + c2 * ld], o10 = R0[r2 + 1 + c2 * ld], o01 = R0[r2 + c2 * ld + ld], o11 = R0[r2 + 1 + c2 * ld + ld];
  int bad = 0;
  double* R = R0;
  double* Wn = R0 == X ? Y : X;
  for (int kt = 0; kt < nt; ++kt) {
    const int k = 2 * kt;
    const double* Rk = R + k * ld;
    const double pa = Rk[k], pc = Rk[k + 1], pb = Rk[k + ld], pd = Rk[k + 1 + ld];      // pivot block [pa pb; pc pd]
    const double ck0 = Rk[r2], ck1 = Rk[r2 + 1], cl0 = Rk[r2 + ld], cl1 = Rk[r2 + 1 + ld];  // my rows in columns k, k + 1
    const double rk0 = R[k + c2 * ld], rl0 = R[k + 1 + c2 * ld], rk1 = R[k + c2 * ld + ld], rl1 = R[k + 1 + c2 * ld + ld];
    const bool inrow = ti == kt, incol = valid && tj == kt;
    // ---- pivot k ----
    if (!(pa > thr)) bad = 1;
    const double ip1 = wg_recip(pa);
    const double a0 = inrow ? -1.0 : ck0, a1 = ck1;
    const double b0 = (incol ? 1.0 : rk0) * ip1, b1 = rk1 * ip1, bb = pb * ip1;
    const double p2 = pd - pc * bb;
    const double y00 = ((inrow || incol) ? 0.0 : o00) - a0 * b0;
    const double y10 = (incol ? 0.0 : o10) - a1 * b0;
    const double y01 = (inrow ? 0.0 : o01) - a0 * b1;
    const double y11 = o11 - a1 * b1;
    const double yl0 = (inrow ? 0.0 : cl0) - a0 * bb, yl1 = cl1 - a1 * bb;       // column k + 1 of the intermediate matrix, my rows
    const double yr0 = (incol ? 0.0 : rl0) - pc * b0, yr1 = rl1 - pc * b1;       // row k + 1, my columns
    // ---- pivot k + 1 ----
    if (!(p2 > thr)) bad = 1;
    const double ip2 = wg_recip(p2);
    const double e0 = yl0, e1 = inrow ? -1.0 : yl1;
    const double f0 = yr0 * ip2, f1 = (incol ? 1.0 : yr1) * ip2;
    o00 = y00 - e0 * f0;
    o10 = (inrow ? 0.0 : y10) - e1 * f0;
    o01 = (incol ? 0.0 : y01) - e0 * f1;
    o11 = ((inrow || incol) ? 0.0 : y11) - e1 * f1;
    if (valid && kt + 1 < nt && (ti == kt + 1 || tj == kt + 1)) {      // the next pair's rows and columns
      double* Wo = Wn + r2 + c2 * ld;
      Wo[0] = o00;
      Wo[1] = o10;
      Wo[ld] = o01;
      Wo[ld + 1] = o11;
    }
    wg_lds_barrier();
    double* T_ = R;
    R = Wn;
    Wn = T_;
  }
  if (valid) {
    double* Xo = X + r2 + c2 * ld;
    Xo[0] = o00;
    Xo[1] = o10;
    Xo[ld] = o01;
    Xo[ld + 1] = o11;
  }
  wg_lds_barrier();
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
    wg_lds_barrier();
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
// in_Y: the matrix lies in Y instead of X (saves the caller a copy when its matrix is dead afterwards anyway).
__device__ __forceinline__ int wg_spd_inverse_pp(double* X, double* Y, int n, int ld, bool in_Y = false) {
  const int tid = threadIdx.x;
  if (in_Y && ((n & 1) || n > 32)) {               // only the pair sweep reads its matrix from either buffer
    for (int e = tid; e < n * ld; e += 256) X[e] = Y[e];
    wg_lds_barrier();
    in_Y = false;
  }
  const double* D = in_Y ? Y : X;
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
      d[j] = D[kk + kk * ld];
    }
    dmax = fmax(dmax, fmax(fmax(fmax(d[0], d[1]), fmax(d[2], d[3])), fmax(fmax(d[4], d[5]), fmax(d[6], d[7]))));
  }
  const double thr = 1e-13 * dmax;
  if ((n & 1) && ld == n) {
    const int bad = wg_spd_inverse_elems(X, Y, n, ld, thr);     // odd number of steps: the result is in Y
    for (int e = tid; e < n * n; e += 256) X[e] = Y[e];
    wg_lds_barrier();
    return bad;
  }
  if (n & 1) {
    for (int e = tid; e <= n; e += 256) {
      X[e + n * ld] = e == n ? 1.0 : 0.0;
      X[n + e * ld] = e == n ? 1.0 : 0.0;
    }
    wg_lds_barrier();
    ++n;
  }
  if (n <= 32) return wg_spd_inverse_pairs(X, Y, n, ld, thr, in_Y ? Y : X);
  return wg_spd_inverse_tiles<true>(X, Y, n, ld, thr);
}

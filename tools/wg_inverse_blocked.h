// EXPERIMENT (not part of the product): blocked Gauss-Jordan inverse on the matrix pipe for the MPC step kernel, with a
// residual check / Newton-Schulz correction, timed by tools/inv_probe against the scalar sweep the product uses
// (koopman-realizations_amd/csrc/kp_wg_inverse.h).  Measured at n = 30: scalar sweep 9.6 us, blocked sweep alone 5.7 us,
// blocked + check 7.2 us (+0.7 us when the correction runs).  Why it was NOT adopted: block elimination with explicitly
// inverted 4 x 4 pivots is only conditionally stable (backward error ~ eps cond(pivot block)), so its result must be
// checked and, at the conditioning of the MPC matrices (inverse Schur complements at full vertices: cond ~1e7), usually
// corrected - which leaves 1.7 us of the 3.9 us gain - and the Goldfarb-Idnani solver at the primal-degenerate vertices of
// the stored MATLAB runs (33 tight rows on 30 variables) failed on 1-7 of 299 replayed steps with the 1e-10-level
// differences of this inverse, while it solves all of them with the scalar sweep.
#pragma once
#include "../koopman-realizations_amd/csrc/kp_wg_inverse.h"

// OR of `v` over the workgroup, the same value in every thread; safe to call back to back (the word is not touched again
// until every thread has read it).
__device__ __forceinline__ int wg_or(int v) {
  __shared__ int wg_or_word;
  if (threadIdx.x == 0) wg_or_word = 0;
  __syncthreads();
  if (v) atomicOr(&wg_or_word, v);
  __syncthreads();
  const int r = wg_or_word;
  __syncthreads();
  return r;
}

// ---- blocked Gauss-Jordan on the matrix pipe --------------------------------------------------------------------------
// The same ping-pong sweep with 4 x 4 pivot BLOCKS: n / 4 barriers instead of n, and the rank-4 update of the whole matrix
// is v_mfma_f64_4x4x4_4b (4 output tiles per instruction, 16 instructions for a 32 x 32 matrix, 4 per wave).  Step K
// reads X, writes every tile of Y:
//     P = X_KK^-1;   Y_KK = P;   Y_KJ = P X_KJ;   Y_IK = -X_IK P;   Y_IJ = X_IJ - X_IK (P X_KJ)
// * P by 2 x 2 block elimination on the 10 distinct elements of the symmetric pivot block, redundantly in every lane
//   (broadcast LDS reads, ~45 flops, two reciprocals); the lane keeps element P[lane & 3][lane >> 4].  P is symmetric, so
//   that element is at once the A-operand layout (row = lane & 3, k = lane >> 4) and the B-operand layout
//   (k = lane >> 4, col = lane & 3) of P - no transposition.  (4 x 4 cofactors, one element per lane, were tried first:
//   a third faster, but their error grows with the CUBE of the block's condition number - the MPC Hessians, rank-22
//   data term + 1e-3 I, broke them.)
// * every wave forms the whole row panel -P X_KJ itself (ng = ceil(nb / 4) instructions): the MFMA output layout IS the
//   B-operand layout, so the panel goes from the accumulators straight into the update; for the tile J = K the operand
//   is -P and the addend 0, which yields -X_IK P.  Tiles of block row K take -operand (P X_KJ, and P itself for J = K).
// * n is padded to a multiple of 4 by a unit diagonal in the loads (addresses clamped, values selected); the padding is
//   never stored.  Wave w owns block rows I = w, w + 4, ...
// A pivot block whose leading minors are not all positive reports `bad`, like a non-positive pivot of the scalar sweep.
// Q = ceil(nb / 4): block rows per wave and groups of 4 block columns (Q = 2 covers 16 < n <= 32, Q = 4 up to 64).
// Straight-line step: every read and every MFMA is unconditional (block rows or columns past the matrix compute values
// that are never stored); only the stores are predicated.
// Padding: only what reaches a stored element is masked.  Stored elements (r, c < n) see the padding through the
// contraction index of the LAST block step alone, so the column-K operand and the row-K operand are zeroed where
// 4K + k >= n (one lane predicate per step) and the pivot block of that step is read with its unit padding; everything
// else is read as it lies (LDS reads past the matrix return some value that only reaches elements never stored).
template <int Q>
__device__ __forceinline__ int wg_spd_inverse_mfma(double* X0, double* Y0, int n, int ld) {
  constexpr int NS = Q, NG = Q;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane >> 4, lj = lane & 3, blk = (lane >> 2) & 3;
  const int nb = (n + 3) >> 2;
  double* X = X0;
  double* Y = Y0;
  // operand offsets of block row I = wave (block row wave + 4 s lies 16 rows further down)
  int ob[NG], oc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    ob[g] = li + (4 * (4 * g + blk) + lj) * ld;
    oc[g] = 4 * wave + li + (4 * (4 * g + blk) + lj) * ld;
  }
  const int oa = 4 * wave + lj + li * ld;
  constexpr bool EARLY = Q <= 2;                   // all tiles of the wave in registers before P is known (else row by row:
                                                   // the 4 x 4 = 16 tiles of Q = 4 would cost the MPC kernel its occupancy)
  int bad = 0;
  for (int K = 0; K < nb; ++K) {
    const int k0 = 4 * K;
    const double* Xp = X + k0 * (ld + 1);          // pivot block
    const double* Xr = X + k0;                     // block row K
    const double* Xc = X + k0 * ld;                // block column K
    // the 10 distinct elements of the symmetric pivot block, wave-uniform addresses (LDS broadcast reads)
    double x00 = Xp[0], x01 = Xp[ld], x02 = Xp[2 * ld], x03 = Xp[3 * ld], x11 = Xp[1 + ld], x12 = Xp[1 + 2 * ld], x13 = Xp[1 + 3 * ld],
           x22 = Xp[2 + 2 * ld], x23 = Xp[2 + 3 * ld], x33 = Xp[3 + 3 * ld];
    if (K == nb - 1 && (n & 3)) {                  // (uniform) unit padding of the last block
      const int v = n & 3;                         // valid rows / columns of this block: 1..3
      x03 = 0.0; x13 = 0.0; x23 = 0.0; x33 = 1.0;
      if (v < 3) { x02 = 0.0; x12 = 0.0; x22 = 1.0; }
      if (v < 2) { x01 = 0.0; x11 = 1.0; }
    }
    // the row-panel operands and (EARLY) this wave's tiles do not depend on P: their reads are issued now
    const bool kin = k0 + li < n;
    double bv[NG], av[EARLY ? NS : 1], cv[EARLY ? NS : 1][NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = Xr[ob[g]];
    if (EARLY) {
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        av[s_] = Xc[oa + 16 * s_];
#pragma unroll
        for (int g = 0; g < NG; ++g) cv[s_][g] = X[oc[g] + 16 * s_];
      }
    }
    // P = X_KK^-1 by 2 x 2 block elimination ([A B; B' D]: A^-1, S = D - B'A^-1 B, S^-1; each 2 x 2 inverse by its
    // adjugate, which is as accurate as elimination at that size), redundantly in every lane - backward stable on SPD
    // blocks of any conditioning, unlike 4 x 4 cofactors - then the lane's element P[lj][li] by selects.
    const double dA = x00 * x11 - x01 * x01;
    const double iA = wg_recip(dA);
    const double a00 = x11 * iA, a01 = -x01 * iA, a11 = x00 * iA;
    const double w00 = a00 * x02 + a01 * x12, w01 = a00 * x03 + a01 * x13, w10 = a01 * x02 + a11 * x12, w11 = a01 * x03 + a11 * x13;
    const double s00 = x22 - (x02 * w00 + x12 * w10), s01 = x23 - (x02 * w01 + x12 * w11), s11 = x33 - (x03 * w01 + x13 * w11);
    const double dS = s00 * s11 - s01 * s01;
    const double iS = wg_recip(dS);
    const double t00 = s11 * iS, t01 = -s01 * iS, t11 = s00 * iS;
    const double q00 = -(w00 * t00 + w01 * t01), q01 = -(w00 * t01 + w01 * t11), q10 = -(w10 * t00 + w11 * t01), q11 = -(w10 * t01 + w11 * t11);
    const double p00 = a00 - (q00 * w00 + q01 * w01), p01 = a01 - (q00 * w10 + q01 * w11), p11 = a11 - (q10 * w10 + q11 * w11);
    if (!(x00 > 0.0) || !(dA > 0.0) || !(s00 > 0.0) || !(dS > 0.0)) bad = 1;      // leading minors of the block
    const double c0 = lj == 0 ? p00 : lj == 1 ? p01 : lj == 2 ? q00 : q01;       // column li of the symmetric P, row lj
    const double c1 = lj == 0 ? p01 : lj == 1 ? p11 : lj == 2 ? q10 : q11;
    const double c2 = lj == 0 ? q00 : lj == 1 ? q10 : lj == 2 ? t00 : t01;
    const double c3 = lj == 0 ? q01 : lj == 1 ? q11 : lj == 2 ? t01 : t11;
    const double p = li == 0 ? c0 : li == 1 ? c1 : li == 2 ? c2 : c3;
    const double np_ = -p;
    // ---- row panel -P X_KJ (all of it in every wave), -P in place of the tile J = K ----
    double nr[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      nr[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(np_, kin ? bv[g] : 0.0, 0.0, 0, 0, 0);
      if (4 * g + blk == K) nr[g] = np_;
    }
    // ---- this wave's block rows ----
    if (EARLY) {
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const int I = wave + 4 * s_;
        const int r = 4 * I + li;
        const double a_ = kin ? av[s_] : 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          double o = __builtin_amdgcn_mfma_f64_4x4x4f64(a_, nr[g], 4 * g + blk == K ? 0.0 : cv[s_][g], 0, 0, 0);
          if (I == K) o = -nr[g];
          if (r < n && 4 * (4 * g + blk) + lj < n) Y[oc[g] + 16 * s_] = o;
        }
      }
    } else {
#pragma unroll 1
      for (int s_ = 0; s_ < NS; ++s_) {
        const int I = wave + 4 * s_;
        const int r = 4 * I + li;
        const double av1 = Xc[oa + 16 * s_];
        double c1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) c1[g] = X[oc[g] + 16 * s_];
        const double a_ = kin ? av1 : 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          double o = __builtin_amdgcn_mfma_f64_4x4x4f64(a_, nr[g], 4 * g + blk == K ? 0.0 : c1[g], 0, 0, 0);
          if (I == K) o = -nr[g];
          if (r < n && 4 * (4 * g + blk) + lj < n) Y[oc[g] + 16 * s_] = o;
        }
      }
    }
    __syncthreads();
    double* T_ = X;
    X = Y;
    Y = T_;
  }
  if (X != X0) {                                   // odd number of block steps: the result is in the second buffer
    for (int e = tid; e < n * ld; e += 256) X0[e] = X[e];
    __syncthreads();
  }
  return wg_or(bad);
}

// ---- residual check and one Newton-Schulz correction of a computed inverse -------------------------------------------
// The blocked sweep applies explicitly inverted 4 x 4 pivot blocks; its backward error grows with the condition number of
// those blocks (block elimination with inverted pivots is only conditionally stable), where the scalar sweep's does not.
// So its result is CHECKED: R = I - C X on the matrix pipe (C = the original matrix), and
//     max|R| <= 1e-13          accepted as it is (the level of the scalar sweep on well-conditioned matrices);
//     max|R| <= 1e-6           one correction X <- X + X R  (residual R^2 <= 1e-12, down to the eps cond(C) of forming R);
//     otherwise (or NaN)       rejected: the caller repeats the inverse with the scalar sweep (C is still intact then).
// Tiles as in the sweep: wave w owns block rows w, w + 4, ..; the products are accumulated in registers, so R can take
// the place of C and the corrected X its own place with one barrier each.
template <int Q, bool NEG>
__device__ __forceinline__ void wg_tiles_mm(const double* A, const double* B, int n, int ld, double (&acc)[Q][Q]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int li = lane >> 4, lj = lane & 3, blk = (lane >> 2) & 3;
  const int nb = (n + 3) >> 2;
  int oa = 4 * wave + lj + li * ld;                      // A-operand: (row 4I + lj, k = 4Kb + li)
  int ob[Q];                                             // B-operand: (k = 4Kb + li, column 4(4g + blk) + lj)
#pragma unroll
  for (int g = 0; g < Q; ++g) ob[g] = li + (4 * (4 * g + blk) + lj) * ld;
  for (int Kb = 0; Kb < nb; ++Kb) {
    const bool kin = 4 * Kb + li < n;                    // contraction index past the matrix: both operands zero
    double bv[Q];
#pragma unroll
    for (int g = 0; g < Q; ++g) {
      const double v = B[ob[g] + 4 * Kb];
      bv[g] = kin ? v : 0.0;
    }
#pragma unroll
    for (int s_ = 0; s_ < Q; ++s_) {
      const double v = A[oa + 16 * s_ + 4 * Kb * ld];
      const double a_ = kin ? (NEG ? -v : v) : 0.0;
#pragma unroll
      for (int g = 0; g < Q; ++g) acc[s_][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_, bv[g], acc[s_][g], 0, 0, 0);
    }
  }
}

// X: computed inverse of the matrix in Cm (both n x n, leading dimension ld).  Returns 0: X accepted (possibly corrected;
// Cm destroyed), 1: rejected (Cm intact).  All 256 threads must call.
template <int Q>
__device__ __forceinline__ int wg_inverse_check_refine(double* X, double* Cm, int n, int ld) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int li = lane >> 4, lj = lane & 3, blk = (lane >> 2) & 3;
  double acc[Q][Q];
  int code = 0;
#pragma unroll
  for (int s_ = 0; s_ < Q; ++s_)
#pragma unroll
    for (int g = 0; g < Q; ++g) {
      const int r = 4 * (wave + 4 * s_) + li, c = 4 * (4 * g + blk) + lj;
      acc[s_][g] = r == c ? 1.0 : 0.0;
    }
  wg_tiles_mm<Q, true>(Cm, X, n, ld, acc);               // R = I - C X
#pragma unroll
  for (int s_ = 0; s_ < Q; ++s_)
#pragma unroll
    for (int g = 0; g < Q; ++g) {
      const int r = 4 * (wave + 4 * s_) + li, c = 4 * (4 * g + blk) + lj;
      if (r < n && c < n) {
        const double ar = fabs(acc[s_][g]);
        if (!(ar <= 1e-13)) code |= 1;
        if (!(ar <= 1e-6)) code |= 2;
      }
    }
  code = wg_or(code);                                    // (also: every wave has finished reading Cm)
  if (code == 0) return 0;
  if (code & 2) return 1;
#pragma unroll
  for (int s_ = 0; s_ < Q; ++s_)
#pragma unroll
    for (int g = 0; g < Q; ++g) {
      const int r = 4 * (wave + 4 * s_) + li, c = 4 * (4 * g + blk) + lj;
      if (r < n && c < n) Cm[r + c * ld] = acc[s_][g];
      acc[s_][g] = (r < n && c < n) ? X[r + c * ld] : 0.0;
    }
  __syncthreads();
  wg_tiles_mm<Q, false>(X, Cm, n, ld, acc);              // X + X R
  __syncthreads();
#pragma unroll
  for (int s_ = 0; s_ < Q; ++s_)
#pragma unroll
    for (int g = 0; g < Q; ++g) {
      const int r = 4 * (wave + 4 * s_) + li, c = 4 * (4 * g + blk) + lj;
      if (r < n && c < n) X[r + c * ld] = acc[s_][g];
    }
  __syncthreads();
  return 0;
}

// Blocked sweep on the matrix pipe + check, any n <= 64 (n = 30: 5.7 us + 0.5 us against 9.6 us for the scalar sweep; it
// needs ~40 VGPRs more, which would cost the BATCHED MPC kernel one of its three workgroups per CU - that kernel keeps the
// scalar sweep; the single-problem kernel, where only latency counts, uses this).
// X: the matrix, replaced by its inverse.  Y: scratch.  Cm: a COPY of the matrix (destroyed).  Falls back to the scalar
// sweep when the check rejects the result; returns non-zero if the matrix is not numerically positive definite.
__device__ __forceinline__ int wg_spd_inverse_fast(double* X, double* Y, double* Cm, int n, int ld) {
  int bad, rej;
  if (n <= 16) { bad = wg_spd_inverse_mfma<1>(X, Y, n, ld); rej = bad ? 1 : wg_inverse_check_refine<1>(X, Cm, n, ld); }
  else if (n <= 32) { bad = wg_spd_inverse_mfma<2>(X, Y, n, ld); rej = bad ? 1 : wg_inverse_check_refine<2>(X, Cm, n, ld); }
  else if (n <= 48) { bad = wg_spd_inverse_mfma<3>(X, Y, n, ld); rej = bad ? 1 : wg_inverse_check_refine<3>(X, Cm, n, ld); }
  else { bad = wg_spd_inverse_mfma<4>(X, Y, n, ld); rej = bad ? 1 : wg_inverse_check_refine<4>(X, Cm, n, ld); }
  if (!rej) {
    // exactly symmetric result (the correction X + X R is symmetric only to the level of R; the active-set solver updates
    // its inverse Schur complement with symmetric formulas)
    for (int e = threadIdx.x; e < n * n; e += 256) {
      const int i = e % n, j = e / n;
      if (i < j) {
        const double v = 0.5 * (X[i + j * ld] + X[j + i * ld]);
        X[i + j * ld] = v;
        X[j + i * ld] = v;
      }
    }
    __syncthreads();
    return 0;
  }
  for (int e = threadIdx.x; e < n * ld; e += 256) X[e] = Cm[e];        // the scalar sweep decides (and reports non-SPD input)
  __syncthreads();
  return wg_spd_inverse_pp(X, Y, n, ld);
}

// Per-step MPC: condensed QP assembly for the linear / bilinear Koopman model and a dense
// strictly convex QP solve, one workgroup per problem.
//
// Replaces Kmpc.get_mpcInput (Kmpc.m:329-387), get_mpcInput_bilinear(_iter) (:750-904) with
// get_costB/H/G/D_bilinear (:569-622), and the quadprog call (:383,810,883).
//
// Algebra: P_k = proj*A^k, Beta(z) = B*kron(I_m,z), S_k = P_k*Beta(z_k);
// (Chat*Bhat)[i,j] = S_{i-j-1} for i > j;  H = (CB)'Q(CB) + R;  f = 2 (CB)'Q(Chat*Ahat*z - Yr).
// The reference rebuilds the (N(Np+1)) x (m Np) matrix Bhat four times per step with dense
// matrix powers; here P_k is precomputed once and only the nproj x m blocks S_k are formed.
//
// QP: Goldfarb-Idnani dual active set run by one wavefront, with the inverse of the active
// constraints' Schur complement (N'H^-1 N)^-1 updated by bordering / deletion formulas, so an
// iteration is O(n^2) lane-parallel work without triangular solves.
#include <algorithm>
#include <cmath>
#include <vector>

#include <cstring>

#include "kp_internal.h"
#include "kp_wg_inverse.h"
#include <type_traits>

#define QP_MAXN 64       // variables (one wave handles <= 64)
#define QP_MAXIT 2000
#define QP_MAX_RELEASE 6   // rows a warm start may release before it is abandoned for the cold start

struct kp_mpc {
  kp_ctx* ctx = nullptr;
  int model_type = 0, N = 0, m = 0, Np = 0, nproj = 0, nvar = 0, nrows = 0, mb = 0;
  double q_run = 0, q_term = 0;
  // device
  double* A = nullptr;     // N x N
  double* B = nullptr;     // N x mb
  double* P = nullptr;     // (Np+1) x [nproj x N]  (P_k column-major nproj x N)
  double* S0 = nullptr;    // linear model: Np x [nproj x m]
  double* PB = nullptr;    // bilinear model: [Np * m * nproj][N]  rows of P_k B_i (S_k = P_k Beta(z) = these rows times z); inside P's allocation
  int stage_doubles = 0;   // size of that allocation
  double* r = nullptr;     // m
  double* Aq = nullptr;    // nrows x nvar column-major (constant: L = F, tack rows)
  double* bq0 = nullptr;   // nrows (c and zeros for the tack rows)
  double* Anorm = nullptr; // nrows
  double* ellv = nullptr;  // K x nrows
  int* ellc = nullptr;     // K x nrows
  int ellK = 1;
  double* work = nullptr;  // per-problem QP export: Hq (nvar^2) | f (nvar)
  size_t work_problems = 0;
  double *d_in = nullptr, *d_out = nullptr;     // d_out: x (nvar) and z (N) per problem, then the status words
  double *h_in = nullptr, *h_out = nullptr;     // pinned staging buffers: one copy in, one copy out per step
  unsigned long long* h_flag = nullptr;         // pinned: [0] sequence number of the last finished single step, [1] start, [2] end stamp
  unsigned long long step_seq = 0;
  int* warm = nullptr;                          // device: [count, rows...] optimal active set of the last single step
  int* d_status = nullptr;                      // points into d_out
  size_t io_problems = 0;
  // state bounds (Kmpc.m:300-318 / :716-730), single-problem steps only
  int sb_n = 0, sb_kmax = 0, sb_rows = 0;       // n bounded outputs, highest power of A needed, 2 n (Np + 1) rows
  double* sb_lohi = nullptr;                    // [lo (n) | hi (n)] scaled down
  double* sb_Apow = nullptr;                    // (kmax + 1) x [N x N]: A^0 .. A^kmax
  double* sb_A = nullptr;                       // dense constraint matrix of a step: (nrows + sb_rows) x nvar, column-major
  double* sb_b = nullptr;                       // | right-hand sides | row norms | solution x
  int* sb_col = nullptr;                        // ELL column table of a dense matrix: col[k * rows + r] = k
  size_t sb_cap = 0;                            // problems sb_A / sb_b are sized for
  double* sb_work = nullptr;                    // [sb_cap][nvar^2 + nvar + nrows]: H, f, b of every problem of a batch
};

// ---- wave-level helpers (64 lanes): DPP inside 16-lane rows, v_readlane across the 4 rows ----
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ double lane_get(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);                 // every lane: sum of its 16-lane row
  return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
// arg max / arg min over the wave: the extreme VALUE by a DPP reduction (2 moves + 1 v_max per step), then the first lane
// that holds it (ballot + s_ff1) hands over its index.  (Carrying (value, index) pairs through the reduction took ~12
// dependent VALU instructions per step, ~600 cycles per call; this is ~220.)  Ties go to the lowest lane.
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
  double m = v;
  m = fmax(m, dpp_mov<0xB1>(m));
  m = fmax(m, dpp_mov<0x4E>(m));
  m = fmax(m, dpp_mov<0x141>(m));
  m = fmax(m, dpp_mov<0x140>(m));
  const double w = fmax(fmax(lane_get(m, 0), lane_get(m, 16)), fmax(lane_get(m, 32), lane_get(m, 48)));
  const unsigned long long mask = __ballot(v == w);
  idx = mask ? __builtin_amdgcn_readlane(idx, __ffsll((long long)mask) - 1) : 0x7fffffff;
  v = w;
}
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
  double m = v;
  m = fmin(m, dpp_mov<0xB1>(m));
  m = fmin(m, dpp_mov<0x4E>(m));
  m = fmin(m, dpp_mov<0x141>(m));
  m = fmin(m, dpp_mov<0x140>(m));
  const double w = fmin(fmin(lane_get(m, 0), lane_get(m, 16)), fmin(lane_get(m, 32), lane_get(m, 48)));
  const unsigned long long mask = __ballot(v == w);
  idx = mask ? __builtin_amdgcn_readlane(idx, __ffsll((long long)mask) - 1) : 0x7fffffff;
  v = w;
}

// Constraint matrix in ELL form: row r has K slots (val[k*mr + r], col[k*mr + r]); unused slots
// carry val = 0, col = 0.  The MPC rows have <= 3 non-zeros, so A x, H^-1 a_p and N'H^-1 a_p cost
// K operations instead of n.
struct EllMat {
  const double* val;
  const int* col;
  const double* norm;  // row 2-norms
  int K;
};

// LDS scratch of the QP solver (doubles): Hinv n*n | HN n*n | Sinv n*n | x,hp,r,lam,d,zd,ap,f: 8n |
// act: n ints | isact: mr bytes | LDS copy of the constraint rows when K <= QP_KLDS: val mr*K, col mr*K ints, norm mr, b mr
#define QP_KLDS 4
__host__ __device__ inline int qp_lds_doubles(int n, int mr) {
  return 3 * n * n + 9 * n + 2 * ((n + 1) / 2) + (mr + 7) / 8 + 8 + mr * QP_KLDS + (mr * QP_KLDS + 1) / 2 + 2 * mr + 2 +
         64;   // (the last 64: scalar exchanges of the workgroup-wide solver)
}

// Wave-local synchronisation: LDS operations of one wave complete in issue order, so lanes only
// need the compiler not to reorder across this point (usable inside multi-wave workgroups).
#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// min 1/2 x'Hq x + f'x  s.t.  A x <= b.   Hq: n x n column-major (LDS or global).
// Executed by ONE wave (all 64 lanes must call).  Returns 0 on success, 1 on infeasible /
// iteration cap / non-SPD Hessian (x_out = NaN then).
// Warm start (warm_q > 0): the caller has put warm_q active rows into act[], the columns H^-1 a_c into HN and the
// inverse of S = N'H^-1 N into Sinv (layout below).  The multipliers of the equality-constrained minimiser are
// formed; constraints with negative multipliers are released one by one (rank-1 downdates); what remains is a
// valid dual-feasible state from which the usual iteration continues.  The optimum is unique, so the result does
// not depend on the start.  warm_out (global, 1 + n ints): the optimal active set for the next call.
__device__ __forceinline__ int qp_goldfarb_idnani(const double* Hq, const double* f, const EllMat A, const double* bvec, int n, int mr,
                                  double* ws, double* x_out, double tol, long long* stamps = nullptr, bool have_hinv = false,
                                  int hinv_bad = 0, int warm_q = 0, int* warm_out = nullptr) {
  const int lane = threadIdx.x & 63;
  double* Hinv = ws;
  double* HN = Hinv + n * n;
  double* Sinv = HN + n * n;
  double* x = Sinv + n * n;
  double* hp = x + n;
  double* r = hp + n;
  double* lam = r + n;
  double* d = lam + n;
  double* zd = d + n;
  double* ap = zd + n;
  double* fl = ap + n;
  double* apv = fl + n;                      // sparse a_p values (K <= n)
  int* apc = (int*)(apv + n);                // and columns
  int* act = apc + n + (n & 1);
  unsigned char* isact = (unsigned char*)(act + n + (n & 1));
  // constraint rows in LDS (the violation scan of every iteration reads all of them; from global memory a scan costs
  // ~3000 cycles of L2 latency)
  // (isact is 8-byte aligned.  No pointer -> integer -> pointer round trips here: they hide the LDS address space from
  // the compiler, which then emits FLAT loads and stores for the whole workspace instead of ds_read / ds_write)
  double* lval = (double*)isact + ((mr + 7) >> 3);
  double* lnorm = lval + mr * QP_KLDS;
  double* lb = lnorm + mr;
  int* lcol = (int*)(lb + mr);
  const bool ell_lds = A.K <= QP_KLDS;
  const double* Aval = A.val;
  const int* Acol = A.col;
  const double* Anorm = A.norm;
  const double* bv_ = bvec;
  if (ell_lds) {
    for (int e = lane; e < mr * A.K; e += 64) { lval[e] = A.val[e]; lcol[e] = A.col[e]; }
    for (int e = lane; e < mr; e += 64) {          // the LDS copy holds 1 / norm (0 for a null row): no division in the scan
      const double nr_ = A.norm[e];
      lnorm[e] = nr_ == 0.0 ? 0.0 : 1.0 / nr_;
      lb[e] = bvec[e];
    }
    Aval = lval; Acol = lcol; Anorm = lnorm; bv_ = lb;
  }

  // ---- Hinv by in-place Gauss-Jordan (SPD: no pivoting) ----
  if (!have_hinv)
    for (int e = lane; e < n * n; e += 64) Hinv[e] = Hq[e];
  for (int e = lane; e < n; e += 64) {
    fl[e] = f[e];
    ap[e] = 0.0;
  }
  for (int e = lane; e < mr; e += 64) isact[e] = 0;
  WSYNC();
  for (int c = lane; c < warm_q; c += 64) isact[act[c]] = 1;
  WSYNC();
  int bad = hinv_bad;
  for (int k = 0; k < (have_hinv ? 0 : n); ++k) {
    const double piv = Hinv[k + k * n];
    if (!(piv > 0.0)) bad = 1;
    const double ip = 1.0 / piv;
    // each lane owns rows i = lane (n <= 64): read its column-k entry once, then update the row
    const int i = lane;
    double cik = 0.0;
    if (i < n) cik = Hinv[i + k * n];
    WSYNC();
    if (i < n && i != k) {
      const double fct = cik * ip;
#pragma unroll 4
      for (int j = 0; j < n; ++j) {
        if (j != k) Hinv[i + j * n] -= fct * Hinv[k + j * n];
      }
      Hinv[i + k * n] = -fct;
    }
    WSYNC();
    if (i < n && i != k) Hinv[k + i * n] *= ip;   // row k (read by everyone above, scaled after)
    if (lane == 0) Hinv[k + k * n] = ip;
    WSYNC();
  }
  for (int i = lane; i < n; i += 64) {
    double s = 0.0;
#pragma unroll 4
    for (int j = 0; j < n; ++j) s += Hinv[i + j * n] * fl[j];
    x[i] = -s;
  }
  WSYNC();

  if (stamps && lane == 0) stamps[4] = wall_clock64();
#ifdef KP_QP_PROF
  long long qpt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, qlast = clock64();
#define QP_TICK(i) do { long long tn_ = clock64(); qpt[i] += tn_ - qlast; qlast = tn_; } while (0)
#else
#define QP_TICK(i) do { } while (0)
#endif
  int q = 0;
  int status = 1;
  int it = bad ? QP_MAXIT : 0;   // non-SPD Hessian: report failure
  // remove active constraint l: swap with the last entry (symmetric permutation of Sinv, column swap of HN), then one
  // rank-1 downdate of the leading block deletes the LAST index of the inverse Schur complement
  auto drop_active = [&](int l) {
    const int last = q - 1;
    if (l != last) {
      if (lane < q) {                       // columns l <-> last (lane = row)
        const double a_ = Sinv[lane + l * n], b_ = Sinv[lane + last * n];
        Sinv[lane + l * n] = b_;
        Sinv[lane + last * n] = a_;
      }
      WSYNC();
      if (lane < q) {                       // rows l <-> last (lane = column)
        const double a_ = Sinv[l + lane * n], b_ = Sinv[last + lane * n];
        Sinv[l + lane * n] = b_;
        Sinv[last + lane * n] = a_;
      }
      for (int i = lane; i < n; i += 64) {
        const double a_ = HN[i + l * n];
        HN[i + l * n] = HN[i + last * n];
        HN[i + last * n] = a_;
      }
      if (lane == 0) {
        const int ta = act[l]; act[l] = act[last]; act[last] = ta;
        const double tl = lam[l]; lam[l] = lam[last]; lam[last] = tl;
      }
      WSYNC();
    }
    const double isl = 1.0 / Sinv[last + last * n];
    if (lane < last) {
      const double ri = Sinv[lane + last * n] * isl;
      // four columns at a time, every read before the first write: the compiler cannot tell the read-modify-write of
      // column j from the reads of column j + 1 (same LDS array) and would otherwise pay one LDS round trip per column
      int j = 0;
      for (; j + 4 <= last; j += 4) {
        const double a0 = Sinv[lane + j * n], a1 = Sinv[lane + (j + 1) * n], a2 = Sinv[lane + (j + 2) * n], a3 = Sinv[lane + (j + 3) * n];
        const double r0 = Sinv[last + j * n], r1 = Sinv[last + (j + 1) * n], r2 = Sinv[last + (j + 2) * n], r3 = Sinv[last + (j + 3) * n];
        Sinv[lane + j * n] = a0 - ri * r0;
        Sinv[lane + (j + 1) * n] = a1 - ri * r1;
        Sinv[lane + (j + 2) * n] = a2 - ri * r2;
        Sinv[lane + (j + 3) * n] = a3 - ri * r3;
      }
      for (; j < last; ++j) Sinv[lane + j * n] -= ri * Sinv[last + j * n];
    }
    if (lane == 0) isact[act[last]] = 0;
    --q;
    WSYNC();
  };
  if (warm_q > 0 && !bad) {
    q = warm_q;
    int n_rel = 0;
    while (q > 0) {
      // lam = Sinv (N x0 - b)
      for (int c = lane; c < q; c += 64) {
        const int row = act[c];
        double v = -bv_[row];
        for (int k = 0; k < A.K; ++k) v += Aval[k * mr + row] * x[Acol[k * mr + row]];
        d[c] = v;
      }
      WSYNC();
      double lmin = 1e300;
      int l = 0x7fffffff;
      for (int c = lane; c < q; c += 64) {
        double s_ = 0.0;
#pragma unroll 4
        for (int k = 0; k < q; ++k) s_ += Sinv[c + k * n] * d[k];
        lam[c] = s_;
        lmin = s_;
        l = c;
      }
      wave_argmin(lmin, l);
      WSYNC();
      if (!(lmin < 0.0)) break;             // dual feasible
      // Every release is a rank-1 downdate of the inverse Schur complement, and their errors accumulate.  A set that
      // needs more than a few is not the neighbourhood a warm start is for (another reference, another state): give
      // it up - the cold start is exact - instead of iterating on a degraded inverse.
      if (++n_rel > QP_MAX_RELEASE) {
        for (int c = lane; c < q; c += 64) isact[act[c]] = 0;
        q = 0;
        break;
      }
      drop_active(l);
    }
    // x = x0 - H^-1 N lam
    for (int i = lane; i < n; i += 64) {
      double s_ = x[i];
#pragma unroll 4
      for (int c = 0; c < q; ++c) s_ -= HN[i + c * n] * lam[c];
      x[i] = s_;
    }
    WSYNC();
  }
  while (it < QP_MAXIT) {
    ++it;
    // most violated inactive constraint (scaled by the row norm)
    double best = -1e300;
    int bestp = 0x7fffffff;
    int infeas = 0;
    // branch-free, two rows per lane at a time: every LDS read of both rows is issued unconditionally and before any
    // arithmetic (a `continue` or a guarded isact read makes the compiler wait for each read in turn; one row after the
    // other doubles the col -> x[col] dependent chain), the rows are then taken or not by selects
    for (int row0 = lane; row0 < mr; row0 += 128) {
      const int rowA = row0, rowB = row0 + 64;
      const bool inB = rowB < mr;
      const int rB = inB ? rowB : rowA;
      double vA = -bv_[rowA], vB = -bv_[rB];
      const double nA = Anorm[rowA], nB = Anorm[rB];
      const int actA = isact[rowA], actB = isact[rB];
      for (int k = 0; k < A.K; ++k) {
        const int cA = Acol[k * mr + rowA], cB = Acol[k * mr + rB];
        const double aA = Aval[k * mr + rowA], aB = Aval[k * mr + rB];
        vA += aA * x[cA];
        vB += aB * x[cB];
      }
      const double iA = ell_lds ? nA : (nA == 0.0 ? 0.0 : 1.0 / nA);
      const double iB = ell_lds ? nB : (nB == 0.0 ? 0.0 : 1.0 / nB);
      const double sA = vA * iA, sB = vB * iB;
      infeas |= ((iA == 0.0 && vA > tol) || (inB && iB == 0.0 && vB > tol)) ? 1 : 0;
      const bool takeA = iA != 0.0 && !actA && sA > best;
      best = takeA ? sA : best;
      bestp = takeA ? rowA : bestp;
      const bool takeB = inB && iB != 0.0 && !actB && sB > best;
      best = takeB ? sB : best;
      bestp = takeB ? rowB : bestp;
    }
    wave_argmax(best, bestp);
    QP_TICK(0);
    infeas = __any(infeas);
    if (infeas) break;
    if (best <= tol) {
      status = 0;
      break;
    }
    const int p = bestp;
    const double bp = bv_[p];
    // sparse a_p: (col_k, val_k), k < K, staged in LDS (apv/apc) for the loops below; dense copy in ap
    if (lane < A.K) {
      double v = Aval[lane * mr + p];
      int cidx = Acol[lane * mr + p];
      apv[lane] = v;
      apc[lane] = cidx;
      if (v != 0.0) ap[cidx] = v;
    }
    WSYNC();
    double app_l = 0.0;
    for (int i = lane; i < n; i += 64) {
      double s = 0.0;
      for (int k = 0; k < A.K; ++k) s += apv[k] * Hinv[i + apc[k] * n];
      hp[i] = s;
      app_l += s * ap[i];
    }
    const double app = wave_sum(app_l);
    WSYNC();
    QP_TICK(1);
    double lam_p = 0.0;
    bool fail = false;
    while (it < QP_MAXIT) {
      ++it;
      // d = N' Hinv a_p ; r = Sinv d ; zd = hp - HN r
      for (int c = lane; c < q; c += 64) {
        double s = 0.0;
        for (int k = 0; k < A.K; ++k) s += apv[k] * HN[apc[k] + c * n];
        d[c] = s;
      }
      WSYNC();
      QP_TICK(6);
      for (int c = lane; c < q; c += 64) {
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < q; ++k) s += Sinv[c + k * n] * d[k];
        r[c] = s;
      }
      WSYNC();
      QP_TICK(7);
      double apz_l = 0.0, apx_l = 0.0;
      for (int i = lane; i < n; i += 64) {
        double s = hp[i];
#pragma unroll 8
        for (int c = 0; c < q; ++c) s -= HN[i + c * n] * r[c];
        zd[i] = s;
        apz_l += ap[i] * s;
        apx_l += ap[i] * x[i];
      }
      QP_TICK(8);
      const double apz = wave_sum(apz_l);
      const double apx = wave_sum(apx_l);
      QP_TICK(9);
      double t1 = 1e300;
      int l = 0x7fffffff;
      for (int c = lane; c < q; c += 64) {
        if (r[c] > 1e-13) {
          double ratio = lam[c] / r[c];
          if (ratio < t1) {
            t1 = ratio;
            l = c;
          }
        }
      }
      wave_argmin(t1, l);
      QP_TICK(2);
      const bool t2fin = apz > 1e-13 * app;
      const double t2 = t2fin ? (apx - bp) / apz : 1e300;
      const double t = fmin(t1, t2);
      if (!(t < 1e299)) {
        fail = true;
        break;
      }
      WSYNC();
      for (int c = lane; c < q; c += 64) lam[c] -= t * r[c];
      lam_p += t;
      if (t2fin)
        for (int i = lane; i < n; i += 64) x[i] -= t * zd[i];
      WSYNC();
      QP_TICK(3);
      if (t2 <= t1) {
        // add p: Sinv <- bordered inverse with w = r, beta = apz (Schur complement)
        if (q >= n) {
          fail = true;
          break;
        }
        const double ib = 1.0 / apz;
        if (lane < q) {                         // lane i owns row i (q <= n <= 64): no index arithmetic
          const double ri = r[lane] * ib;
          int j = 0;
          for (; j + 4 <= q; j += 4) {           // reads before writes, as in drop_active
            const double a0 = Sinv[lane + j * n], a1 = Sinv[lane + (j + 1) * n], a2 = Sinv[lane + (j + 2) * n], a3 = Sinv[lane + (j + 3) * n];
            const double r0 = r[j], r1 = r[j + 1], r2 = r[j + 2], r3 = r[j + 3];
            Sinv[lane + j * n] = a0 + ri * r0;
            Sinv[lane + (j + 1) * n] = a1 + ri * r1;
            Sinv[lane + (j + 2) * n] = a2 + ri * r2;
            Sinv[lane + (j + 3) * n] = a3 + ri * r3;
          }
          for (; j < q; ++j) Sinv[lane + j * n] += ri * r[j];
        }
        for (int c = lane; c < q; c += 64) {
          Sinv[c + q * n] = -r[c] * ib;
          Sinv[q + c * n] = -r[c] * ib;
        }
        for (int i = lane; i < n; i += 64) HN[i + q * n] = hp[i];
        if (lane == 0) {
          Sinv[q + q * n] = ib;
          act[q] = p;
          lam[q] = lam_p;
          isact[p] = 1;
        }
        ++q;
        WSYNC();
        QP_TICK(4);
        break;
      }
      // partial step: drop active constraint l (swap-with-last rank-1 downdate, no compaction sweeps)
      {
        drop_active(l);
        QP_TICK(5);
      }
    }
    // clear the dense copy of a_p
    if (lane < A.K) ap[apc[lane]] = 0.0;
    WSYNC();
    if (fail) break;
  }
  WSYNC();
  if (stamps && lane == 0) {
    stamps[8] = it;
    stamps[9] = q;
  }
#ifdef KP_QP_PROF
  if (stamps && lane == 0)
    printf("qp prof (cycles): scan+argmax %lld  a_p/hp %lld  ratios %lld  step %lld  add(border) %lld  drop %lld | d %lld r %lld zd %lld sums %lld  it %d q %d\n",
           qpt[0], qpt[1], qpt[2], qpt[3], qpt[4], qpt[5], qpt[6], qpt[7], qpt[8], qpt[9], it, q);
#endif
  for (int i = lane; i < n; i += 64) x_out[i] = status == 0 ? x[i] : __builtin_nan("");
  if (warm_out) {                            // optimal active set: the start of the next call
    if (lane == 0) warm_out[0] = status == 0 ? q : 0;
    for (int c = lane; c < q; c += 64) warm_out[1 + c] = act[c];
  }
  return status;
}

// ---- the same dual active-set iteration by the WHOLE workgroup (256 threads) ----------------------------------------------
// For n <= 32 variables with the constraint rows in LDS (K <= QP_KLDS) and H^-1 already formed by the caller.  One wave
// spends an iteration in dependent chains: a 30-term dot product per lane (r = S^-1 d, z = hp - HN r), 30 columns of the
// bordering update per lane, two rows of the violation scan per lane.  Here every such loop is two-dimensional:
//   * matrix-vector products: 8 adjacent lanes share a row and split the columns (4 terms each), summed by DPP
//     (quad_perm x2 + row_half_mirror);
//   * bordering / removal updates of S^-1: thread (i = tid & 31, j = tid >> 5, +8, ..) owns elements, no chains at all;
//   * violation scan: one row per thread;
// and the few scalars that steer the iteration (most violated row, step lengths) go through LDS: each wave reduces, lane 0
// stores, barrier, every thread combines the four entries in the same order - the control flow is uniform.
// Same algorithm, thresholds and data layout as qp_goldfarb_idnani (the products are summed in another order: results
// agree to rounding).  ~10 barriers per iteration.  red: 64 doubles of LDS for the exchanges.
struct WgX {
  double* v;
  int* i;
};
__device__ __forceinline__ void wgx_argmax(double& v, int& idx, WgX x) {
  wave_argmax(v, idx);
  if ((threadIdx.x & 63) == 0) { x.v[threadIdx.x >> 6] = v; x.i[threadIdx.x >> 6] = idx; }
  __syncthreads();
  double b = x.v[0];
  int bi = x.i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const double c = x.v[w];
    const int ci = x.i[w];
    if (c > b) { b = c; bi = ci; }
  }
  v = b; idx = bi;
}
__device__ __forceinline__ void wgx_argmin(double& v, int& idx, WgX x) {
  wave_argmin(v, idx);
  if ((threadIdx.x & 63) == 0) { x.v[threadIdx.x >> 6] = v; x.i[threadIdx.x >> 6] = idx; }
  __syncthreads();
  double b = x.v[0];
  int bi = x.i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const double c = x.v[w];
    const int ci = x.i[w];
    if (c < b) { b = c; bi = ci; }
  }
  v = b; idx = bi;
}
// two sums at once (a over x.v[0..4), b over x.v[4..8))
__device__ __forceinline__ void wgx_sum2(double& a, double& b, WgX x) {
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { x.v[threadIdx.x >> 6] = a; x.v[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]);
  b = (x.v[4] + x.v[5]) + (x.v[6] + x.v[7]);
}
__device__ __forceinline__ double sum8(double v) {      // over the 8 adjacent lanes of an aligned group, in every lane
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v;
}

__device__ __forceinline__ int qp_gi_wg(const double* f, const EllMat A, const double* bvec, int n, int mr, double* ws, double* red,
                                        double* x_out, double tol, long long* stamps, int hinv_bad, int warm_q, int* warm_out) {
  const int tid = threadIdx.x;
  double* Hinv = ws;
  double* HN = Hinv + n * n;
  double* Sinv = HN + n * n;
  double* x = Sinv + n * n;
  double* hp = x + n;
  double* r = hp + n;
  double* lam = r + n;
  double* d = lam + n;
  double* zd = d + n;
  double* ap = zd + n;
  double* fl = ap + n;
  double* apv = fl + n;
  int* apc = (int*)(apv + n);
  int* act = apc + n + (n & 1);
  unsigned char* isact = (unsigned char*)(act + n + (n & 1));
  double* lval = (double*)isact + ((mr + 7) >> 3);
  double* lnorm = lval + mr * QP_KLDS;
  double* lb = lnorm + mr;
  int* lcol = (int*)(lb + mr);
  // exchange areas (distinct per reduction site: a site is not reached again before every thread has read it)
  WgX xs{red, (int*)(red + 4)}, xr{red + 8, (int*)(red + 12)}, xq{red + 16, (int*)(red + 28)}, xw{red + 32, (int*)(red + 36)};
  const int K = A.K;
  for (int e = tid; e < mr * K; e += 256) { lval[e] = A.val[e]; lcol[e] = A.col[e]; }
  for (int e = tid; e < mr; e += 256) {
    const double nr_ = A.norm[e];
    lnorm[e] = nr_ == 0.0 ? 0.0 : 1.0 / nr_;
    lb[e] = bvec[e];
    isact[e] = 0;
  }
  for (int e = tid; e < n; e += 256) {
    fl[e] = f[e];
    ap[e] = 0.0;
  }
  __syncthreads();
  for (int c = tid; c < warm_q; c += 256) isact[act[c]] = 1;
  // x = -Hinv f : row i = tid >> 3, the 8 lanes of a group split the columns
  const int gi = tid >> 3, gp = tid & 7;
  {
    double s_ = 0.0;
    if (gi < n)
      for (int j = gp; j < n; j += 8) s_ += Hinv[gi + j * n] * fl[j];
    s_ = sum8(s_);
    if (gi < n && gp == 0) x[gi] = -s_;
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[4] = wall_clock64();
  int q = 0;
  int status = 1;
  int it = hinv_bad ? QP_MAXIT : 0;
  // r_out[c] = sum_k M[c + k n] v[k], c < rows, k < cols   (all threads; caller places the barrier)
  auto matvec = [&](const double* M, const double* v, int rows, int cols, double* out) {
    double s_ = 0.0;
    if (gi < rows)
      for (int k = gp; k < cols; k += 8) s_ += M[gi + k * n] * v[k];
    s_ = sum8(s_);
    if (gi < rows && gp == 0) out[gi] = s_;
  };
  // removal of active constraint l: swap with the last one, then the rank-1 downdate that deletes the last index
  auto drop_active = [&](int l) {
    const int last = q - 1;
    if (l != last) {
      if (tid < q) {                                  // columns l <-> last
        const double a_ = Sinv[tid + l * n], b_ = Sinv[tid + last * n];
        Sinv[tid + l * n] = b_;
        Sinv[tid + last * n] = a_;
      }
      if (tid >= 64 && tid < 64 + n) {                // (another wave) columns of HN
        const int i = tid - 64;
        const double a_ = HN[i + l * n];
        HN[i + l * n] = HN[i + last * n];
        HN[i + last * n] = a_;
      }
      if (tid == 128) {
        const int ta = act[l]; act[l] = act[last]; act[last] = ta;
        const double tl = lam[l]; lam[l] = lam[last]; lam[last] = tl;
      }
      __syncthreads();
      if (tid < q) {                                  // rows l <-> last
        const double a_ = Sinv[l + tid * n], b_ = Sinv[last + tid * n];
        Sinv[l + tid * n] = b_;
        Sinv[last + tid * n] = a_;
      }
      __syncthreads();
    }
    const double isl = wg_recip(Sinv[last + last * n]);
    {
      const int i = tid & 31;
      if (i < last) {
        const double ri = Sinv[i + last * n] * isl;
        for (int j = tid >> 5; j < last; j += 8) Sinv[i + j * n] -= ri * Sinv[last + j * n];
      }
    }
    if (tid == 0) isact[act[last]] = 0;
    --q;
    __syncthreads();
  };
  if (warm_q > 0 && !hinv_bad) {
    q = warm_q;
    int n_rel = 0;
    while (q > 0) {
      if (tid < q) {
        const int row = act[tid];
        double v = -lb[row];
        for (int k = 0; k < K; ++k) v += lval[k * mr + row] * x[lcol[k * mr + row]];
        d[tid] = v;
      }
      __syncthreads();
      matvec(Sinv, d, q, q, lam);
      __syncthreads();
      double lmin = tid < q ? lam[tid] : 1e300;
      int l = tid < q ? tid : 0x7fffffff;
      wgx_argmin(lmin, l, xw);
      if (!(lmin < 0.0)) break;
      if (++n_rel > QP_MAX_RELEASE) {
        if (tid < q) isact[act[tid]] = 0;
        q = 0;
        break;
      }
      drop_active(l);
    }
    // x = x0 - HN lam
    {
      double s_ = 0.0;
      if (gi < n)
        for (int c = gp; c < q; c += 8) s_ += HN[gi + c * n] * lam[c];
      s_ = sum8(s_);
      __syncthreads();
      if (gi < n && gp == 0) x[gi] -= s_;
    }
    __syncthreads();
  }
  while (it < QP_MAXIT) {
    ++it;
    // ---- most violated inactive constraint ----
    double best = -1e300;
    int bestp = 0x7fffffff;
    int infeas = 0;
    for (int row = tid; row < mr; row += 256) {
      double v = -lb[row];
      for (int k = 0; k < K; ++k) v += lval[k * mr + row] * x[lcol[k * mr + row]];
      const double in_ = lnorm[row];
      const double sv_ = v * in_;
      infeas |= (in_ == 0.0 && v > tol) ? 1 : 0;
      if (in_ != 0.0 && !isact[row] && sv_ > best) { best = sv_; bestp = row; }
    }
    infeas = __any(infeas);
    if ((tid & 63) == 0) xs.i[4 + (tid >> 6)] = infeas;     // rides on the barrier of the arg max
    wgx_argmax(best, bestp, xs);
    infeas = xs.i[4] | xs.i[5] | xs.i[6] | xs.i[7];
    if (infeas) break;
    if (best <= tol) {
      status = 0;
      break;
    }
    const int p = bestp;
    const double bp = lb[p];
    if (tid < K) {
      const double v = lval[tid * mr + p];
      const int cidx = lcol[tid * mr + p];
      apv[tid] = v;
      apc[tid] = cidx;
      if (v != 0.0) ap[cidx] = v;
    }
    __syncthreads();
    double app = 0.0;
    if (tid < n) {                                   // (n <= 32: wave 0 alone holds the terms of a_p'H^-1 a_p)
      double s_ = 0.0;
      for (int k = 0; k < K; ++k) s_ += apv[k] * Hinv[tid + apc[k] * n];
      hp[tid] = s_;
      app = s_ * ap[tid];
    }
    if (tid < 64) {
      app = wave_sum(app);
      if (tid == 0) xq.v[0] = app;
    }
    __syncthreads();                                 // (also publishes hp)
    app = xq.v[0];
    double lam_p = 0.0;
    bool fail = false;
    while (it < QP_MAXIT) {
      ++it;
      if (tid < q) {
        double s_ = 0.0;
        for (int k = 0; k < K; ++k) s_ += apv[k] * HN[apc[k] + tid * n];
        d[tid] = s_;
      }
      __syncthreads();
      matvec(Sinv, d, q, q, r);
      __syncthreads();
      // zd = hp - HN r ; a_p'zd ; a_p'x
      double apz = 0.0, apx = 0.0;
      {
        double s_ = 0.0;
        if (gi < n)
          for (int c = gp; c < q; c += 8) s_ += HN[gi + c * n] * r[c];
        s_ = sum8(s_);
        if (gi < n && gp == 0) {
          const double z_ = hp[gi] - s_;
          zd[gi] = z_;
          apz = ap[gi] * z_;
          apx = ap[gi] * x[gi];
        }
      }
      wgx_sum2(apz, apx, xr);                       // (publishes zd)
      double t1 = 1e300;
      int l = 0x7fffffff;
      if (tid < q && r[tid] > 1e-13) {
        t1 = lam[tid] * wg_recip(r[tid]);
        l = tid;
      }
      wgx_argmin(t1, l, xw);
      const bool t2fin = apz > 1e-13 * app;
      const double t2 = t2fin ? (apx - bp) * wg_recip(apz) : 1e300;
      const double t = fmin(t1, t2);
      if (!(t < 1e299)) {
        fail = true;
        break;
      }
      if (tid < q) lam[tid] -= t * r[tid];
      lam_p += t;
      if (t2fin && tid >= 64 && tid < 64 + n) x[tid - 64] -= t * zd[tid - 64];
      __syncthreads();
      if (t2 <= t1) {
        if (q >= n) {
          fail = true;
          break;
        }
        const double ib = wg_recip(apz);
        {
          const int i = tid & 31;
          if (i < q) {
            const double ri = r[i] * ib;
            for (int j = tid >> 5; j < q; j += 8) Sinv[i + j * n] += ri * r[j];
          }
        }
        if (tid < q) {
          Sinv[tid + q * n] = -r[tid] * ib;
          Sinv[q + tid * n] = -r[tid] * ib;
        }
        if (tid >= 64 && tid < 64 + n) HN[tid - 64 + q * n] = hp[tid - 64];
        if (tid == 128) {
          Sinv[q + q * n] = ib;
          act[q] = p;
          lam[q] = lam_p;
          isact[p] = 1;
        }
        ++q;
        __syncthreads();
        break;
      }
      drop_active(l);
    }
    if (tid < K) ap[apc[tid]] = 0.0;
    __syncthreads();
    if (fail) break;
  }
  if (stamps && tid == 0) {
    stamps[8] = it;
    stamps[9] = q;
  }
  for (int i = tid; i < n; i += 256) x_out[i] = status == 0 ? x[i] : __builtin_nan("");
  if (warm_out) {
    if (tid == 0) warm_out[0] = status == 0 ? q : 0;
    for (int c = tid; c < q; c += 256) warm_out[1 + c] = act[c];
  }
  return status;
}

// ---- generic QP shim kernel -----------------------------------------------------------------
// One wave per problem.  Problem p = blockIdx.x reads H + p sH, f + p sf, the ELL values / row norms + p sA / p sb and
// b + p sb, writes x + p n and status[p] (strides in doubles; 0 for a single problem).  sticky: a problem whose status
// word already carries a failure (an earlier linearisation pass) is left alone.
__global__ __launch_bounds__(64) void kp_qp_kernel(const double* H, const double* f, EllMat A, const double* b, int n, int mr,
                                                   double* x, int* status, size_t sH, size_t sf, size_t sA, size_t sb, int sticky) {
  extern __shared__ double sm[];
  const size_t p = blockIdx.x;
  if (sticky && status[p] != KP_OK) return;
  EllMat Ap = A;
  Ap.val = A.val + p * sA;
  Ap.norm = A.norm + p * sb;
  int st = qp_goldfarb_idnani(H + p * sH, f + p * sf, Ap, b + p * sb, n, mr, sm, x + p * (size_t)n, 1e-10);
  if (threadIdx.x == 0) status[p] = st ? KP_ERR_QP_FAIL : KP_OK;
}

// dense (mr x n, column-major) -> ELL on the host
static void to_ell(const double* A, int mr, int n, std::vector<double>& val, std::vector<int>& col, std::vector<double>& norm, int& K) {
  K = 1;
  for (int r = 0; r < mr; ++r) {
    int c = 0;
    for (int j = 0; j < n; ++j) c += A[r + (size_t)j * mr] != 0.0;
    K = std::max(K, c);
  }
  val.assign((size_t)K * mr, 0.0);
  col.assign((size_t)K * mr, 0);
  norm.assign(mr, 0.0);
  for (int r = 0; r < mr; ++r) {
    int k = 0;
    double s = 0.0;
    for (int j = 0; j < n; ++j) {
      double v = A[r + (size_t)j * mr];
      if (v != 0.0) {
        val[(size_t)k * mr + r] = v;
        col[(size_t)k * mr + r] = j;
        s += v * v;
        ++k;
      }
    }
    norm[r] = std::sqrt(s);
  }
}

extern "C" int kp_qp_solve(kp_ctx* ctx, const double* H, const double* f, const double* A, const double* b, int n, int mr,
                           double* x, int* status) {
  if (!ctx || !H || !f || !x || n < 1 || n > QP_MAXN || mr < 0 || (mr > 0 && (!A || !b)))
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_qp_solve: bad argument (n <= 64)") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<double> val, norm;
  std::vector<int> col;
  int K = 1;
  to_ell(A, mr, n, val, col, norm, K);
  size_t bH = (size_t)n * n * 8, bV = val.size() * 8, bCo = col.size() * 4;
  char* ws = (char*)ctx->workspace(7, bH + bV + bCo + (size_t)(2 * n + 2 * mr) * 8 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_qp_solve: out of device memory");
  double* dH = (double*)ws;
  double* dV = (double*)(ws + bH);
  double* df = (double*)(ws + bH + bV);
  double* dx = df + n;
  double* db = dx + n;
  double* dn = db + mr;
  int* dC = (int*)(dn + mr);
  int* dst = dC + col.size();
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dH, H, bH, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(df, f, (size_t)n * 8, hipMemcpyHostToDevice, s));
  if (mr) {
    KP_HIP(ctx, hipMemcpyAsync(dV, val.data(), bV, hipMemcpyHostToDevice, s));
    KP_HIP(ctx, hipMemcpyAsync(dC, col.data(), bCo, hipMemcpyHostToDevice, s));
    KP_HIP(ctx, hipMemcpyAsync(db, b, (size_t)mr * 8, hipMemcpyHostToDevice, s));
    KP_HIP(ctx, hipMemcpyAsync(dn, norm.data(), (size_t)mr * 8, hipMemcpyHostToDevice, s));
  }
  EllMat E{dV, dC, dn, K};
  size_t lds = (size_t)qp_lds_doubles(n, mr) * 8;
  if (lds > 160 * 1024) return ctx->fail(KP_ERR_ARG, "kp_qp_solve: problem too large");
  static KpLdsCache qp_lds;
  KP_HIP(ctx, kp_ensure_lds(qp_lds, (const void*)kp_qp_kernel, lds));
  hipLaunchKernelGGL(kp_qp_kernel, dim3(1), dim3(64), lds, s, dH, df, E, db, n, mr, dx, dst, (size_t)0, (size_t)0, (size_t)0, (size_t)0, 0);
  KP_HIP(ctx, hipGetLastError());
  int st = 0;
  KP_HIP(ctx, hipMemcpyAsync(x, dx, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipMemcpyAsync(&st, dst, sizeof(int), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));   // also keeps the host vectors alive until the copies are done
  if (status) *status = st;
  return KP_OK;
}

// ---- MPC setup ---------------------------------------------------------------------------------
// P_0 = proj, P_{k+1} = P_k A   (Ahat rows of Kmpc.m:168-172 / 528-532 projected by Chat :193,540);
// linear model: S_k = P_k B.
__global__ __launch_bounds__(256) void kp_mpc_setup_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                           const double* __restrict__ proj, int N, int m, int Np, int nproj,
                                                           int linear, double* __restrict__ P, double* __restrict__ S0,
                                                           double* __restrict__ PB) {
  const int tid = threadIdx.x;
  for (int e = tid; e < nproj * N; e += 256) P[e] = proj[e];
  __syncthreads();
  for (int k = 0; k < Np; ++k) {
    const double* Pk = P + (size_t)k * nproj * N;
    double* Pn = P + (size_t)(k + 1) * nproj * N;
    for (int e = tid; e < nproj * N; e += 256) {
      int p = e % nproj, c = e / nproj;
      double s = 0.0;
      for (int j = 0; j < N; ++j) s += Pk[p + j * nproj] * A[j + (size_t)c * N];
      Pn[e] = s;
    }
    __syncthreads();
  }
  if (linear) {
    for (int e = tid; e < Np * nproj * m; e += 256) {
      int k = e / (nproj * m), rem = e % (nproj * m), p = rem % nproj, a = rem / nproj;
      const double* Pk = P + (size_t)k * nproj * N;
      double s = 0.0;
      for (int j = 0; j < N; ++j) s += Pk[p + j * nproj] * B[j + (size_t)a * N];
      S0[e] = s;
    }
  } else {
    // bilinear: S_k = P_k Beta(z) with Beta(z) = [B_1 z ... B_m z] (Kmpc.m:578-585), i.e. entry (p, i) of S_k is row p of
    // P_k B_i times z.  Those rows are constants of the model: Np m nproj N doubles (40 KB for the bench's N = 84) instead of
    // the N x N m of B (169 KB) that a step would otherwise stream through its one CU to form Beta(z) first.
    for (int e = tid; e < Np * nproj * m * N; e += 256) {
      const int j = e % N, row = e / N;
      const int k = row / (nproj * m), rem = row % (nproj * m), p = rem % nproj, i = rem / nproj;
      const double* Pk = P + (size_t)k * nproj * N;
      const double* Bi = B + (size_t)i * N * N + (size_t)j * N;
      double s = 0.0;
      for (int c = 0; c < N; ++c) s += Pk[p + c * nproj] * Bi[c];
      PB[e] = s;
    }
  }
}


struct MpcArgs {
  BasisDev basis;   // used when zeta != nullptr (fused lift)
  int model_type, N, m, Np, nproj, nvar, nrows, iters, has_basis;
  double q_run, q_term;
  const double *A, *B, *P, *S0, *PB, *r, *Aq, *bq0, *Anorm;
  EllMat ell;
  int* warm;            // [1 + nvar] active set of the previous single-problem step (nullptr: cold start)
  int alias;            // batched launches: z | beta | S | e live inside the solver's scratch (dead before it is first written)
  int assemble_only;    // stop after the QP data have been exported (state-bound steps solve with the generic QP kernel)
  const double* U_lin;  // [nb][nvar] or nullptr: re-linearise along the lifted horizon of THESE inputs (pass >= 2 of a
                        // state-bound step, Kmpc.m:890-895) instead of starting from z
  const double* z;      // [nb][N]      (or nullptr with zeta)
  const double* zeta;   // [nb][nzeta]  (fused lift)
  const double* u_prev; // [nb][m]
  const double* Yr;     // [nb][nproj*(Np+1)]
  double* U;            // [nb][nvar]   x = vec(U') : u_0 (m), u_1 (m), ...
  double* z_out;        // [nb][N] or nullptr
  double* qp_export;    // [nb][nvar*nvar + nvar + nrows] or nullptr
  int* status;          // [nb]
  long long* stamps;    // [8] wall_clock64 stamps + [8] counters of problem 0 (diagnostics), or nullptr
  unsigned long long* done_flag;   // pinned host word: the kernel stores done_seq there when its outputs are visible (or nullptr)
  unsigned long long done_seq;
  int qp_wg;            // active-set iteration by the whole workgroup (qp_gi_wg) when the problem qualifies
  int stage_off;        // single-problem launches: P | PB are copied to LDS at this offset (doubles) at kernel start; 0: read from L2
  int stage_doubles;    // multiple of 512 (every wave issues the same number of 1 KB LDS-DMA loads)
};

// LDS (doubles): z N | beta N*m | S Np*nproj*m | e (Np+1)*nproj | Hq nvar^2 | f nvar | b nrows | zh (Np+1)*N (iters>1)
//                | full nfull (fused lift) | qp scratch
__host__ __device__ inline int mpc_lds_doubles(int N, int m, int Np, int nproj, int nvar, int nrows, int iters, int nfull) {
  return N + N * m + Np * nproj * m + (Np + 1) * nproj + nvar * nvar + nvar + nrows + (iters > 1 ? (Np + 1) * N : 0) + nfull + 4 +
         qp_lds_doubles(nvar, nrows);
}

// WARM: single-problem instantiation with the active-set warm start; the batched one carries none of that code
// (its registers and branches cost the batch 1.8x when they were a run-time option).
template <bool WARM>
__device__ __forceinline__ void mpc_step_body(const MpcArgs& a) {
  extern __shared__ __align__(16) double sm[];
  const int tid = threadIdx.x;
  const int pb = blockIdx.x;
  const int N = a.N, m = a.m, Np = a.Np, nproj = a.nproj, nv = a.nvar, nr = a.nrows;
  // a.alias (batched launches, one linearisation pass): the assembly's inputs z | beta | S | e are dead once Hq and f
  // exist, which is before the solver's scratch is first written - they live in its second n x n block, and a problem
  // takes 40.7 KB instead of 44.6: FOUR workgroups per CU instead of three.
  const int n_asm = N + N * m + Np * nproj * m + (Np + 1) * nproj;
  double* Hq = sm + (a.alias ? 0 : n_asm);   // nv x nv  (= 2H)
  double* f = Hq + nv * nv;
  double* bq = f + nv;
  double* zh = bq + nr;                  // (Np+1) x N   lifted horizon (iters > 1)
  double* full = zh + ((a.iters > 1 || a.U_lin) ? (Np + 1) * N : 0);
  int* st_sh = (int*)(full + (a.has_basis ? a.basis.nfull : 0));   // one slot for the QP status
  double* qpws = (double*)st_sh + 1;
  qpws += (qpws - sm) & 1;                // 16-byte aligned (sm is), by index arithmetic: the pointer stays an LDS pointer
  double* z = a.alias ? qpws + nv * nv : sm;
  double* beta = z + N;                  // N x m  (Beta(z) = B kron(I,z), Ksysid.m:1288-1289)
  double* S = beta + N * m;              // [Np][nproj x m]
  double* ev = S + Np * nproj * m;       // [(Np+1)][nproj]   P_i z - Yr_i
  const double* Yr = a.Yr + (size_t)pb * nproj * (Np + 1);
  const double* up = a.u_prev + (size_t)pb * m;
  long long* stamps = (a.stamps && pb == 0) ? a.stamps : nullptr;
  if (stamps && tid == 0) stamps[0] = wall_clock64();

  // The step's inputs lie in page-locked HOST memory in single-problem launches (zero-copy: no copy command on the latency
  // path of a closed loop), so every access is a PCIe round trip of ~2.3 us.  u_prev, Yr and zeta / z are requested HERE,
  // before anything else that has to be waited for - one round trip instead of three behind each other (dictionary ->
  // right-hand sides -> tracking error) - and go straight to the LDS arrays that are written for good only later: u_prev
  // to f, Yr to e (which becomes P z - Yr in place), zeta (or the lifted state itself) to z, which the dictionary evaluation
  // reads and only the step after it overwrites.  By LDS-DMA (4-byte pieces: any alignment of the caller's block), one array
  // per wave: no register waits for them.  (Measured and undone: everything that consumes them moved into the first pass of
  // the linearisation loop, so that the loop's hoisted address arithmetic - ~2 us of scalar instructions after the tracking
  // error - would run while the words are in flight: the hoisted block grew to ~1 500 instructions, longer than the round
  // trip it was to hide behind, 32.6 against 31.7 us.)
  const double* zin = a.has_basis ? a.zeta + (size_t)pb * a.basis.nzeta : a.z + (size_t)pb * N;
  const int nzin = a.has_basis ? a.basis.nzeta : N, nyr = (Np + 1) * nproj;
  {
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(1))) const char glb_char;
    const int w_ = tid >> 6, l4 = (tid & 63) * 4;
    const double* src = w_ == 0 ? zin : w_ == 1 ? Yr : up;
    double* dst = w_ == 0 ? z : w_ == 1 ? ev : f;
    const int bytes = w_ == 0 ? nzin * 8 : w_ == 1 ? nyr * 8 : w_ == 2 ? m * 8 : 0;
    for (int o = 0; o < bytes; o += 256)
      if (o + l4 < bytes) __builtin_amdgcn_global_load_lds((glb_char*)src + o + l4, (lds_char*)dst + o, 4, 0, 0);
  }
  const double r_diag = tid < nv ? a.r[tid % m] : 0.0;      // diagonal of R for the Hessian (element tid of its first round)
  const double r_bq0 = tid < nr ? a.bq0[tid] : 0.0;         // constant right-hand sides (first round of the loop below)
  ColDesc cd0 = {};
  const bool pre0 = a.has_basis && tid < a.basis.nfull;
  if (pre0) cd0 = a.basis.cols[tid];
  // warm start: the previous step's active set - its size and row ids are requested here, the rows' (sparse) entries, whose
  // addresses need the ids, behind the barrier that publishes the host words (the ids have arrived by then): both round trips hide behind
  // the host words and the lift, and no wait for an ordinary load stands between the LDS-DMA requests and that barrier (hipcc
  // drains the whole vector-memory counter at such a wait while an LDS-DMA is in flight)
  int warm_q0 = 0, warm_row = 0;
  double warm_val[QP_KLDS];
  int warm_col[QP_KLDS];
#pragma unroll
  for (int k = 0; k < QP_KLDS; ++k) { warm_val[k] = 0.0; warm_col[k] = 0; }
  if (WARM && a.warm && tid < nv) warm_row = a.warm[1 + tid];      // (the set's size is wave-uniform: the compiler would wait for it at once)
  {
    // P | PB (constants of the model that every step reads once, 55 KB at N = 84) go to LDS by LDS-DMA while the host words
    // are on their way: no registers, nothing to wait for before the barrier below (hipcc drains vmcnt there)
    if (a.stage_off) {
      typedef __attribute__((address_space(3))) char lds_char;
      typedef __attribute__((address_space(1))) const char glb_char;
      lds_char* dst = (lds_char*)(sm + a.stage_off);
      glb_char* src = (glb_char*)a.P;
      const int wbase = (tid & ~63) * 16, lane16 = (tid & 63) * 16;
      for (int o = wbase; o < a.stage_doubles * 8; o += 256 * 16)       // wave-uniform LDS base; each lane its own 16 source bytes
        __builtin_amdgcn_global_load_lds(src + o + lane16, dst + o, 16, 0, 0);
    }
  }
  int status = 0;
  // lifted horizon along an input sequence x = [u_0; u_1; ...] (Kmpc.m:891-895)
  auto lifted_horizon = [&](const double* x) {
    if (tid < 64) {
      for (int c = tid; c < N; c += 64) zh[c] = z[c];
    }
    __syncthreads();
    for (int j = 0; j < Np; ++j) {
      const double* zj = zh + (size_t)j * N;
      for (int e = tid; e < N; e += 256) {
        double s = 0.0;
        for (int c = 0; c < N; ++c) s += a.A[e + (size_t)c * N] * zj[c];
        for (int i = 0; i < m; ++i) {
          const double* Bi = a.B + (size_t)i * N * N;
          double t = 0.0;
          for (int c = 0; c < N; ++c) t += Bi[e + (size_t)c * N] * zj[c];
          s += t * x[j * m + i];
        }
        zh[(size_t)(j + 1) * N + e] = s;
      }
      __syncthreads();
    }
  };
    // every LDS-DMA above (global_load ... lds) must have landed before another wave reads those bytes: that is the vector-memory
    // counter, which a workgroup-scope __syncthreads() is NOT obliged to wait for (one hipcc build lowered it to
    // `s_waitcnt lgkmcnt(0); s_barrier`) - so the wait is written out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (WARM && a.warm) warm_q0 = a.warm[0];
    if (WARM && a.warm && a.ell.K <= QP_KLDS) {
#pragma unroll
      for (int k = 0; k < QP_KLDS; ++k) {
        warm_val[k] = (tid < nv && k < a.ell.K) ? a.ell.val[k * nr + warm_row] : 0.0;
        warm_col[k] = (tid < nv && k < a.ell.K) ? a.ell.col[k * nr + warm_row] : 0;
      }
    }
    if (stamps && tid == 0) stamps[10] = wall_clock64();      // host words and staged constants have landed
  // ---- lifted state (Kmpc.m:842) ----
  if (a.has_basis) {
    const BasisDev& b = a.basis;
    const double* zeta = z;
    for (int c = tid; c < b.nfull; c += 256) {
      const ColDesc cd = c == tid ? cd0 : b.cols[c];
      double v;
      if (cd.kind == COL_MONO && b.nvars <= 8) {
        // the exponent bytes come with the descriptor (kp_eval_col fetches them one by one between branches: a dependent L2
        // round trip per variable, 6.7 us for this phase at 6 variables)
        int ex[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ex[i] = ((uint32_t)(i < 4 ? cd.aux : cd.pad) >> (8 * (i & 3))) & 0xff;
        v = 1.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double x = zeta[i < b.nvars ? i : 0];
          for (int k = 0; k < ex[i]; ++k) v *= x;
        }
      } else {
        v = kp_eval_col(b, cd, zeta, 1);
      }
      full[c] = v;
    }
    wg_lds_barrier();
    for (int c = tid; c < N; c += 256) {
      double v;
      if (b.k_pcs == 0)
        v = full[c];
      else if (c < b.nvars)
        v = zeta[c];
      else if (c < b.nvars + b.k_pcs) {
        const double* pc = b.pcs + (size_t)(c - b.nvars) * b.nfull;
        v = 0.0;
        for (int i = 0; i < b.nfull; ++i) v += pc[i] * full[i];
      } else
        v = 1.0;
      z[c] = v;
    }
  }
  wg_lds_barrier();
  if (stamps && tid == 0) stamps[11] = wall_clock64();        // lifted state
  if (a.z_out)
    for (int c = tid; c < N; c += 256) a.z_out[(size_t)pb * N + c] = z[c];

  // constraint right-hand side: b = c (E = 0 without state bounds, Kmpc.m:862) and the
  // "memory" rows +-u_0 <= +-u_prev (Kmpc.m:865-870)
  for (int e = tid; e < nr; e += 256) {
    double v = e == tid ? r_bq0 : a.bq0[e];
    int k = e - (nr - 2 * m);
    if (k >= 0) v = k < m ? f[k] : -f[k - m];
    bq[e] = v;
  }
  // e_i = P_i z - Yr_i
  auto track_err = [&](auto Pp) {          // (Pp: P in global memory or its LDS copy - two instantiations, no generic pointer)
    for (int e4 = tid; e4 < (Np + 1) * nproj * 4; e4 += 256) {   // each dot product split over 4 lanes
      int e = e4 >> 2, part = e4 & 3;
      int i = e / nproj, p = e % nproj;
      auto Pi = Pp + (size_t)i * nproj * N;
      double s = 0.0;
#pragma unroll 4
      for (int j = part; j < N; j += 4) s += Pi[p + j * nproj] * z[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (part == 0) ev[e] = s - ev[e];
    }
  };
  if (a.stage_off) track_err(sm + a.stage_off);
  else track_err(a.P);
  if (stamps && tid == 0) stamps[1] = wall_clock64();
  int iter0 = 0;
  if (a.U_lin) {                         // (uniform) a later pass of a state-bound step: linearise along the previous solution
    __syncthreads();
    lifted_horizon(a.U_lin + (size_t)pb * nv);
    iter0 = 1;
  }
  for (int iter = iter0; iter < iter0 + a.iters; ++iter) {
    // ---- S_k = P_k * Beta(z_k)  (get_costB_bilinear, Kmpc.m:578-585: block i uses z(i,:) when a
    // horizon of lifted states is given, else z) ----
    if (stamps && tid == 0) stamps[6] = wall_clock64();        // (what precedes is the loop's hoisted address arithmetic)
    if (a.model_type == KP_MODEL_BILINEAR) {
      for (int k = 0; k < Np; ++k) {
        const double* zk = iter == 0 ? z : zh + (size_t)k * N;
        if (iter == 0 && k > 0) break;   // same z for every block: one pass over all of them
        int k0 = k, k1 = iter == 0 ? Np : k + 1;
        auto s_rows = [&](auto PBp) {
          for (int e4 = tid; e4 < (k1 - k0) * nproj * m * 4; e4 += 256) {
            const int e = k0 * nproj * m + (e4 >> 2), part = e4 & 3;
            auto row = PBp + (size_t)e * N;        // row p of P_k B_i, contiguous: four lanes share a dot product
            double s = 0.0;
#pragma unroll 8
            for (int j = part; j < N; j += 4) s += row[j] * zk[j];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            if (part == 0) S[e] = s;
          }
        };
        if (a.stage_off) s_rows(sm + a.stage_off + (a.PB - a.P));
        else s_rows(a.PB);
      }
    } else {
      for (int e = tid; e < Np * nproj * m; e += 256) S[e] = a.S0[e];
    }
    wg_lds_barrier();
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    // ---- Hq = 2 (CB'Q CB + R),  f = 2 CB'Q e   (Kmpc.m:604,879,883) ----
    // With Sall = [S_0 S_1 ... S_{Np-1}] (nproj x nv: block k, input a is column k m + a) the blocks of CB_i are columns of Sall,
    // so every term of H is an entry of the Gram matrix T = Sall' Sall, and H(r1, r2) walks down one diagonal of it:
    //   H(j1 m + a1, j2 m + a2) = 2 sum_{t < Np - j2} q_{j2 + 1 + t} T((j2 - j1 + t) m + a1, t m + a2)        (j2 >= j1).
    // Stage 1 forms T once (nv (nv / 2 + 1) short dot products, into the solver's scratch, which is free until H^-1), stage 2
    // is one LDS read and one multiply-add per term - the direct form recomputed every T entry up to Np times inside a
    // doubly nested loop of dependent LDS round trips (4.5 us of a 34 us step).  Same terms in the same order: same bits.
    // Both stages use symmetry: the pairs {r, (r + d) mod nv}, d = 0 .. nv / 2, cover the upper triangle (the last d twice
    // when nv is even: the same value stored twice); small quotients by a float reciprocal with one correction (a 32-bit
    // integer division is ~35 instructions).
    {
      const float rnv = 1.0f / (float)nv, rmf = 1.0f / (float)m;
      auto divmod = [](int x, int d, float rd, int& q, int& r) {
        q = (int)((float)x * rd);
        r = x - q * d;
        if (r < 0) { r += d; --q; }
        if (r >= d) { r -= d; ++q; }
      };
      double* Tm = qpws;
      // The loops below fetch the operands of FOUR terms (and of all nproj products of a term) before they use the first: a
      // loop that reads, waits, adds and branches per term is a chain of LDS round trips (~130 cycles each, ten per element).
      // Indices past the end are clamped (the reads stay inside the arrays), the sums take only the valid terms, in order.
      auto stage1 = [&](auto NPc) {
        constexpr int NP = decltype(NPc)::value;          // nproj (0: any)
        for (int e = tid; e < nv * (nv / 2 + 1); e += 256) {
          int d, i0;
          divmod(e, nv, rnv, d, i0);
          int jx = i0 + d;
          if (jx >= nv) jx -= nv;
          const int c1 = min(i0, jx), c2 = max(i0, jx);
          const double* S1 = S + c1 * nproj;
          const double* S2 = S + c2 * nproj;
          double t = 0.0;
          if constexpr (NP > 0) {
            double x1[NP], x2[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { x1[p] = S1[p]; x2[p] = S2[p]; }
#pragma unroll
            for (int p = 0; p < NP; ++p) t += x1[p] * x2[p];
          } else {
            for (int p = 0; p < nproj; ++p) t += S1[p] * S2[p];
          }
          Tm[c1 + c2 * nv] = t;
          Tm[c2 + c1 * nv] = t;
        }
        // f = 2 CB'Q e by the LAST threads (they have the fewest elements of the two H rounds)
        for (int e = 255 - tid; e < nv; e += 256) {
          int j, a1;
          divmod(e, m, rmf, j, a1);
          double s = 0.0;
          if constexpr (NP > 0) {
            const int cnt = Np - j, last = cnt - 1;
            const double* S0 = S + a1 * nproj;               // column a1 of S_0; term t uses S_t and e_{j + 1 + t}
            const double* e0 = ev + (j + 1) * nproj;
            for (int t0 = 0; t0 < cnt; t0 += 4) {
              double xs[4][NP], xe[4][NP];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int tt = min(t0 + u, last);
#pragma unroll
                for (int p = 0; p < NP; ++p) { xs[u][p] = S0[tt * nproj * m + p]; xe[u][p] = e0[tt * nproj + p]; }
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                double t = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) t += xs[u][p] * xe[u][p];
                if (t0 + u < last) s += a.q_run * t;
                else if (t0 + u == last) s += a.q_term * t;
              }
            }
          } else {
            for (int i = j + 1; i <= Np; ++i) {
              double qi = i == Np ? a.q_term : a.q_run;
              const double* S1 = S + (i - j - 1) * nproj * m + a1 * nproj;
              double t = 0.0;
              for (int p = 0; p < nproj; ++p) t += S1[p] * ev[i * nproj + p];
              s += qi * t;
            }
          }
          f[e] = 2.0 * s;
        }
      };
      switch (nproj) {
        case 1: stage1(std::integral_constant<int, 1>()); break;
        case 2: stage1(std::integral_constant<int, 2>()); break;
        case 3: stage1(std::integral_constant<int, 3>()); break;
        case 4: stage1(std::integral_constant<int, 4>()); break;
        default: stage1(std::integral_constant<int, 0>()); break;
      }
      wg_lds_barrier();
      const int dstride = m * (nv + 1);
      for (int e = tid; e < nv * (nv / 2 + 1); e += 256) {
        int d, i0;
        divmod(e, nv, rnv, d, i0);
        int jx = i0 + d;
        if (jx >= nv) jx -= nv;
        const int r1 = min(i0, jx), r2 = max(i0, jx);
        int j1, a1, j2, a2;
        divmod(r1, m, rmf, j1, a1);
        divmod(r2, m, rmf, j2, a2);
        const double* tp = Tm + ((j2 - j1) * m + a1) + a2 * nv;      // T(S_{i-j1-1} column a1, S_{i-j2-1} column a2) at i = j2 + 1
        const int cnt = Np - j2, last = cnt - 1;
        double s = 0.0;
        for (int t0 = 0; t0 < cnt; t0 += 4) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = tp[min(t0 + u, last) * dstride];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (t0 + u < last) s += a.q_run * v[u];
            else if (t0 + u == last) s += a.q_term * v[u];
          }
        }
        if (r1 == r2) s += (e == tid && tid < nv) ? r_diag : a.r[a1];
        Hq[r1 + r2 * nv] = 2.0 * s;
        Hq[r2 + r1 * nv] = 2.0 * s;
      }
    }
    wg_lds_barrier();
    if (a.qp_export) {
      double* ex = a.qp_export + (size_t)pb * (nv * nv + nv + nr);
      for (int e = tid; e < nv * nv; e += 256) ex[e] = Hq[e];
      for (int e = tid; e < nv; e += 256) ex[nv * nv + e] = f[e];
      for (int e = tid; e < nr; e += 256) ex[nv * nv + nv + e] = bq[e];
    }
    if (a.assemble_only) return;                 // (uniform) H, f, b are in qp_export, z in z_out
    // ---- QP by wave 0 ----
    if (stamps && tid == 0) stamps[3] = wall_clock64();
    double* xout = a.U + (size_t)pb * nv;
    // H^-1 by the whole workgroup into the solver's scratch (first n*n doubles).  Hq is not needed after this
    // point (the solver works from H^-1; the next linearisation pass rebuilds it), so it is the second buffer of the
    // ping-pong inverse.
    // (the solver's own single-wave inverse, have_hinv = false, measured 63 us against 38 us for the first workgroup
    //  version, 27 us for the two-barrier in-place one and 10 us for this one, nv = 30)
    const int hbad = wg_spd_inverse_pp(qpws, Hq, nv, nv, true);       // (the pair sweep reads Hq where it lies)
    if (stamps && tid == 0) stamps[12] = wall_clock64();      // H^-1
    const bool have_hinv = true;
    // warm start from the previous step's active set (closed loops change it by a few rows per step): the whole
    // workgroup forms HN = H^-1 N' and S = N H^-1 N' and inverts S; wave 0 then only has to release rows whose
    // multiplier is negative and add the newly violated ones
    int wq = 0;
    if (WARM && a.warm && !hbad) {
      if (iter != iter0) {                     // later linearisation passes start from the set the previous pass left
        warm_q0 = a.warm[0];
        if (tid < nv) warm_row = a.warm[1 + tid];
        if (a.ell.K <= QP_KLDS) {
#pragma unroll
          for (int k = 0; k < QP_KLDS; ++k) {
            warm_val[k] = (tid < nv && k < a.ell.K) ? a.ell.val[k * nr + warm_row] : 0.0;
            warm_col[k] = (tid < nv && k < a.ell.K) ? a.ell.col[k * nr + warm_row] : 0;
          }
        }
      }
      wq = min(warm_q0, nv);
      double* HNw = qpws + nv * nv;
      double* Sw = qpws + 2 * nv * nv;
      int* actw = (int*)(qpws + 3 * nv * nv + 9 * nv) + nv + (nv & 1);     // the solver's act[] (behind apv and apc)
      if (wq > 0) {
        const int K = a.ell.K;
        if (K <= QP_KLDS && 3 * QP_KLDS <= 2 * nv) {          // (the staged rows take 1.5 QP_KLDS nv doubles of Hq's nv^2)
          // rows staged in LDS (Hq is free between the two inverses): val [wq][4] | col [wq][4]; no global access and
          // no integer division in the two products (lane = row of the result, waves stride the columns)
          double* sv = Hq;
          int* sc = (int*)(Hq + QP_KLDS * nv);
          if (tid < wq) {
            actw[tid] = warm_row;
#pragma unroll
            for (int k = 0; k < QP_KLDS; ++k) {
              sv[tid * QP_KLDS + k] = warm_val[k];
              sc[tid * QP_KLDS + k] = warm_col[k];
            }
          }
          wg_lds_barrier();
          // lane = row of the result, groups of 32 (nv <= 32) or 64 lanes stride the columns; always QP_KLDS terms (the
          // entries behind a row's last one are 0 * column 0), so the reads of a sum do not wait for each other and
          // two columns are in flight per thread - the loops over a run-time K were chains of dependent LDS round trips
          const int RL = nv <= 32 ? 32 : 64, NG = 256 / RL;
          const int li_ = tid & (RL - 1), cg = tid / RL;
          if (li_ < nv) {
#pragma unroll 2
            for (int c = cg; c < wq; c += NG) {
              double v4[QP_KLDS], h4[QP_KLDS];
#pragma unroll
              for (int k = 0; k < QP_KLDS; ++k) {
                v4[k] = sv[c * QP_KLDS + k];
                h4[k] = qpws[li_ + sc[c * QP_KLDS + k] * nv];
              }
              double sacc = 0.0;
#pragma unroll
              for (int k = 0; k < QP_KLDS; ++k) sacc += v4[k] * h4[k];
              HNw[li_ + c * nv] = sacc;
            }
          }
          wg_lds_barrier();
          if (li_ < wq) {
            double v4[QP_KLDS];
            int c4[QP_KLDS];
#pragma unroll
            for (int k = 0; k < QP_KLDS; ++k) {
              v4[k] = sv[li_ * QP_KLDS + k];
              c4[k] = sc[li_ * QP_KLDS + k];
            }
#pragma unroll 2
            for (int c2 = cg; c2 < wq; c2 += NG) {
              double h4[QP_KLDS];
#pragma unroll
              for (int k = 0; k < QP_KLDS; ++k) h4[k] = HNw[c4[k] + c2 * nv];
              double sacc = 0.0;
#pragma unroll
              for (int k = 0; k < QP_KLDS; ++k) sacc += v4[k] * h4[k];
              Sw[li_ + c2 * nv] = sacc;
            }
          }
          wg_lds_barrier();
        } else {
          for (int c = tid; c < wq; c += 256) actw[c] = a.warm[1 + c];
          __syncthreads();
          for (int e = tid; e < nv * wq; e += 256) {
            const int i = e % nv, c = e / nv, row = actw[c];
            double sacc = 0.0;
            for (int k = 0; k < K; ++k) sacc += a.ell.val[k * nr + row] * qpws[i + a.ell.col[k * nr + row] * nv];
            HNw[i + c * nv] = sacc;
          }
          __syncthreads();
          for (int e = tid; e < wq * wq; e += 256) {
            const int c = e % wq, c2 = e / wq, row = actw[c];
            double sacc = 0.0;
            for (int k = 0; k < K; ++k) sacc += a.ell.val[k * nr + row] * HNw[a.ell.col[k * nr + row] + c2 * nv];
            Sw[c + c2 * nv] = sacc;
          }
          __syncthreads();
        }
        if (stamps && tid == 0) stamps[13] = wall_clock64();  // H^-1 N', N H^-1 N'
        if (wg_spd_inverse_pp(Sw, Hq, wq, nv)) wq = 0;   // dependent rows: cold start
        if (stamps && tid == 0) stamps[14] = wall_clock64();  // its inverse
      }
    }
    // single-problem launches: the iteration by all four waves (12 % faster cold solves; in the batched kernel it would
    // cost the third workgroup per CU and is slower: 1.64 against 1.27 ms per 4096 problems)
    if (WARM && a.qp_wg && nv <= 32 && a.ell.K <= QP_KLDS) {
      const int st = qp_gi_wg(f, a.ell, bq, nv, nr, qpws, qpws + qp_lds_doubles(nv, nr) - 64, xout, 1e-10, stamps, hbad, wq,
                              WARM ? a.warm : nullptr);
      if (tid == 0) *st_sh = st;
    } else if (tid < 64) {
      int st = WARM ? qp_goldfarb_idnani(Hq, f, a.ell, bq, nv, nr, qpws, xout, 1e-10, stamps, have_hinv, hbad, wq, a.warm)
                    : qp_goldfarb_idnani(Hq, f, a.ell, bq, nv, nr, qpws, xout, 1e-10, stamps, have_hinv, hbad);
      if (tid == 0) *st_sh = st;
    }
    // the other waves wait here (the solver itself only uses wave-local synchronisation)
    __syncthreads();
    if (stamps && tid == 0) stamps[5] = wall_clock64();
    status = *st_sh;
    if (status || iter == iter0 + a.iters - 1) break;
    // ---- lifted horizon for the next linearisation (Kmpc.m:891-895) ----
    lifted_horizon(xout);
  }
  if (tid == 0) a.status[pb] = status ? KP_ERR_QP_FAIL : KP_OK;
  if (a.done_flag && pb == 0) {
    // single zero-copy step: tell the spinning host thread that U, z and the status are in its memory.  Every wave has
    // waited for its stores at the barrier; lane 0 then releases at system scope and stores the sequence number.
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      if (stamps) { a.done_flag[1] = (unsigned long long)stamps[0]; a.done_flag[2] = (unsigned long long)wall_clock64(); }
      __threadfence_system();
      __hip_atomic_store(&a.done_flag[0], a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Two entry points over the same body: the single-problem kernel (warm start, workgroup-wide solver) may use every register
// it likes; the batched one is held to 128 so that FOUR workgroups share a CU (with a.alias their LDS fits).
template <bool WARM>
__global__ __launch_bounds__(256) void kp_mpc_step_kernel(MpcArgs a) {
  mpc_step_body<WARM>(a);
}
__global__ __launch_bounds__(256, 4) void kp_mpc_step_batch_kernel(MpcArgs a) {
  mpc_step_body<false>(a);
}

// ---- host API ------------------------------------------------------------------------------------

static int dev_alloc_copy(kp_ctx* ctx, double** dst, const double* src, size_t n) {
  *dst = nullptr;
  if (!n) return KP_OK;
  KP_HIP(ctx, hipMalloc((void**)dst, n * 8));
  if (src) KP_HIP(ctx, hipMemcpy(*dst, src, n * 8, hipMemcpyHostToDevice));
  return KP_OK;
}

extern "C" int kp_mpc_destroy(kp_mpc* M) {
  if (!M) return KP_OK;
  (void)hipSetDevice(M->ctx->device);
  double* ptrs[] = {M->A, M->B, M->P, M->S0, M->r, M->Aq, M->bq0, M->Anorm, M->work, M->d_in, M->d_out};
  for (double* p : ptrs)
    if (p) (void)hipFree(p);
  if (M->h_in) (void)hipHostFree(M->h_in);
  if (M->h_out) (void)hipHostFree(M->h_out);
  if (M->warm) (void)hipFree(M->warm);
  if (M->ellc) (void)hipFree(M->ellc);
  if (M->ellv) (void)hipFree(M->ellv);
  if (M->sb_lohi) (void)hipFree(M->sb_lohi);
  if (M->sb_Apow) (void)hipFree(M->sb_Apow);
  if (M->sb_A) (void)hipFree(M->sb_A);
  if (M->sb_b) (void)hipFree(M->sb_b);
  if (M->sb_col) (void)hipFree(M->sb_col);
  if (M->sb_work) (void)hipFree(M->sb_work);
  delete M;
  return KP_OK;
}

extern "C" int kp_mpc_create(kp_ctx* ctx, int model_type, const double* A, const double* B, int N, int m, int Np,
                             const double* proj, int nproj, double q_run, double q_term, const double* r, const double* lo,
                             const double* hi, double slope_lim, double smooth_lim, kp_mpc** out) {
  if (!ctx || !out) return KP_ERR_ARG;
  *out = nullptr;
  if (!A || !B || !proj || !r || N < 1 || m < 1 || Np < 1 || nproj < 1)
    return ctx->fail(KP_ERR_ARG, "kp_mpc_create: bad argument");
  if (model_type != KP_MODEL_LINEAR && model_type != KP_MODEL_BILINEAR)
    return ctx->fail(KP_ERR_ARG, "kp_mpc_create: model_type must be linear or bilinear (NMPC is out of scope)");
  if ((lo == nullptr) != (hi == nullptr)) return ctx->fail(KP_ERR_ARG, "kp_mpc_create: lo and hi must both be given or both NULL");
  const int nvar = m * Np;
  if (nvar > QP_MAXN) return ctx->fail(KP_ERR_ARG, "kp_mpc_create: m*horizon must be <= 64");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const bool has_slope = !std::isnan(slope_lim) && Np >= 2, has_smooth = !std::isnan(smooth_lim) && Np >= 3;
  // constraint rows in the order the reference stacks them (Kmpc.m:641-730, then :865-870)
  int nb = lo ? 2 * m * (Np + 1) : 0, ns = has_slope ? 2 * m * (Np - 1) : 0, nsm = has_smooth ? 2 * m * (Np - 2) : 0;
  const int nrows = nb + ns + nsm + 2 * m;
  std::vector<double> Aq((size_t)nrows * nvar, 0.0), bq(nrows, 0.0);
  auto at = [&](int row, int col) -> double& { return Aq[(size_t)col * nrows + row]; };
  int row0 = 0;
  if (lo) {  // Fbounds: kron(I_Np, [-I; I]) in the first 2m*Np rows, last 2m rows zero (Kmpc.m:646-663)
    for (int j = 0; j < Np; ++j)
      for (int i = 0; i < m; ++i) {
        at(row0 + j * 2 * m + i, j * m + i) = -1.0;
        bq[row0 + j * 2 * m + i] = -lo[i];
        at(row0 + j * 2 * m + m + i, j * m + i) = 1.0;
        bq[row0 + j * 2 * m + m + i] = hi[i];
      }
    row0 += nb;
  }
  if (has_slope) {  // [u_{j+1} - u_j ; -(u_{j+1} - u_j)] <= slope_lim   (Kmpc.m:670-688)
    int h = m * (Np - 1);
    for (int j = 0; j < Np - 1; ++j)
      for (int i = 0; i < m; ++i) {
        int rr = j * m + i;
        at(row0 + rr, j * m + i) = -1.0;
        at(row0 + rr, (j + 1) * m + i) = 1.0;
        at(row0 + h + rr, j * m + i) = 1.0;
        at(row0 + h + rr, (j + 1) * m + i) = -1.0;
        bq[row0 + rr] = slope_lim;
        bq[row0 + h + rr] = slope_lim;
      }
    row0 += ns;
  }
  if (has_smooth) {  // u_j - 2u_{j+1} + u_{j+2}   (Kmpc.m:694-708)
    int h = m * (Np - 2);
    for (int j = 0; j < Np - 2; ++j)
      for (int i = 0; i < m; ++i) {
        int rr = j * m + i;
        at(row0 + rr, j * m + i) = 1.0;
        at(row0 + rr, (j + 1) * m + i) = -2.0;
        at(row0 + rr, (j + 2) * m + i) = 1.0;
        at(row0 + h + rr, j * m + i) = -1.0;
        at(row0 + h + rr, (j + 1) * m + i) = 2.0;
        at(row0 + h + rr, (j + 2) * m + i) = -1.0;
        bq[row0 + rr] = smooth_lim;
        bq[row0 + h + rr] = smooth_lim;
      }
    row0 += nsm;
  }
  for (int i = 0; i < m; ++i) {  // memory rows (Kmpc.m:865): [I; -I] on u_0
    at(row0 + i, i) = 1.0;
    at(row0 + m + i, i) = -1.0;
  }
  kp_mpc* M = new kp_mpc();
  M->ctx = ctx;
  M->model_type = model_type;
  M->N = N; M->m = m; M->Np = Np; M->nproj = nproj; M->nvar = nvar; M->nrows = nrows;
  M->mb = model_type == KP_MODEL_BILINEAR ? N * m : m;
  M->q_run = q_run; M->q_term = q_term;
  int rc = dev_alloc_copy(ctx, &M->A, A, (size_t)N * N);
  if (!rc) rc = dev_alloc_copy(ctx, &M->B, B, (size_t)N * M->mb);
  // P | PB in ONE allocation (PB at an even offset), padded to a multiple of 4 KB: a single-problem step copies the block
  // to LDS with 1 KB loads per wave
  const size_t nP = ((size_t)(Np + 1) * nproj * N + 1) & ~(size_t)1;
  const size_t nPB = model_type == KP_MODEL_BILINEAR ? (size_t)Np * nproj * m * N : 0;
  M->stage_doubles = (int)((nP + nPB + 511) & ~(size_t)511);
  if (!rc) rc = dev_alloc_copy(ctx, &M->P, nullptr, (size_t)M->stage_doubles);
  if (!rc && nPB) M->PB = M->P + nP;
  if (!rc) rc = dev_alloc_copy(ctx, &M->S0, nullptr, (size_t)Np * nproj * m);
  if (!rc) rc = dev_alloc_copy(ctx, &M->r, r, m);
  if (!rc) rc = dev_alloc_copy(ctx, &M->Aq, Aq.data(), Aq.size());
  if (!rc) rc = dev_alloc_copy(ctx, &M->bq0, bq.data(), nrows);
  std::vector<double> ev, en;
  std::vector<int> ec;
  to_ell(Aq.data(), nrows, nvar, ev, ec, en, M->ellK);
  if (!rc) rc = dev_alloc_copy(ctx, &M->Anorm, en.data(), nrows);
  if (!rc) rc = dev_alloc_copy(ctx, &M->ellv, ev.data(), ev.size());
  if (!rc) {
    hipError_t e2 = hipMalloc((void**)&M->ellc, ec.size() * 4);
    if (e2 == hipSuccess) e2 = hipMemcpy(M->ellc, ec.data(), ec.size() * 4, hipMemcpyHostToDevice);
    if (e2 != hipSuccess) rc = ctx->fail(KP_ERR_HIP, std::string("kp_mpc_create: ") + hipGetErrorString(e2));
  }
  double* dproj = nullptr;
  if (!rc) rc = dev_alloc_copy(ctx, &dproj, proj, (size_t)nproj * N);
  if (rc) {
    if (dproj) (void)hipFree(dproj);
    kp_mpc_destroy(M);
    return rc;
  }
  hipLaunchKernelGGL(kp_mpc_setup_kernel, dim3(1), dim3(256), 0, ctx->stream, M->A, M->B, dproj, N, m, Np, nproj,
                     model_type == KP_MODEL_LINEAR ? 1 : 0, M->P, M->S0, M->PB);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(dproj);
  if (e != hipSuccess) {
    kp_mpc_destroy(M);
    return ctx->fail(KP_ERR_HIP, std::string("kp_mpc_create: ") + hipGetErrorString(e));
  }
  *out = M;
  return KP_OK;
}

extern "C" int kp_mpc_dims(const kp_mpc* M, int* nvar, int* nrows) {
  if (!M) return KP_ERR_ARG;
  if (nvar) *nvar = M->nvar;
  if (nrows) *nrows = M->nrows;
  return KP_OK;
}

// ---- state bounds (Kmpc.m:300-318, :716-730) -----------------------------------------------------------------------
// The reference bounds the entries s = i n + j (block i = 0..Np, j < n) of the STACKED lifted state [z_0; z_1; ...]
// (the kron block of E is written into its first (Np+1) n columns, :306 - not strided by N; restated literally, as the
// oracle does): entry s belongs to step k = s / N, component c = s % N, so
//   (Ahat z)_s = (A^k z)_c,   (Bhat U)_s = sum_{j < k} (A^(k-1-j) Bz)_c,: u_j,   Bz = Beta(z) (bilinear) or B (linear),
// rows  -(Bhat U)_s <= -lo_j + (Ahat z)_s  and  (Bhat U)_s <= hi_j - (Ahat z)_s.  These rows are dense in U and, for a
// bilinear model, change with z: kp_mpc_sb_kernel writes the whole constraint matrix of the step densely (the constant
// sparse rows expanded, then the state-bound rows) and the generic QP kernel solves with it.
__global__ void kp_matmul_nn_kernel(const double* __restrict__ X, const double* __restrict__ Y, int N, double* __restrict__ Z) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * N) return;
  const int i = e % N, j = e / N;
  double s = 0.0;
  for (int k = 0; k < N; ++k) s += X[i + (size_t)k * N] * Y[k + (size_t)j * N];
  Z[e] = s;
}

__global__ __launch_bounds__(256) void kp_mpc_sb_kernel(int model_type, int N, int m, int Np, int nvar, int nrows, int n_sb, int kmax,
                                                        const double* __restrict__ Apow, const double* __restrict__ B,
                                                        const double* __restrict__ Aq, const double* __restrict__ b_in,
                                                        const double* __restrict__ z, const double* __restrict__ lohi,
                                                        double* __restrict__ Ad, double* __restrict__ bd, double* __restrict__ nrm,
                                                        size_t b_stride) {
  extern __shared__ double sm[];
  double* Bz = sm;                         // N x m
  double* zs = Bz + N * m;                 // N
  const int tid = threadIdx.x;
  const int mr = nrows + 2 * n_sb * (Np + 1);
  {                                        // problem p = blockIdx.x of a batch
    const size_t p = blockIdx.x;
    b_in += p * b_stride; z += p * N;
    Ad += p * (size_t)mr * nvar; bd += p * mr; nrm += p * mr;
  }
  for (int c = tid; c < N; c += 256) zs[c] = z[c];
  __syncthreads();
  for (int e = tid; e < N * m; e += 256) {
    const int r = e % N, i = e / N;
    double s;
    if (model_type == KP_MODEL_BILINEAR) {
      s = 0.0;
      const double* Bi = B + (size_t)i * N * N;
      for (int c = 0; c < N; ++c) s += Bi[r + (size_t)c * N] * zs[c];
    } else {
      s = B[r + (size_t)i * N];
    }
    Bz[e] = s;
  }
  __syncthreads();
  // constant rows: expanded copy
  for (int e = tid; e < nrows * nvar; e += 256) {
    const int r = e % nrows, v = e / nrows;
    Ad[r + (size_t)v * mr] = Aq[e];
  }
  for (int r = tid; r < nrows; r += 256) bd[r] = b_in[r];
  // state-bound rows: one (stacked entry, variable) pair per thread and pass
  const int nst = n_sb * (Np + 1);
  for (int e = tid; e < nst * nvar; e += 256) {
    const int st = e % nst, v = e / nst;            // stacked entry, variable v = jj * m + ii
    const int blk = st / n_sb, j = st % n_sb;
    const int k = st / N, c = st % N;
    const int jj = v / m, ii = v % m;
    double val = 0.0;
    if (jj < k) {
      const double* Ap = Apow + (size_t)(k - 1 - jj) * N * N;       // row c of A^(k-1-jj)
      for (int r = 0; r < N; ++r) val += Ap[c + (size_t)r * N] * Bz[r + ii * N];
    }
    const int rneg = nrows + blk * 2 * n_sb + j, rpos = rneg + n_sb;
    Ad[rneg + (size_t)v * mr] = -val;
    Ad[rpos + (size_t)v * mr] = val;
  }
  for (int st = tid; st < nst; st += 256) {
    const int blk = st / n_sb, j = st % n_sb;
    const int k = st / N, c = st % N;
    const double* Ap = Apow + (size_t)k * N * N;
    double az = 0.0;
    for (int r = 0; r < N; ++r) az += Ap[c + (size_t)r * N] * zs[r];
    const int rneg = nrows + blk * 2 * n_sb + j, rpos = rneg + n_sb;
    bd[rneg] = -lohi[j] + az;
    bd[rpos] = lohi[n_sb + j] - az;
  }
  __syncthreads();
  __threadfence_block();
  for (int r = tid; r < mr; r += 256) {             // row norms (the solver scales violations by them)
    double s = 0.0;
    for (int v = 0; v < nvar; ++v) {
      const double a = Ad[r + (size_t)v * mr];
      s += a * a;
    }
    nrm[r] = sqrt(s);
  }
}

extern "C" int kp_mpc_set_state_bounds(kp_mpc* M, int n, const double* lo, const double* hi) {
  if (!M) return KP_ERR_ARG;
  kp_ctx* ctx = M->ctx;
  if (n < 0 || n > M->N || (n > 0 && (!lo || !hi))) return ctx->fail(KP_ERR_ARG, "kp_mpc_set_state_bounds: bad argument");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  double** bufs[] = {&M->sb_lohi, &M->sb_Apow, &M->sb_A, &M->sb_b, &M->sb_work};
  for (double** p : bufs)
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  M->sb_cap = 0;
  if (M->sb_col) { (void)hipFree(M->sb_col); M->sb_col = nullptr; }
  M->sb_n = 0; M->sb_rows = 0;
  if (n == 0) return KP_OK;
  const int N = M->N, Np = M->Np, nv = M->nvar;
  const int kmax = ((Np + 1) * n - 1) / N;
  const int mr = M->nrows + 2 * n * (Np + 1);
  std::vector<double> lh(2 * n);
  for (int i = 0; i < n; ++i) { lh[i] = lo[i]; lh[n + i] = hi[i]; }
  int rc = dev_alloc_copy(ctx, &M->sb_lohi, lh.data(), lh.size());
  if (rc) return rc;
  KP_HIP(ctx, hipMalloc((void**)&M->sb_Apow, (size_t)(kmax + 1) * N * N * 8));
  std::vector<double> eye((size_t)N * N, 0.0);
  for (int i = 0; i < N; ++i) eye[(size_t)i * N + i] = 1.0;
  KP_HIP(ctx, hipMemcpy(M->sb_Apow, eye.data(), (size_t)N * N * 8, hipMemcpyHostToDevice));
  for (int p = 1; p <= kmax; ++p) {
    hipLaunchKernelGGL(kp_matmul_nn_kernel, dim3((N * N + 255) / 256), dim3(256), 0, ctx->stream, M->sb_Apow + (size_t)(p - 1) * N * N, M->A, N,
                       M->sb_Apow + (size_t)p * N * N);
    KP_HIP(ctx, hipGetLastError());
  }
  std::vector<int> col((size_t)mr * nv);
  for (int k = 0; k < nv; ++k)
    for (int r = 0; r < mr; ++r) col[(size_t)k * mr + r] = k;
  KP_HIP(ctx, hipMalloc((void**)&M->sb_col, col.size() * sizeof(int)));
  KP_HIP(ctx, hipMemcpy(M->sb_col, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice));
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  M->sb_n = n; M->sb_kmax = kmax; M->sb_rows = 2 * n * (Np + 1);
  return KP_OK;
}

static int mpc_run(kp_mpc* M, const kp_basis* basis, int nb, const double* z, const double* zeta, const double* u_prev,
                   const double* Yr, int iters, double* U_out, double* z_out, int* status) {
  kp_ctx* ctx = M->ctx;
  if (nb < 1 || !u_prev || !Yr || !U_out || iters < 1 || (!z && !zeta)) return ctx->fail(KP_ERR_ARG, "kp_mpc_step: bad argument");
  if (iters > 1 && M->model_type != KP_MODEL_BILINEAR) iters = 1;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const int N = M->N, m = M->m, Np = M->Np, nproj = M->nproj, nv = M->nvar, nr = M->nrows;
  int nzeta = 0;
  if (zeta) {
    if (!basis) return ctx->fail(KP_ERR_ARG, "kp_mpc_step_zeta: basis required");
    if (basis->dev.N != N || basis->dev.model_type == KP_MODEL_NONLINEAR)
      return ctx->fail(KP_ERR_ARG, "kp_mpc_step_zeta: basis does not match the controller's model");
    nzeta = basis->dev.nzeta;
  }
  const size_t n_out = (size_t)nv + N;
  const size_t n_ex = (size_t)nv * nv + nv + nr;
  const size_t in_per = (size_t)std::max(N, 64) + m + (size_t)nproj * (Np + 1);
  if (M->io_problems < (size_t)nb) {
    if (M->d_in) (void)hipFree(M->d_in);
    if (M->d_out) (void)hipFree(M->d_out);
    if (M->work) (void)hipFree(M->work);
    if (M->h_in) (void)hipHostFree(M->h_in);
    if (M->h_out) (void)hipHostFree(M->h_out);
    M->d_in = M->d_out = M->work = M->h_in = M->h_out = nullptr;
    M->d_status = nullptr;
    M->io_problems = 0;
    size_t cap = (size_t)nb;
    KP_HIP(ctx, hipMalloc((void**)&M->d_in, cap * in_per * 8));
    KP_HIP(ctx, hipMalloc((void**)&M->d_out, cap * n_out * 8 + cap * sizeof(int) + 8));
    KP_HIP(ctx, hipMalloc((void**)&M->work, (n_ex + 16) * 8));
    KP_HIP(ctx, hipMemset(M->work, 0, (n_ex + 16) * 8));      // (stamps a step does not reach read as zero, not as stale memory)
    KP_HIP(ctx, hipHostMalloc((void**)&M->h_in, cap * in_per * 8, hipHostMallocDefault));
    KP_HIP(ctx, hipHostMalloc((void**)&M->h_out, cap * n_out * 8 + cap * sizeof(int) + 8, hipHostMallocDefault));
    M->io_problems = cap;
  }
  M->d_status = (int*)(M->d_out + M->io_problems * n_out);
  const size_t nz = zeta ? nzeta : N;
  double* d_z = M->d_in;
  double* d_up = d_z + (size_t)nb * nz;
  double* d_yr = d_up + (size_t)nb * m;
  // inputs packed in the pinned buffer in device order: one host-to-device copy
  const size_t n_in = (size_t)nb * (nz + m + (size_t)nproj * (Np + 1));
  memcpy(M->h_in, zeta ? zeta : z, (size_t)nb * nz * 8);
  memcpy(M->h_in + (size_t)nb * nz, u_prev, (size_t)nb * m * 8);
  memcpy(M->h_in + (size_t)nb * (nz + m), Yr, (size_t)nb * nproj * (Np + 1) * 8);
  // single problem: the kernel reads the (few hundred bytes of) inputs straight from the pinned, device-mapped
  // staging buffer and writes its outputs there - no copy commands at all on the latency path of a closed loop
  static const bool no_zc = getenv("KP_MPC_NO_ZEROCOPY") != nullptr;
  const bool zc = nb == 1 && !no_zc;
  if (zc) {
    d_z = M->h_in;
    d_up = d_z + (size_t)nb * nz;
    d_yr = d_up + (size_t)nb * m;
  } else {
    KP_HIP(ctx, hipMemcpyAsync(d_z, M->h_in, n_in * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  MpcArgs a{};
  if (zeta) a.basis = basis->dev;
  a.has_basis = zeta ? 1 : 0;
  a.model_type = M->model_type; a.N = N; a.m = m; a.Np = Np; a.nproj = nproj; a.nvar = nv; a.nrows = nr; a.iters = iters;
  a.q_run = M->q_run; a.q_term = M->q_term;
  a.A = M->A; a.B = M->B; a.P = M->P; a.S0 = M->S0; a.PB = M->PB; a.r = M->r; a.Aq = M->Aq; a.bq0 = M->bq0; a.Anorm = M->Anorm;
  a.ell = EllMat{M->ellv, M->ellc, M->Anorm, M->ellK};
  a.z = zeta ? nullptr : d_z;
  a.zeta = zeta ? d_z : nullptr;
  a.u_prev = d_up; a.Yr = d_yr;
  a.U = zc ? M->h_out : M->d_out;
  a.z_out = a.U + (size_t)nb * nv;
  a.qp_export = nb == 1 ? M->work : nullptr;
  a.assemble_only = M->sb_n > 0 ? 1 : 0;
  a.U_lin = nullptr;
  const bool sb = M->sb_n > 0;
  const int mr_sb = nr + M->sb_rows;
  if (sb) {
    // state-bound steps: per problem the exported H, f, b, the dense constraint matrix of the step, right-hand sides,
    // row norms and the solution of the generic QP kernel
    if (M->sb_cap < (size_t)nb) {
      if (M->sb_A) (void)hipFree(M->sb_A);
      if (M->sb_b) (void)hipFree(M->sb_b);
      if (M->sb_work) (void)hipFree(M->sb_work);
      M->sb_A = M->sb_b = M->sb_work = nullptr;
      M->sb_cap = 0;
      KP_HIP(ctx, hipMalloc((void**)&M->sb_A, (size_t)nb * mr_sb * nv * 8));
      KP_HIP(ctx, hipMalloc((void**)&M->sb_b, (size_t)nb * (2 * mr_sb + nv + 1) * 8 + 64));
      KP_HIP(ctx, hipMalloc((void**)&M->sb_work, (size_t)nb * n_ex * 8));
      M->sb_cap = (size_t)nb;
    }
    a.qp_export = M->sb_work;
    a.iters = 1;                       // the linearisation passes are driven from the host (the QP runs in its own kernel)
  }
  static const bool no_warm = getenv("KP_MPC_NO_WARM") != nullptr;
  static const int qp_wg = [] { const char* e = getenv("KP_QP_WG"); return e ? atoi(e) : 1; }();
  a.qp_wg = qp_wg;
  if (!M->warm) {
    KP_HIP(ctx, hipMalloc((void**)&M->warm, (size_t)(1 + nv) * sizeof(int)));
    KP_HIP(ctx, hipMemsetAsync(M->warm, 0, (size_t)(1 + nv) * sizeof(int), ctx->stream));
  }
  a.warm = (nb == 1 && !no_warm) ? M->warm : nullptr;
  a.status = zc ? (int*)(M->h_out + M->io_problems * n_out) : M->d_status;
  a.stamps = nb == 1 ? (long long*)(M->work + n_ex) : nullptr;
  // single zero-copy step without state bounds: the host does not go through hipStreamSynchronize (a 15-20 us wake-up) or
  // event records - the kernel stores a sequence number into pinned memory behind its outputs and the host spins on it
  static const bool no_spin = getenv("KP_MPC_NO_SPIN") != nullptr;
  const bool spin = zc && !sb && !no_spin;
  a.done_flag = nullptr;
  a.done_seq = 0;
  if (spin) {
    if (!M->h_flag) {
      KP_HIP(ctx, hipHostMalloc((void**)&M->h_flag, 64, hipHostMallocDefault));
      memset(M->h_flag, 0, 64);
    }
    a.done_flag = M->h_flag;
    a.done_seq = ++M->step_seq;
  }
  size_t lds = (size_t)mpc_lds_doubles(N, m, Np, nproj, nv, nr, (sb && iters > 1) ? 2 : iters, zeta ? basis->dev.nfull : 0) * 8 + 32;
  {
    // batched launches: the assembly's inputs inside the solver's scratch, no exchange words of the workgroup-wide solver
    const int n_asm = N + N * m + Np * nproj * m + (Np + 1) * nproj;
    a.alias = (nb > 1 && !a.warm && !sb && iters == 1 && n_asm <= 2 * nv * nv) ? 1 : 0;
    if (a.alias) lds -= (size_t)(n_asm + 64) * 8;
  }
  if (lds > 160 * 1024) return ctx->fail(KP_ERR_ARG, "kp_mpc_step: problem too large for LDS");
  // single-problem launches: the model's P | PB block behind everything else in LDS when it fits (a batch reads it from L2,
  // shared by its workgroups, and needs its LDS for occupancy)
  static const bool no_stage = getenv("KP_MPC_NO_STAGE") != nullptr;
  a.stage_off = 0;
  a.stage_doubles = M->stage_doubles;
  if (nb == 1 && !no_stage && lds + 16 + (size_t)M->stage_doubles * 8 <= 160 * 1024) {
    a.stage_off = (int)((lds / 8 + 1) & ~(size_t)1);
    lds = (size_t)(a.stage_off + M->stage_doubles) * 8;
  }
  static KpLdsCache step_lds[2];
  const int wk = a.warm != nullptr;
  static KpLdsCache batch_lds;
  const bool bk = a.alias != 0;                       // the batched entry point
  if (bk) KP_HIP(ctx, kp_ensure_lds(batch_lds, (const void*)kp_mpc_step_batch_kernel, lds));
  else KP_HIP(ctx, kp_ensure_lds(step_lds[wk], wk ? (const void*)kp_mpc_step_kernel<true> : (const void*)kp_mpc_step_kernel<false>, lds));
  if (!spin) KP_HIP(ctx, hipEventRecord(ctx->evp[4], ctx->stream));
  if (!sb) {
    if (bk) hipLaunchKernelGGL(kp_mpc_step_batch_kernel, dim3(nb), dim3(256), lds, ctx->stream, a);
    else if (wk) hipLaunchKernelGGL(kp_mpc_step_kernel<true>, dim3(nb), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL(kp_mpc_step_kernel<false>, dim3(nb), dim3(256), lds, ctx->stream, a);
    KP_HIP(ctx, hipGetLastError());
  } else {
    // Every pass: the step kernel assembles H, f (along the lifted horizon of the previous pass's inputs from pass 2 on,
    // Kmpc.m:874-895) and the right-hand sides of the constant rows and stops; the dense constraint matrix of the step
    // is formed ONCE from z (the reference computes A = get_constraintL_bilinear(zrow) before its iteration loop,
    // Kmpc.m:861); the generic QP kernel solves, one wave per problem.
    double* bd = M->sb_b;                                   // [nb][mr]
    double* nrm = bd + (size_t)nb * mr_sb;                  // [nb][mr]
    double* dx = nrm + (size_t)nb * mr_sb;                  // [nb][nv]
    int* dst = (int*)(dx + (size_t)nb * nv);                // [nb]
    const size_t lq = (size_t)qp_lds_doubles(nv, mr_sb) * 8 + 64;
    static KpLdsCache sbqp_lds;
    KP_HIP(ctx, kp_ensure_lds(sbqp_lds, (const void*)kp_qp_kernel, lq));
    EllMat E{M->sb_A, M->sb_col, nrm, nv};
    for (int pass = 0; pass < iters; ++pass) {
      a.U_lin = pass > 0 ? dx : nullptr;
      if (wk) hipLaunchKernelGGL(kp_mpc_step_kernel<true>, dim3(nb), dim3(256), lds, ctx->stream, a);
      else hipLaunchKernelGGL(kp_mpc_step_kernel<false>, dim3(nb), dim3(256), lds, ctx->stream, a);
      KP_HIP(ctx, hipGetLastError());
      if (pass == 0) {
        hipLaunchKernelGGL(kp_mpc_sb_kernel, dim3(nb), dim3(256), (size_t)(N * m + N) * 8, ctx->stream, M->model_type, N, m, Np, nv, nr, M->sb_n,
                           M->sb_kmax, M->sb_Apow, M->B, M->Aq, M->sb_work + (size_t)nv * nv + nv, a.z_out, M->sb_lohi, M->sb_A, bd, nrm, n_ex);
        KP_HIP(ctx, hipGetLastError());
      }
      hipLaunchKernelGGL(kp_qp_kernel, dim3(nb), dim3(64), lq, ctx->stream, M->sb_work, M->sb_work + (size_t)nv * nv, E, bd, nv, mr_sb, dx, dst,
                         n_ex, n_ex, (size_t)mr_sb * nv, (size_t)mr_sb, pass > 0 ? 1 : 0);
      KP_HIP(ctx, hipGetLastError());
    }
    // x and the status words into the output block the host reads below
    KP_HIP(ctx, hipMemcpyAsync(a.U, dx, (size_t)nb * nv * 8, zc ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    KP_HIP(ctx, hipMemcpyAsync(a.status, dst, (size_t)nb * sizeof(int), zc ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    if (nb == 1)   // kp_mpc_last_qp reads the single-problem export
      KP_HIP(ctx, hipMemcpyAsync(M->work, M->sb_work, n_ex * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (!spin) KP_HIP(ctx, hipEventRecord(ctx->evp[5], ctx->stream));
  // one device-to-host copy: x and z of every problem, then the status words
  const size_t out_bytes = M->io_problems * n_out * 8 + (size_t)nb * sizeof(int);
  if (!zc) KP_HIP(ctx, hipMemcpyAsync(M->h_out, M->d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  bool spun = false;
  if (spin) {
    volatile unsigned long long* fl = M->h_flag;
    for (long it = 0; it < 20000000L; ++it) {          // a few ms at most; a slower kernel falls through to the stream wait
      if (__atomic_load_n(&fl[0], __ATOMIC_ACQUIRE) == a.done_seq) { spun = true; break; }
      __builtin_ia32_pause();
    }
  }
  if (!spun) KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // x = [u_0; u_1; ...] (m each)  ->  U (Np x m column-major) = reshape(x,[m,Np])'  (Kmpc.m:884)
  const double* x = M->h_out;
  const int* st = (const int*)(M->h_out + M->io_problems * n_out);
  if (z_out) memcpy(z_out, M->h_out + (size_t)nb * nv, (size_t)nb * N * 8);
  for (int p = 0; p < nb; ++p) {
    for (int j = 0; j < Np; ++j)
      for (int i = 0; i < m; ++i) U_out[(size_t)p * nv + (size_t)i * Np + j] = x[(size_t)p * nv + j * m + i];
    if (status) status[p] = st[p];
  }
  float ms = 0;
  if (spin) ctx->timers[2] = (double)(M->h_flag[2] - M->h_flag[1]) * 1e-5;        // wall_clock64 ticks of 10 ns -> ms
  else if (hipEventElapsedTime(&ms, ctx->evp[4], ctx->evp[5]) == hipSuccess) ctx->timers[2] = ms;
  return KP_OK;
}

extern "C" int kp_mpc_step(kp_mpc* M, const double* z, const double* u_prev, const double* Yr, int iters, double* U_out,
                           int* status) {
  if (!M) return KP_ERR_ARG;
  return mpc_run(M, nullptr, 1, z, nullptr, u_prev, Yr, iters, U_out, nullptr, status);
}

extern "C" int kp_mpc_step_zeta(kp_mpc* M, const kp_basis* basis, const double* zeta, const double* u_prev, const double* Yr,
                                int iters, double* U_out, double* z_out, int* status) {
  if (!M) return KP_ERR_ARG;
  return mpc_run(M, basis, 1, nullptr, zeta, u_prev, Yr, iters, U_out, z_out, status);
}

extern "C" int kp_mpc_step_batch(kp_mpc* M, int nb, const double* z, const double* u_prev, const double* Yr, double* U_out,
                                 int* status) {
  if (!M) return KP_ERR_ARG;
  return mpc_run(M, nullptr, nb, z, nullptr, u_prev, Yr, 1, U_out, nullptr, status);
}

extern "C" int kp_mpc_last_qp(kp_mpc* M, double* Hq, double* f, double* Aq, double* bq) {
  if (!M || !M->work) return M ? M->ctx->fail(KP_ERR_ARG, "kp_mpc_last_qp: no single-problem step has run") : KP_ERR_ARG;
  kp_ctx* ctx = M->ctx;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nv = M->nvar, nr = M->nrows;
  if (Hq) KP_HIP(ctx, hipMemcpy(Hq, M->work, nv * nv * 8, hipMemcpyDeviceToHost));
  if (f) KP_HIP(ctx, hipMemcpy(f, M->work + nv * nv, nv * 8, hipMemcpyDeviceToHost));
  if (bq) KP_HIP(ctx, hipMemcpy(bq, M->work + nv * nv + nv, nr * 8, hipMemcpyDeviceToHost));
  if (Aq) KP_HIP(ctx, hipMemcpy(Aq, M->Aq, nr * nv * 8, hipMemcpyDeviceToHost));
  return KP_OK;
}

// Diagnostics: phase times (microseconds) of the most recent single-problem step:
// [0] lift + e, [1] Beta/S, [2] H/f, [3] Hinv (Gauss-Jordan), [4] active-set iterations,
// [5] total kernel; counts[0] = solver iterations, counts[1] = active constraints at the optimum.
// every stamp of the most recent single step relative to its first, in microseconds (slots the step did not reach hold the value
// of an earlier step, or minus the first stamp if no step reached them yet): [1] tracking error, [2] S_k, [3] H and f, [4] solver entered, [5] solved, [10] inputs landed, [11] lifted
// state, [6] iteration loop entered, [12] H^-1, [13] warm-start products, [14] inverse of the warm set's Schur complement
extern "C" int kp_mpc_last_stamps(kp_mpc* M, double* us16) {
  if (!M || !M->work || !us16) return KP_ERR_ARG;
  kp_ctx* ctx = M->ctx;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_ex = (size_t)M->nvar * M->nvar + M->nvar + M->nrows;
  long long st[16];
  KP_HIP(ctx, hipMemcpy(st, M->work + n_ex, sizeof(st), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) us16[i] = (i == 8 || i == 9) ? (double)st[i] : (st[i] - st[0]) * 0.01;
  return KP_OK;
}

extern "C" int kp_mpc_last_profile(kp_mpc* M, double* us, int* counts) {
  if (!M || !M->work || !us) return KP_ERR_ARG;
  kp_ctx* ctx = M->ctx;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_ex = (size_t)M->nvar * M->nvar + M->nvar + M->nrows;
  long long st[16];
  KP_HIP(ctx, hipMemcpy(st, M->work + n_ex, sizeof(st), hipMemcpyDeviceToHost));
  const double tick = 0.01;  // wall_clock64: 100 MHz
  us[0] = (st[1] - st[0]) * tick;
  us[1] = (st[2] - st[1]) * tick;
  us[2] = (st[3] - st[2]) * tick;
  us[3] = (st[4] - st[3]) * tick;
  us[4] = (st[5] - st[4]) * tick;
  us[5] = (st[5] - st[0]) * tick;
  if (counts) {
    counts[0] = (int)st[8];
    counts[1] = (int)st[9];
  }
  return KP_OK;
}

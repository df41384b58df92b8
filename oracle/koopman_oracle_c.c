/* koopman_oracle_c.c - plain-C restatement of the fit half of the hot path (Ksysid.get_Koopman), the compiled sibling of
 * oracle/koopman_oracle.py.
 *
 * TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY: built by oracle/Makefile into oracle/libkoopman_oracle.so and loaded by
 * tests/ and by bench.py's cpu_baseline leg.  Nothing under koopman-realizations_amd/ links, loads or calls it.
 * Pinned against oracle/koopman_oracle.py (itself pinned to the reference's stored artefacts) in tests/test_oracle_c.py.
 *
 * What it restates, literally:
 *   ko_lift_rows     the per-snapshot lift loop of get_Koopman, Ksysid.m:1030-1065: for every snapshot pair the row
 *                    [psi(x), u] (linear, :1062), psi(x) (x) [1; u] (bilinear, :1049-1053) or psi([x; u]) (nonlinear,
 *                    :1039-1043) with psi = [zeta; monomials of def_polyLift (:629-677, exponent table from
 *                    partitions.m:206-219); 1]
 *   ko_qr_lstsq      K = Px \ Py (Ksysid.m:1069): MATLAB's mldivide on a full-rank rectangular system is a Householder
 *                    QR least-squares solve; restated as unblocked Householder QR (LAPACK dgeqr2 / dorm2r order of
 *                    operations) + back substitution
 *   ko_get_koopman   the two together (lasso = Inf branch)
 * All matrices column-major (MATLAB layout), f64.  OpenMP threads over columns / rows where the reference's BLAS would
 * thread; ko_set_threads(1) gives the single-thread row of the baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void ko_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int ko_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* psi(v) for one point: v (nvars), exps (nmono x nvars, row-major bytes: monomials AFTER the first nvars), out (N = nvars +
 * nmono + 1): [v ; monomials ; 1]   (Ksysid.m:484-505) */
static void lift_point(const double* v, int nvars, const uint8_t* exps, int nmono, double* out) {
  for (int i = 0; i < nvars; ++i) out[i] = v[i];
  for (int r = 0; r < nmono; ++r) {
    double p = 1.0;
    const uint8_t* e = exps + (size_t)r * nvars;
    for (int i = 0; i < nvars; ++i)
      for (int k = 0; k < e[i]; ++k) p *= v[i];
    out[nvars + r] = p;
  }
  out[nvars + nmono] = 1.0;
}

/* model_type: 0 linear, 1 bilinear, 2 nonlinear.  zeta: Ns x nzeta, u: Ns x m (column-major).  P: Ns x W. */
int ko_lift_rows(int model_type, int nzeta, int m, const uint8_t* exps, int nmono, const double* zeta, const double* u, int64_t Ns,
                 double* P) {
  const int nvars = model_type == 2 ? nzeta + m : nzeta;
  const int N = nvars + nmono + 1;
  const int W = model_type == 0 ? N + m : model_type == 1 ? N * (m + 1) : N;
  if (nvars > 64) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < Ns; ++s) {
    double v[64], psi[4096];
    for (int i = 0; i < nzeta; ++i) v[i] = zeta[(size_t)i * Ns + s];
    if (model_type == 2)
      for (int i = 0; i < m; ++i) v[nzeta + i] = u[(size_t)i * Ns + s];
    lift_point(v, nvars, exps, nmono, psi);
    for (int c = 0; c < N; ++c) P[(size_t)c * Ns + s] = psi[c];
    if (model_type == 0)
      for (int i = 0; i < m; ++i) P[(size_t)(N + i) * Ns + s] = u[(size_t)i * Ns + s];
    else if (model_type == 1)
      for (int i = 0; i < m; ++i) {
        const double ui = u[(size_t)i * Ns + s];
        for (int c = 0; c < N; ++c) P[(size_t)((i + 1) * N + c) * Ns + s] = psi[c] * ui;
      }
  }
  (void)W;
  return 0;
}

/* min ||A X - B||_F for A (rows x n, full column rank, OVERWRITTEN), B (rows x nrhs, OVERWRITTEN); X: n x nrhs. */
int ko_qr_lstsq(double* A, int64_t rows, int n, double* B, int nrhs, double* X) {
  if (rows < n) return -1;
  double* tau = (double*)malloc((size_t)n * sizeof(double));
  if (!tau) return -2;
  for (int k = 0; k < n; ++k) {
    double* a = A + (size_t)k * rows;
    /* Householder reflector of a[k:rows] (dlarfg) */
    double xnorm = 0.0;
    for (int64_t i = k + 1; i < rows; ++i) xnorm += a[i] * a[i];
    xnorm = sqrt(xnorm);
    const double alpha = a[k];
    if (xnorm == 0.0) {
      tau[k] = 0.0;
    } else {
      double beta = -copysign(hypot(alpha, xnorm), alpha);
      tau[k] = (beta - alpha) / beta;
      const double sc = 1.0 / (alpha - beta);
      for (int64_t i = k + 1; i < rows; ++i) a[i] *= sc;
      a[k] = beta;
    }
    const double t = tau[k];
    if (t != 0.0) {
      /* apply H = I - tau v v' (v_k = 1) to the trailing columns of A and to B */
#pragma omp parallel for schedule(static)
      for (int j = k + 1; j < n + nrhs; ++j) {
        double* c = j < n ? A + (size_t)j * rows : B + (size_t)(j - n) * rows;
        double w = c[k];
        for (int64_t i = k + 1; i < rows; ++i) w += a[i] * c[i];
        w *= t;
        c[k] -= w;
        for (int64_t i = k + 1; i < rows; ++i) c[i] -= w * a[i];
      }
    }
  }
  int rc = 0;
  for (int k = 0; k < n; ++k)
    if (A[(size_t)k * rows + k] == 0.0) rc = -3;      /* rank deficient: not handled here (MATLAB pivots) */
  if (!rc) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < nrhs; ++j) {                  /* R X = (Q'B)(1:n, :) */
      const double* b = B + (size_t)j * rows;
      double* x = X + (size_t)j * n;
      for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < n; ++c) s -= A[(size_t)c * rows + i] * x[c];
        x[i] = s / A[(size_t)i * rows + i];
      }
    }
  }
  free(tau);
  return rc;
}

/* get_Koopman, least-squares branch (Ksysid.m:1030-1069): lift both halves of every pair, K = Px \ Py.  K: W x W. */
int ko_get_koopman(int model_type, int nzeta, int m, const uint8_t* exps, int nmono, const double* alpha, const double* beta,
                   const double* u, int64_t Ns, double* K) {
  const int nvars = model_type == 2 ? nzeta + m : nzeta;
  const int N = nvars + nmono + 1;
  const int W = model_type == 0 ? N + m : model_type == 1 ? N * (m + 1) : N;
  double* Px = (double*)malloc((size_t)Ns * W * sizeof(double));
  double* Py = (double*)malloc((size_t)Ns * W * sizeof(double));
  if (!Px || !Py) { free(Px); free(Py); return -2; }
  int rc = ko_lift_rows(model_type, nzeta, m, exps, nmono, alpha, u, Ns, Px);
  if (!rc) rc = ko_lift_rows(model_type, nzeta, m, exps, nmono, beta, u, Ns, Py);
  if (!rc) rc = ko_qr_lstsq(Px, Ns, W, Py, W, K);
  free(Px);
  free(Py);
  return rc;
}

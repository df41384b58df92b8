/* koopman_cpu_abi.c - the CPU backend of the SAME C ABI (include/koopman_hip.h) for the fit path, built on the plain-C
 * restatement next to it (koopman_oracle_c.c: per-row lift of Ksysid.m:1030-1065 + Householder-QR `\`, Ksysid.m:1069).
 *
 * TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY ("the build's CPU backend of the same C ABI restating the
 * reference algorithm", run on the GPU box's host cores beside the GPU number).  It lives under oracle/, is built into
 * oracle/libkoopman_cpu.so and is loaded by bench.py's cpu_baseline leg and by tests/test_oracle_c.py - never by the package:
 * libkoopman_hip.so has no CPU path and fails loudly without a GPU.
 *
 * Entry points (signatures are the header's own - this file includes it, a mismatch does not compile):
 *   kp_create / kp_destroy / kp_last_error / kp_device_count (0: no device is involved)
 *   kp_basis_create (monomial blocks, no pcs) / kp_basis_dims / kp_basis_destroy
 *   kp_snapshots_upload / kp_snapshots_destroy     (host copies)
 *   kp_lift                                        (lift.full / econ_full / the rows of Px)
 *   kp_fit (least squares, K_out required) / kp_fit_gram
 * Everything else of the header is GPU-only and not exported here.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "koopman_hip.h"

int ko_lift_rows(int model_type, int nzeta, int m, const uint8_t* exps, int nmono, const double* zeta, const double* u, int64_t Ns, double* P);
int ko_qr_lstsq(double* A, int64_t rows, int n, double* B, int nrhs, double* X);

struct kp_ctx { char err[256]; };
struct kp_basis { kp_ctx* ctx; int model_type, nzeta, m, nvars, nmono, N, W; uint8_t* exps; };
struct kp_snapshots { kp_ctx* ctx; int64_t Ns; int nzeta, m; double *alpha, *beta, *u; };

static char g_err[256] = "";
static int fail(kp_ctx* c, int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  if (c) snprintf(c->err, sizeof c->err, "%s", msg);
  return code;
}

int kp_device_count(int* count) { if (count) *count = 0; return KP_OK; }
int kp_create(int device_id, kp_ctx** ctx) {
  (void)device_id;
  if (!ctx) return KP_ERR_ARG;
  *ctx = (kp_ctx*)calloc(1, sizeof(kp_ctx));
  return *ctx ? KP_OK : KP_ERR_HIP;
}
int kp_destroy(kp_ctx* ctx) { free(ctx); return KP_OK; }
const char* kp_last_error(const kp_ctx* ctx) { return ctx ? ctx->err : g_err; }

int kp_basis_create(kp_ctx* ctx, const kp_basis_desc* d, kp_basis** out) {
  if (!ctx || !d || !out) return KP_ERR_ARG;
  *out = NULL;
  if (d->k_pcs > 0) return fail(ctx, KP_ERR_ARG, "CPU baseline: dictionaries with a pcs projection are not restated here");
  int nmono = 0;
  for (int b = 0; b < d->n_blocks; ++b) {
    if (d->block_type[b] != KP_BLOCK_POLY) return fail(ctx, KP_ERR_ARG, "CPU baseline: monomial blocks only");
    nmono += d->block_count[b];
  }
  kp_basis* B = (kp_basis*)calloc(1, sizeof(kp_basis));
  if (!B) return KP_ERR_HIP;
  B->ctx = ctx; B->model_type = d->model_type; B->nzeta = d->nzeta; B->m = d->m;
  B->nvars = d->model_type == KP_MODEL_NONLINEAR ? d->nzeta + d->m : d->nzeta;
  B->nmono = nmono;
  B->N = B->nvars + nmono + 1;                                     /* [v ; monomials ; 1], Ksysid.m:484-505 */
  B->W = d->model_type == KP_MODEL_LINEAR ? B->N + d->m : d->model_type == KP_MODEL_BILINEAR ? B->N * (d->m + 1) : B->N;
  B->exps = (uint8_t*)malloc((size_t)(nmono > 0 ? nmono : 1) * B->nvars);
  if (!B->exps) { free(B); return KP_ERR_HIP; }
  if (nmono) memcpy(B->exps, d->poly_exps, (size_t)nmono * B->nvars);
  *out = B;
  return KP_OK;
}
int kp_basis_destroy(kp_basis* b) { if (b) { free(b->exps); free(b); } return KP_OK; }
int kp_basis_dims(const kp_basis* b, int* nvars, int* nfull, int* N, int* W) {
  if (!b) return KP_ERR_ARG;
  if (nvars) *nvars = b->nvars;
  if (nfull) *nfull = b->N;
  if (N) *N = b->N;
  if (W) *W = b->W;
  return KP_OK;
}

static double* dup(const double* src, size_t n) {
  double* p = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
  if (p && n) memcpy(p, src, n * sizeof(double));
  return p;
}
int kp_snapshots_upload(kp_ctx* ctx, const double* alpha, const double* beta, const double* u, int64_t Ns, int nzeta, int m, kp_snapshots** out) {
  if (!ctx || !out || Ns < 0 || nzeta < 1 || m < 0 || (Ns > 0 && (!alpha || !beta || (m > 0 && !u)))) return fail(ctx, KP_ERR_ARG, "kp_snapshots_upload: bad argument");
  kp_snapshots* s = (kp_snapshots*)calloc(1, sizeof(kp_snapshots));
  if (!s) return KP_ERR_HIP;
  s->ctx = ctx; s->Ns = Ns; s->nzeta = nzeta; s->m = m;
  s->alpha = dup(alpha, (size_t)Ns * nzeta); s->beta = dup(beta, (size_t)Ns * nzeta); s->u = dup(u, (size_t)Ns * m);
  if (!s->alpha || !s->beta || !s->u) { kp_snapshots_destroy(s); return KP_ERR_HIP; }
  *out = s;
  return KP_OK;
}
int kp_snapshots_destroy(kp_snapshots* s) { if (s) { free(s->alpha); free(s->beta); free(s->u); free(s); } return KP_OK; }

int kp_lift(kp_ctx* ctx, const kp_basis* b, int what, const double* zeta, const double* u, int64_t rows, double* out) {
  if (!ctx || !b || !zeta || !out || rows < 0) return fail(ctx, KP_ERR_ARG, "kp_lift: bad argument");
  if (what == KP_LIFT_ROW) {
    if (b->m > 0 && !u) return fail(ctx, KP_ERR_ARG, "kp_lift: u required for the rows of Px");
    return ko_lift_rows(b->model_type, b->nzeta, b->m, b->exps, b->nmono, zeta, u, rows, out) ? KP_ERR_ARG : KP_OK;
  }
  /* lift.full = lift.econ_full without pcs: the first N columns of the row (for 'nonlinear' the row itself) */
  if (b->model_type == KP_MODEL_NONLINEAR) {
    if (!u) return fail(ctx, KP_ERR_ARG, "kp_lift: u required ('nonlinear' dictionaries lift [zeta; u])");
    return ko_lift_rows(b->model_type, b->nzeta, b->m, b->exps, b->nmono, zeta, u, rows, out) ? KP_ERR_ARG : KP_OK;
  }
  return ko_lift_rows(KP_MODEL_NONLINEAR, b->nzeta, 0, b->exps, b->nmono, zeta, NULL, rows, out) ? KP_ERR_ARG : KP_OK;
}

/* Px, Py of Ksysid.get_Koopman (Ksysid.m:1030-1065), caller frees */
static int lift_pairs(kp_ctx* ctx, const kp_basis* b, const kp_snapshots* s, double** Px, double** Py) {
  if (s->nzeta != b->nzeta || s->m != b->m) return fail(ctx, KP_ERR_ARG, "kp_fit: snapshot/basis dimension mismatch");
  *Px = (double*)malloc((size_t)(s->Ns > 0 ? s->Ns : 1) * b->W * sizeof(double));
  *Py = (double*)malloc((size_t)(s->Ns > 0 ? s->Ns : 1) * b->W * sizeof(double));
  if (!*Px || !*Py) { free(*Px); free(*Py); return fail(ctx, KP_ERR_HIP, "kp_fit: out of memory"); }
  int rc = ko_lift_rows(b->model_type, b->nzeta, b->m, b->exps, b->nmono, s->alpha, s->u, s->Ns, *Px);
  if (!rc) rc = ko_lift_rows(b->model_type, b->nzeta, b->m, b->exps, b->nmono, s->beta, s->u, s->Ns, *Py);
  if (rc) { free(*Px); free(*Py); return fail(ctx, KP_ERR_ARG, "kp_fit: lift failed"); }
  return KP_OK;
}

int kp_fit(kp_ctx* ctx, const kp_basis* b, const kp_snapshots* s, const double* lasso, int n_lasso, double* K_out) {
  if (!ctx || !b || !s || n_lasso < 1 || !K_out) return fail(ctx, KP_ERR_ARG, "kp_fit (CPU baseline): K_out is required");
  for (int i = 0; i < n_lasso; ++i)
    if (lasso && lasso[i] < 1e6) return fail(ctx, KP_ERR_ARG, "CPU baseline: the least-squares branch only (Ksysid.m:1068-1069)");
  double *Px, *Py;
  int rc = lift_pairs(ctx, b, s, &Px, &Py);
  if (rc) return rc;
  rc = ko_qr_lstsq(Px, s->Ns, b->W, Py, b->W, K_out);              /* K = Px \ Py: Householder QR, as mldivide */
  free(Px); free(Py);
  if (rc) return fail(ctx, KP_ERR_NOT_SPD, "kp_fit (CPU baseline): rank-deficient dictionary");
  for (int i = 1; i < n_lasso; ++i) memcpy(K_out + (size_t)i * b->W * b->W, K_out, (size_t)b->W * b->W * sizeof(double));
  return KP_OK;
}

int kp_fit_gram(kp_ctx* ctx, const kp_basis* b, const kp_snapshots* s, double* G, double* C) {
  if (!ctx || !b || !s) return fail(ctx, KP_ERR_ARG, "kp_fit_gram: NULL handle");
  double *Px, *Py;
  int rc = lift_pairs(ctx, b, s, &Px, &Py);
  if (rc) return rc;
  const int W = b->W;
  const int64_t Ns = s->Ns;
#pragma omp parallel for schedule(dynamic)
  for (int j = 0; j < W; ++j)
    for (int i = 0; i < W; ++i) {
      double g = 0.0, c = 0.0;
      const double *xi = Px + (size_t)i * Ns, *xj = Px + (size_t)j * Ns, *yj = Py + (size_t)j * Ns;
      for (int64_t k = 0; k < Ns; ++k) { g += xi[k] * xj[k]; c += xi[k] * yj[k]; }
      if (G) G[(size_t)j * W + i] = g;                               /* PxTPx, Ksysid.m:1114 */
      if (C) C[(size_t)j * W + i] = c;                               /* PxTPy, :1125 */
    }
  free(Px); free(Py);
  return KP_OK;
}

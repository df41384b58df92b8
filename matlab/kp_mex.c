/* kp_mex.c - MEX gateway between MATLAB and libkoopman_hip.so (C ABI: include/koopman_hip.h).
 *
 * NOT compiled in the build image (no MATLAB, no mex.h there); shipped so that the drop-in claim is inspectable.
 * Build on a MATLAB host:   mex -I../include kp_mex.c -L../koopman-realizations_amd -lkoopman_hip
 *
 * One entry point, dispatched on a command string; opaque handles travel as uint64 scalars.  MATLAB arrays are
 * column-major doubles, which is the library's own layout, so matrices are passed without copies or transposes.
 *
 *   h   = kp_mex('create', device_id)                                   kp_create: the device's shared context (+1 reference)
 *         kp_mex('destroy', h)                                          drops a reference; kp_destroy at the last one
 *   b   = kp_mex('basis_create', h, desc)                               kp_basis_create; desc: struct with fields model_type,
 *                                                                       nzeta, m, block_type (int32 row), block_count (int32 row),
 *                                                                       poly_exps (uint8, nvars x rows), gauss_centres, pcs
 *   d   = kp_mex('basis_dims', b)                                       [nvars nfull N W]
 *   P   = kp_mex('lift', h, b, what, zeta, u)                           kp_lift (what: 0 full, 1 econ, 2 row of Px)
 *   s   = kp_mex('snapshots_upload', h, alpha, beta, u)                 kp_snapshots_upload
 *         kp_mex('snapshots_update', h, s, alpha, beta, u)                kp_snapshots_update (refill in place, returns once staged)
 *   s   = kp_mex('snapshots_resident', h, alpha, beta, u)               the context's own resident object, refilled (owned by the MEX file)
 *         kp_mex('snapshots_destroy', s) / kp_mex('basis_destroy', b) / kp_mex('mpc_destroy', m)
 *   K   = kp_mex('fit', h, b, s, lasso)                                 kp_fit: W x W x numel(lasso)   (get_Koopman, train_models)
 *   [G,C] = kp_mex('fit_gram', h, b, s)                                 kp_fit_gram
 *   K   = kp_mex('fit_solve', h, G, C) / kp_mex('fit_lasso', h, G, C, t)  kp_fit_solve / kp_fit_lasso on given Grams
 *   r   = kp_mex('last_rank', h)                                        kp_fit_last_rank
 *   [A,B,M] = kp_mex('model_project', h, K, G, C, N, m)                 kp_model_project (get_model, Ksysid.m:1206-1225)
 *   Y   = kp_mex('rollout', h, model_type, A, B, z0, U, n_out)          kp_rollout (val_model / val_BLmodel)
 *   m   = kp_mex('mpc_create', h, model_type, A, B, Np, proj, q_run, q_term, r, lo, hi, slope, smooth)
 *         kp_mex('mpc_set_state_bounds', m, lo, hi)
 *   [U,z] = kp_mex('mpc_step_zeta', m, b, zeta, u_prev, Yr, iters)      kp_mpc_step_zeta; U is NaN when the QP failed
 *   U   = kp_mex('mpc_step', m, z, u_prev, Yr, iters)                   kp_mpc_step (lifted state given: loaded models)
 *   x   = kp_mex('qp_solve', h, H, f, A, b)                             kp_qp_solve (signature of quadprog_gurobi.m:1)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "koopman_hip.h"
#include "mex.h"

/* ONE context per device, shared by every KsysidHip / KmpcHip object of the MATLAB session and reference counted:
 * value-class objects are copied and dropped freely (a loop like evaluate_rand_models.m builds a Ksysid per system and
 * degree), so a context per object would leak a HIP context, its streams and its workspaces per object.  'create' hands
 * out the device's context (creating it on first use), 'destroy' drops one reference; mexAtExit releases what is left. */
#define MAXDEV 16
static kp_ctx* g_ctx[MAXDEV];
static int g_refs[MAXDEV];
/* one resident snapshot object per context, refilled in place ('snapshots_resident'): a value-class method such as
 * KsysidHip.get_Koopman cannot keep a handle between calls, the locked MEX file can */
static kp_snapshots* g_snaps[MAXDEV];
static int g_snaps_nz[MAXDEV], g_snaps_m[MAXDEV];

static void release_slot(int i) {
  if (!g_ctx[i]) return;
  if (g_snaps[i]) kp_snapshots_destroy(g_snaps[i]);
  g_snaps[i] = NULL;
  kp_destroy(g_ctx[i]);
  g_ctx[i] = NULL;
  g_refs[i] = 0;
}

static void at_exit(void) {
  for (int i = 0; i < MAXDEV; ++i) release_slot(i);
}

static int slot_of(const kp_ctx* c) {
  for (int i = 0; i < MAXDEV; ++i)
    if (g_ctx[i] == c && c) return i;
  return -1;
}

static void* get_handle(const mxArray* a) {
  if (!mxIsUint64(a) || mxGetNumberOfElements(a) != 1) mexErrMsgIdAndTxt("kp:handle", "handle must be a uint64 scalar");
  return (void*)(uintptr_t)(*(uint64_t*)mxGetData(a));
}
static mxArray* put_handle(void* p) {
  mxArray* a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *(uint64_t*)mxGetData(a) = (uint64_t)(uintptr_t)p;
  return a;
}
static const double* dbl(const mxArray* a) {
  if (mxIsEmpty(a)) return NULL;
  if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("kp:type", "real double array expected");
  return mxGetPr(a);
}
static void check(int rc, const kp_ctx* ctx) {
  if (rc != KP_OK) mexErrMsgIdAndTxt("kp:error", "libkoopman_hip error %d: %s", rc, kp_last_error(ctx));
}
static const mxArray* field(const mxArray* s, const char* name) {
  const mxArray* f = mxGetField(s, 0, name);
  if (!f) mexErrMsgIdAndTxt("kp:desc", "descriptor field '%s' missing", name);
  return f;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  char cmd[64];
  if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("kp:usage", "kp_mex(command, ...)");
  if (!mexIsLocked()) { mexLock(); mexAtExit(at_exit); }

  if (!strcmp(cmd, "create")) {
    const int dev = nrhs > 1 ? (int)mxGetScalar(prhs[1]) : 0;
    if (dev < 0 || dev >= MAXDEV) mexErrMsgIdAndTxt("kp:usage", "create: device id out of range");
    if (!g_ctx[dev]) check(kp_create(dev, &g_ctx[dev]), NULL);
    ++g_refs[dev];
    plhs[0] = put_handle(g_ctx[dev]);
  } else if (!strcmp(cmd, "destroy")) {
    const int slot = slot_of((kp_ctx*)get_handle(prhs[1]));
    if (slot < 0) mexErrMsgIdAndTxt("kp:handle", "destroy: unknown context");
    if (--g_refs[slot] <= 0) release_slot(slot);
  } else if (!strcmp(cmd, "basis_create")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const mxArray* d = prhs[2];
    kp_basis_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.model_type = (int32_t)mxGetScalar(field(d, "model_type"));
    desc.nzeta = (int32_t)mxGetScalar(field(d, "nzeta"));
    desc.m = (int32_t)mxGetScalar(field(d, "m"));
    const mxArray* bt = field(d, "block_type");
    const mxArray* bc = field(d, "block_count");
    if (!mxIsInt32(bt) || !mxIsInt32(bc)) mexErrMsgIdAndTxt("kp:desc", "block_type / block_count must be int32");
    desc.n_blocks = (int32_t)mxGetNumberOfElements(bt);
    desc.block_type = (const int32_t*)mxGetData(bt);
    desc.block_count = (const int32_t*)mxGetData(bc);
    const mxArray* pe = field(d, "poly_exps");
    if (!mxIsEmpty(pe) && !mxIsUint8(pe)) mexErrMsgIdAndTxt("kp:desc", "poly_exps must be uint8 (nvars x rows)");
    desc.poly_exps = mxIsEmpty(pe) ? NULL : (const uint8_t*)mxGetData(pe);
    desc.gauss_centres = dbl(field(d, "gauss_centres"));
    const mxArray* pcs = field(d, "pcs");
    desc.k_pcs = mxIsEmpty(pcs) ? 0 : (int32_t)mxGetN(pcs);
    desc.pcs = dbl(pcs);
    kp_basis* b = NULL;
    check(kp_basis_create(c, &desc, &b), c);
    plhs[0] = put_handle(b);
  } else if (!strcmp(cmd, "basis_dims")) {
    int v[4];
    check(kp_basis_dims((kp_basis*)get_handle(prhs[1]), &v[0], &v[1], &v[2], &v[3]), NULL);
    plhs[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
    for (int i = 0; i < 4; ++i) mxGetPr(plhs[0])[i] = v[i];
  } else if (!strcmp(cmd, "basis_destroy")) {
    kp_basis_destroy((kp_basis*)get_handle(prhs[1]));
  } else if (!strcmp(cmd, "snapshots_destroy")) {
    kp_snapshots_destroy((kp_snapshots*)get_handle(prhs[1]));
  } else if (!strcmp(cmd, "mpc_destroy")) {
    kp_mpc_destroy((kp_mpc*)get_handle(prhs[1]));
  } else if (!strcmp(cmd, "lift")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    kp_basis* b = (kp_basis*)get_handle(prhs[2]);
    const int what = (int)mxGetScalar(prhs[3]);
    int nv, nf, N, W;
    check(kp_basis_dims(b, &nv, &nf, &N, &W), c);
    const mwSize rows = mxGetM(prhs[4]);
    const int width = what == KP_LIFT_FULL ? nf : what == KP_LIFT_ECON ? N : W;
    plhs[0] = mxCreateDoubleMatrix(rows, (mwSize)width, mxREAL);
    check(kp_lift(c, b, what, dbl(prhs[4]), nrhs > 5 ? dbl(prhs[5]) : NULL, (int64_t)rows, mxGetPr(plhs[0])), c);
  } else if (!strcmp(cmd, "snapshots_upload")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    kp_snapshots* s = NULL;
    check(kp_snapshots_upload(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), (int64_t)mxGetM(prhs[2]), (int)mxGetN(prhs[2]),
                              (int)mxGetN(prhs[4]), &s), c);
    plhs[0] = put_handle(s);
  } else if (!strcmp(cmd, "snapshots_resident")) {
    /* s = kp_mex('snapshots_resident', h, alpha, beta, u): the context's resident object, refilled in place
     * (kp_snapshots_update: no hipMalloc / hipFree per call, staged chunked transfer); owned by the MEX file - do not destroy */
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int nz = (int)mxGetN(prhs[2]), m_ = (int)mxGetN(prhs[4]);
    const int slot = slot_of(c);
    if (slot < 0) mexErrMsgIdAndTxt("kp:handle", "snapshots_resident: unknown context");
    if (g_snaps[slot] && (g_snaps_nz[slot] != nz || g_snaps_m[slot] != m_)) {
      kp_snapshots_destroy(g_snaps[slot]);
      g_snaps[slot] = NULL;
    }
    if (!g_snaps[slot]) {
      check(kp_snapshots_upload(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), (int64_t)mxGetM(prhs[2]), nz, m_, &g_snaps[slot]), c);
      g_snaps_nz[slot] = nz;
      g_snaps_m[slot] = m_;
    } else {
      check(kp_snapshots_update(c, g_snaps[slot], dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), (int64_t)mxGetM(prhs[2])), c);
    }
    plhs[0] = put_handle(g_snaps[slot]);
  } else if (!strcmp(cmd, "snapshots_update")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    kp_snapshots* s = (kp_snapshots*)get_handle(prhs[2]);
    check(kp_snapshots_update(c, s, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), (int64_t)mxGetM(prhs[3])), c);
  } else if (!strcmp(cmd, "fit")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    kp_basis* b = (kp_basis*)get_handle(prhs[2]);
    int nv, nf, N, W;
    check(kp_basis_dims(b, &nv, &nf, &N, &W), c);
    const int nl = (int)mxGetNumberOfElements(prhs[4]);
    mwSize dims[3] = {(mwSize)W, (mwSize)W, (mwSize)nl};
    plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    check(kp_fit(c, b, (kp_snapshots*)get_handle(prhs[3]), dbl(prhs[4]), nl, mxGetPr(plhs[0])), c);
    if (kp_last_error(c)[0] == 'w') mexWarnMsgIdAndTxt("kp:rankDeficient", "%s", kp_last_error(c));   /* like mldivide */
  } else if (!strcmp(cmd, "fit_gram")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    kp_basis* b = (kp_basis*)get_handle(prhs[2]);
    int nv, nf, N, W;
    check(kp_basis_dims(b, &nv, &nf, &N, &W), c);
    plhs[0] = mxCreateDoubleMatrix(W, W, mxREAL);
    mxArray* Cm = mxCreateDoubleMatrix(W, W, mxREAL);
    check(kp_fit_gram(c, b, (kp_snapshots*)get_handle(prhs[3]), mxGetPr(plhs[0]), mxGetPr(Cm)), c);
    if (nlhs > 1) plhs[1] = Cm; else mxDestroyArray(Cm);
  } else if (!strcmp(cmd, "fit_solve")) {
    /* K = kp_mex('fit_solve', h, G, C): G K = C on the device (kp_fit_solve) */
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int W = (int)mxGetM(prhs[2]), nc = (int)mxGetN(prhs[3]);
    plhs[0] = mxCreateDoubleMatrix(W, nc, mxREAL);
    check(kp_fit_solve(c, dbl(prhs[2]), dbl(prhs[3]), W, nc, mxGetPr(plhs[0])), c);
  } else if (!strcmp(cmd, "fit_lasso")) {
    /* K = kp_mex('fit_lasso', h, G, C, t): min 1/2 ||Px K - Py||^2 s.t. ||vec K||_1 <= t from the Grams (solve_KoopmanQP,
     * Ksysid.m:1095-1176; the caller forms t = lasso * N, :996) */
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int W = (int)mxGetM(prhs[2]), nc = (int)mxGetN(prhs[3]);
    int iters = 0;
    plhs[0] = mxCreateDoubleMatrix(W, nc, mxREAL);
    check(kp_fit_lasso(c, dbl(prhs[2]), dbl(prhs[3]), W, nc, mxGetScalar(prhs[4]), 20000, 1e-10, mxGetPr(plhs[0]), &iters), c);
  } else if (!strcmp(cmd, "last_rank")) {
    int r = -1;
    check(kp_fit_last_rank((kp_ctx*)get_handle(prhs[1]), &r), NULL);
    plhs[0] = mxCreateDoubleScalar(r);
  } else if (!strcmp(cmd, "model_project")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int N = (int)mxGetScalar(prhs[5]), m = (int)mxGetScalar(prhs[6]);
    plhs[0] = mxCreateDoubleMatrix(N, N, mxREAL);
    mxArray* B = mxCreateDoubleMatrix(N, m, mxREAL);
    mxArray* M = mxCreateDoubleMatrix(N, N, mxREAL);
    check(kp_model_project(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), N, m, mxGetPr(plhs[0]), mxGetPr(B), mxGetPr(M)), c);
    if (nlhs > 1) plhs[1] = B; else mxDestroyArray(B);
    if (nlhs > 2) plhs[2] = M; else mxDestroyArray(M);
  } else if (!strcmp(cmd, "rollout")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int mt = (int)mxGetScalar(prhs[2]);
    const int N = (int)mxGetM(prhs[3]), T = (int)mxGetM(prhs[6]), m = (int)mxGetN(prhs[6]), n_out = (int)mxGetScalar(prhs[7]);
    plhs[0] = mxCreateDoubleMatrix(T, n_out, mxREAL);
    check(kp_rollout(c, mt, 1, dbl(prhs[3]), dbl(prhs[4]), N, m, dbl(prhs[5]), dbl(prhs[6]), T, n_out, mxGetPr(plhs[0])), c);
  } else if (!strcmp(cmd, "mpc_create")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int mt = (int)mxGetScalar(prhs[2]);
    const int N = (int)mxGetM(prhs[3]), Np = (int)mxGetScalar(prhs[5]), nproj = (int)mxGetM(prhs[6]);
    const int m = (int)mxGetNumberOfElements(prhs[9]);
    const double slope = mxIsEmpty(prhs[12]) ? NAN : mxGetScalar(prhs[12]);
    const double smooth = mxIsEmpty(prhs[13]) ? NAN : mxGetScalar(prhs[13]);
    kp_mpc* mp = NULL;
    check(kp_mpc_create(c, mt, dbl(prhs[3]), dbl(prhs[4]), N, m, Np, dbl(prhs[6]), nproj, mxGetScalar(prhs[7]), mxGetScalar(prhs[8]),
                        dbl(prhs[9]), dbl(prhs[10]), dbl(prhs[11]), slope, smooth, &mp), c);
    plhs[0] = put_handle(mp);
  } else if (!strcmp(cmd, "mpc_set_state_bounds")) {
    check(kp_mpc_set_state_bounds((kp_mpc*)get_handle(prhs[1]), (int)mxGetNumberOfElements(prhs[2]), dbl(prhs[2]), dbl(prhs[3])), NULL);
  } else if (!strcmp(cmd, "mpc_step_zeta")) {
    kp_mpc* mp = (kp_mpc*)get_handle(prhs[1]);
    kp_basis* b = (kp_basis*)get_handle(prhs[2]);
    int nvar, nrows, nv, nf, N, W, status = 0;
    check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
    check(kp_basis_dims(b, &nv, &nf, &N, &W), NULL);
    const int m = (int)mxGetNumberOfElements(prhs[4]), Np = nvar / m;
    plhs[0] = mxCreateDoubleMatrix(Np, m, mxREAL);
    mxArray* z = mxCreateDoubleMatrix(N, 1, mxREAL);
    /* QP failure: U comes back NaN and the call itself succeeds - Ksim.m:220-222 tests any(isnan(U)) */
    check(kp_mpc_step_zeta(mp, b, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), nrhs > 6 ? (int)mxGetScalar(prhs[6]) : 1, mxGetPr(plhs[0]),
                           mxGetPr(z), &status), NULL);
    if (nlhs > 1) plhs[1] = z; else mxDestroyArray(z);
  } else if (!strcmp(cmd, "mpc_step")) {
    /* U = kp_mex('mpc_step', m, z, u_prev, Yr, iters): the step from an already lifted state (loaded models lift with the
     * current load estimate on the host side of the boundary, Kmpc.m:347-348) */
    kp_mpc* mp = (kp_mpc*)get_handle(prhs[1]);
    int nvar, nrows, status = 0;
    check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
    const int m = (int)mxGetNumberOfElements(prhs[3]), Np = nvar / m;
    plhs[0] = mxCreateDoubleMatrix(Np, m, mxREAL);
    check(kp_mpc_step(mp, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), nrhs > 5 ? (int)mxGetScalar(prhs[5]) : 1, mxGetPr(plhs[0]), &status), NULL);
  } else if (!strcmp(cmd, "qp_solve")) {
    kp_ctx* c = (kp_ctx*)get_handle(prhs[1]);
    const int n = (int)mxGetM(prhs[2]), mr = (int)mxGetM(prhs[4]);
    int status = 0;
    plhs[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
    check(kp_qp_solve(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), n, mr, mxGetPr(plhs[0]), &status), c);
  } else {
    mexErrMsgIdAndTxt("kp:usage", "unknown command '%s'", cmd);
  }
}

/* kp_mex.c - MEX gateway between MATLAB and libkoopman_hip.so (C ABI: include/koopman_hip.h).
 *
 * Build on a MATLAB host:   mex -I../include kp_mex.c -L../koopman-realizations_amd -lkoopman_hip
 * In the build image (no MATLAB) the same file is compiled against the functional mex.h stand-in of tests/mex_shim/ and
 * EXECUTED by tests/test_mex_gateway.py: every command below runs on the GPU box and is compared with the direct C-ABI call.
 *
 * One entry point, dispatched on a command string through the table at the end of this file (name, number of arguments
 * after the command, number of outputs, handler): kp_mex('commands') returns that table, and tests/test_host_logic.py checks
 * every kp_mex(...) call in the .m files of matlab/ against it.  Opaque handles travel as uint64 scalars.  MATLAB arrays are column-major
 * doubles, which is the library's own layout, so matrices are passed without copies or transposes; stacks of matrices are
 * 3-D arrays (W x W x count).
 *
 * context     h = kp_mex('create' [, device_id])          the device's shared context (+1 reference)      kp_create
 *             kp_mex('destroy', h)                         drops a reference; kp_destroy at the last one
 *             n = kp_mex('device_count')                                                                   kp_device_count
 *             [name, num_cu, hbm_bytes] = kp_mex('device_info', h)                                         kp_device_info
 *             s = kp_mex('last_error' [, h])                                                               kp_last_error
 *             ms = kp_mex('timer_get', h, which)                                                           kp_timer_get
 *             kp_mex('synchronize', h)                                                                     kp_synchronize
 *             [tab, names] = kp_mex('commands')            tab(i,:) = [nrhs_min nrhs_max nlhs_max], names newline-joined
 * dictionary  b = kp_mex('basis_create', h, desc)          desc: struct with fields model_type, nzeta, m, block_type
 *                                                          (int32 row), block_count (int32 row), poly_exps (uint8, nvars x
 *                                                          rows), gauss_centres, pcs                       kp_basis_create
 *             d = kp_mex('basis_dims', b)                  [nvars nfull N W]                               kp_basis_dims
 *             d = kp_mex('basis_desc_dims', desc)          the same from the descriptor alone              kp_basis_desc_dims
 *             kp_mex('basis_destroy', b)
 *             [V, lam, sweeps] = kp_mex('sym_eig', h, S)                                                   kp_sym_eig
 *             P = kp_mex('lift', h, b, what, zeta [, u])   what: 0 full, 1 econ, 2 row of Px               kp_lift
 * snapshots   s = kp_mex('snapshots_upload', h, alpha, beta, u)                                            kp_snapshots_upload
 *             kp_mex('snapshots_update', h, s, alpha, beta, u)                                             kp_snapshots_update
 *             s = kp_mex('snapshots_resident', h, alpha, beta, u)   the context's own object, refilled (owned by the MEX file)
 *             kp_mex('snapshots_destroy', s)
 * fit         K = kp_mex('fit', h, b, s, lasso)            W x W x numel(lasso)   (get_Koopman, train_models)   kp_fit
 *             kp_mex('fit_async', h, b, s)                 enqueue one least-squares fit (K stays on the device)  kp_fit, K_out = NULL
 *             K = kp_mex('fit_get_K', h, index, W)         fit `index` of the asynchronous batch           kp_fit_get_K
 *             kp_mex('fit_async_slots', h, n)                                                              kp_fit_async_slots
 *             [G, C] = kp_mex('fit_gram', h, b, s)                                                         kp_fit_gram
 *             K = kp_mex('fit_solve', h, G, C)                                                             kp_fit_solve
 *             K = kp_mex('fit_lasso', h, G, C, t)                                                          kp_fit_lasso
 *             [K, iters] = kp_mex('fit_lasso_batch', h, G, C, t)     K: W x ncols x numel(t)               kp_fit_lasso_batch
 *             K = kp_mex('fit_refine', h, b, s, K, steps)                                                  kp_fit_refine
 *             r = kp_mex('last_rank', h)                                                                   kp_fit_last_rank
 *             q = kp_mex('last_pivot_ratio', h)                                                            kp_fit_last_pivot_ratio
 *             [K, G, C, status] = kp_mex('fit_batch', h, b, s, nb, Ns_each)                                kp_fit_batch
 *             K = kp_mex('fit_sharded', h, b, s, lasso)    [G, C] = kp_mex('fit_gram_sharded', h, b, s)    kp_fit_(gram_)sharded
 * models      [MA, MB, M] = kp_mex('model_project', h, K, G, C, N, m)   the projected M*A, M*B (Ksysid.m:1224-1225) and M   kp_model_project
 *             [A, B, M, status] = kp_mex('model_project_batch', h, K, G, C, N, m)   stacks W x W x nb      kp_model_project_batch
 *             Y = kp_mex('rollout', h, model_type, A, B, z0, U, n_out)       z0 N x batch, U T x m x batch kp_rollout
 *             Z = kp_mex('rollout_nl', h, b, Kf, zeta0, U)                   zeta0 nzeta x batch           kp_rollout_nl
 * sweep       t = kp_mex('traj_upload', h, Y, U, Yv, Uv, ntrials)   Y rows x n x nb, U rows x m x nb, Yv Tv x n x nb, ...  kp_traj_upload
 *             t = kp_mex('traj_create', h, nb, ntrials, T, n, m, Tv)   kp_mex('traj_put', t, which, block)   kp_mex('traj_finish', t)
 *             kp_mex('traj_destroy', t)     d = kp_mex('traj_dims', t)     sc = kp_mex('traj_scale', t)    2(n+m) x nb
 *             [err, K, status] = kp_mex('sweep_eval', h, t, b, lasso)        err n x nb                    kp_sweep_eval
 *             [err, status] = kp_mex('sweep_eval_nested', h, t, b, lasso, n_deg)   err n x nb x n_deg      kp_sweep_eval_nested
 *             K = kp_mex('sweep_nested_get_K', h, nb, Wmax, n_deg, deg_index, W)                           kp_sweep_nested_get_K
 * MPC         m = kp_mex('mpc_create', h, model_type, A, B, Np, proj, q_run, q_term, r, lo, hi, slope, smooth)   kp_mpc_create
 *             kp_mex('mpc_set_state_bounds', m, lo, hi)    kp_mex('mpc_destroy', m)    d = kp_mex('mpc_dims', m)   [nvar nrows]
 *             [U, z] = kp_mex('mpc_step_zeta', m, b, zeta, u_prev, Yr [, iters])   U is NaN when the QP failed   kp_mpc_step_zeta
 *             U = kp_mex('mpc_step', m, z, u_prev, Yr [, iters])                                           kp_mpc_step
 *             [U, status] = kp_mex('mpc_step_batch', m, Z, Uprev, YR)    Z N x nb, ..., U Np x m x nb      kp_mpc_step_batch
 *             [Hq, f, Aq, bq] = kp_mex('mpc_last_qp', m)     [us, counts] = kp_mex('mpc_last_profile', m)
 *             us16 = kp_mex('mpc_last_stamps', m)
 *             x = kp_mex('qp_solve', h, H, f, A, b)        signature of quadprog_gurobi.m:1                kp_qp_solve
 * comm        id = kp_mex('comm_unique_id')                uint8 1 x 128                                   kp_comm_unique_id
 *             kp_mex('comm_create', h, id, rank, world)    kp_mex('comm_destroy', h)    kp_mex('comm_abandon', h)
 *             rw = kp_mex('comm_info', h)                  [rank world]
 *             A = kp_mex('comm_allgather', h, v)           numel(v) x world        v = kp_mex('comm_allreduce_sum', h, v)
 *             K = kp_mex('comm_allgather_fit', h, index, W)             W x W x world
 *             K = kp_mex('comm_allgather_fits', h, first, count, W)     W x W x count x world
 *             K = kp_mex('comm_gather_fits', h, root, first, count, W)  the same on rank `root` only, [] elsewhere   kp_comm_gather_fits
 * one caller, several GPUs (no Parallel Computing Toolbox, no second process)
 *             g = kp_mex('multi_create', device_ids)       kp_mex('multi_destroy', g)   n = kp_mex('multi_size', g)
 *             K = kp_mex('multi_fit', g, desc, alpha, beta, u, lasso)           W x W x numel(lasso)       kp_multi_fit
 *             K = kp_mex('multi_fit_sharded', g, desc, alpha, beta, u, lasso)                              kp_multi_fit_sharded
 *             ms = kp_mex('multi_timers', g)               4 x n_dev
 *             t = kp_mex('multi_traj_upload', g, Y, U, Yv, Uv, ntrials)    kp_mex('multi_traj_destroy', t)
 *             [err, status] = kp_mex('multi_sweep_eval_nested', g, t, desc, lasso, n_deg)
 *             m = kp_mex('multi_mpc_create', g, model_type, A, B, Np, proj, q_run, q_term, r, lo, hi, slope, smooth)
 *             kp_mex('multi_mpc_set_state_bounds', m, lo, hi)   kp_mex('multi_mpc_destroy', m)
 *             [U, status] = kp_mex('multi_mpc_step_batch', m, Z, Uprev, YR)
 * Not exposed: kp_host_alloc / kp_host_free / kp_multi_host_alloc / kp_stream / kp_multi_ctx (raw pointers have no MATLAB
 * meaning; results are MATLAB arrays, which the library fills through its own page-locked staging).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
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
/* kp_multi objects of the session (released by mexAtExit when the script did not) */
#define MAXMULTI 8
static kp_multi* g_multi[MAXMULTI];

static void reg_del(void* p);
static void reg_clear(void);
static void release_slot(int i) {
  if (!g_ctx[i]) return;
  if (g_snaps[i]) { reg_del(g_snaps[i]); kp_snapshots_destroy(g_snaps[i]); }
  g_snaps[i] = NULL;
  reg_del(g_ctx[i]);
  kp_destroy(g_ctx[i]);
  g_ctx[i] = NULL;
  g_refs[i] = 0;
}

static void at_exit(void) {
  for (int i = 0; i < MAXMULTI; ++i) {
    if (g_multi[i]) kp_multi_destroy(g_multi[i]);
    g_multi[i] = NULL;
  }
  /* (whatever basis / MPC / trajectory handles the scripts never released go with their contexts' device memory at process
   * exit; the registry only forgets them) */
  for (int i = 0; i < MAXDEV; ++i) release_slot(i);
  reg_clear();
}

static int slot_of(const kp_ctx* c) {
  for (int i = 0; i < MAXDEV; ++i)
    if (g_ctx[i] == c && c) return i;
  return -1;
}

/* ---- registry of live handles -------------------------------------------------------------------------------------------
 * A handle is a raw pointer in a uint64.  MATLAB objects that carry one can be saved and loaded (Ksysid.save_class, Ksysid.m:
 * 436-448), copied into another session, or outlive `clear kp_mex`: the value then names memory that was freed, or never was
 * ours.  Every handle handed out is recorded here, every command that takes one checks it, and a *_destroy command on an
 * unknown value is an error MATLAB can catch - not a free() of a stale pointer. */
enum { H_CTX = 1, H_BASIS, H_SNAPS, H_MPC, H_TRAJ, H_MULTI, H_MTRAJ, H_MMPC };
static const char* const g_kind_name[] = {"?", "context", "basis", "snapshots", "mpc", "trajectories", "multi", "multi trajectories", "multi mpc"};
/* every entry carries the KIND of object it names: a live basis handle given to mpc_step, or a context handle given to
 * basis_destroy, is an error too - not a cast to the wrong struct (and an address the allocator hands out again after a free
 * validates only for the kind it was registered as) */
typedef struct { void* p; int kind; } live_rec;
static live_rec* g_live;
static size_t g_nlive, g_caplive;
static void reg_add(void* p, int kind) {
  if (!p) return;
  if (g_nlive == g_caplive) {
    const size_t cap = g_caplive ? 2 * g_caplive : 64;
    live_rec* q = (live_rec*)realloc(g_live, cap * sizeof *q);
    if (!q) mexErrMsgIdAndTxt("kp:memory", "out of host memory");
    g_live = q;
    g_caplive = cap;
  }
  g_live[g_nlive].p = p;
  g_live[g_nlive++].kind = kind;
}
static int reg_find(const void* p) {
  for (size_t i = 0; i < g_nlive; ++i)
    if (g_live[i].p == p) return (int)i;
  return -1;
}
static void reg_del(void* p) {
  const int i = reg_find(p);
  if (i >= 0) g_live[i] = g_live[--g_nlive];
}
static void reg_clear(void) {
  free(g_live);
  g_live = NULL;
  g_nlive = g_caplive = 0;
}

/* ---- argument helpers ------------------------------------------------------------------------------------------------ */
static void* get_handle(const mxArray* a, int kind) {
  if (!mxIsUint64(a) || mxGetNumberOfElements(a) != 1) mexErrMsgIdAndTxt("kp:handle", "handle must be a uint64 scalar");
  void* p = (void*)(uintptr_t)(*(uint64_t*)mxGetData(a));
  if (!p) mexErrMsgIdAndTxt("kp:handle", "null handle");
  const int i = reg_find(p);
  if (i < 0)
    mexErrMsgIdAndTxt("kp:handle", "unknown or stale handle (released already, loaded from a MAT file, or from before `clear kp_mex`)");
  if (g_live[i].kind != kind)
    mexErrMsgIdAndTxt("kp:handle", "wrong kind of handle: a %s handle where a %s handle is expected", g_kind_name[g_live[i].kind], g_kind_name[kind]);
  return p;
}
/* a *_destroy command: the handle leaves the registry before its memory goes */
static void* take_handle(const mxArray* a, int kind) {
  void* p = get_handle(a, kind);
  reg_del(p);
  return p;
}
static mxArray* put_handle(void* p, int kind) {
  if (reg_find(p) < 0) reg_add(p, kind);                /* (the shared context and the resident snapshot object are handed out repeatedly) */
  mxArray* a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *(uint64_t*)mxGetData(a) = (uint64_t)(uintptr_t)p;
  return a;
}
static const double* dbl(const mxArray* a) {
  if (mxIsEmpty(a)) return NULL;
  if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("kp:type", "real double array expected");
  return mxGetPr(a);
}
static const double* dbl_n(const mxArray* a, size_t n, const char* what) {       /* exactly n doubles */
  const double* p = dbl(a);
  if (mxGetNumberOfElements(a) != n) mexErrMsgIdAndTxt("kp:size", "%s: %d elements expected, got %d", what, (int)n, (int)mxGetNumberOfElements(a));
  return p;
}
static int int_arg(const mxArray* a, const char* what) {
  if (mxGetNumberOfElements(a) != 1 || !mxIsNumeric(a)) mexErrMsgIdAndTxt("kp:type", "%s: numeric scalar expected", what);
  return (int)mxGetScalar(a);
}
static void check(int rc, const kp_ctx* ctx) {
  if (rc != KP_OK) mexErrMsgIdAndTxt("kp:error", "libkoopman_hip error %d: %s", rc, kp_last_error(ctx));
}
static void check_multi(int rc, const kp_multi* g) {
  if (rc != KP_OK) mexErrMsgIdAndTxt("kp:error", "libkoopman_hip error %d: %s", rc, kp_multi_last_error(g));
}
static const mxArray* field(const mxArray* s, const char* name) {
  if (!mxIsStruct(s)) mexErrMsgIdAndTxt("kp:desc", "dictionary descriptor must be a struct");
  const mxArray* f = mxGetField(s, 0, name);
  if (!f) mexErrMsgIdAndTxt("kp:desc", "descriptor field '%s' missing", name);
  return f;
}
/* third dimension of a stack (1 for a matrix) */
static size_t pages(const mxArray* a) {
  const size_t m = mxGetM(a);
  if (mxGetNumberOfDimensions(a) < 3) return 1;
  return m ? mxGetNumberOfElements(a) / (m * mxGetDimensions(a)[1]) : 0;
}
static mxArray* dstack(size_t a, size_t b, size_t c) {
  const mwSize dims[3] = {a, b, c};
  return mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
}
/* int status vector -> double column */
static mxArray* ints_out(const int* v, size_t n, size_t rows, size_t cols) {
  mxArray* a = mxCreateDoubleMatrix(rows, cols, mxREAL);
  for (size_t i = 0; i < n; ++i) mxGetPr(a)[i] = v[i];
  return a;
}
static void set_or_drop(int nlhs, mxArray* plhs[], int k, mxArray* a) {
  if (nlhs > k || k == 0) plhs[k] = a;
  else mxDestroyArray(a);
}
static void basis_desc(const mxArray* d, kp_basis_desc* desc) {
  memset(desc, 0, sizeof *desc);
  desc->model_type = (int32_t)mxGetScalar(field(d, "model_type"));
  desc->nzeta = (int32_t)mxGetScalar(field(d, "nzeta"));
  desc->m = (int32_t)mxGetScalar(field(d, "m"));
  const mxArray* bt = field(d, "block_type");
  const mxArray* bc = field(d, "block_count");
  if ((!mxIsEmpty(bt) && !mxIsInt32(bt)) || (!mxIsEmpty(bc) && !mxIsInt32(bc))) mexErrMsgIdAndTxt("kp:desc", "block_type / block_count must be int32");
  if (mxGetNumberOfElements(bt) != mxGetNumberOfElements(bc)) mexErrMsgIdAndTxt("kp:desc", "block_type and block_count differ in length");
  desc->n_blocks = (int32_t)mxGetNumberOfElements(bt);
  desc->block_type = (const int32_t*)mxGetData(bt);
  desc->block_count = (const int32_t*)mxGetData(bc);
  const mxArray* pe = field(d, "poly_exps");
  if (!mxIsEmpty(pe) && !mxIsUint8(pe)) mexErrMsgIdAndTxt("kp:desc", "poly_exps must be uint8 (nvars x rows)");
  desc->poly_exps = mxIsEmpty(pe) ? NULL : (const uint8_t*)mxGetData(pe);
  desc->gauss_centres = dbl(field(d, "gauss_centres"));
  const mxArray* pcs = field(d, "pcs");
  desc->k_pcs = mxIsEmpty(pcs) ? 0 : (int32_t)mxGetN(pcs);
  desc->pcs = dbl(pcs);
  /* the byte table must hold what the block counts announce (a short table would be read past its end) */
  const int nvars = desc->nzeta + (desc->model_type == KP_MODEL_NONLINEAR ? desc->m : 0);
  size_t rows = 0, cen = 0;
  for (int b = 0; b < desc->n_blocks; ++b) {
    const int t = desc->block_type[b], c = desc->block_count[b];
    if (t == KP_BLOCK_POLY || t == KP_BLOCK_HERMITE) rows += (size_t)c;
    else if (t == KP_BLOCK_FOURIER_SPARSER) rows += 2 * (size_t)c;
    else if (t == KP_BLOCK_GAUSSIAN) cen += (size_t)c;
  }
  if (mxGetNumberOfElements(pe) < rows * (size_t)nvars) mexErrMsgIdAndTxt("kp:desc", "poly_exps holds fewer rows than block_count announces");
  if (mxGetNumberOfElements(field(d, "gauss_centres")) < cen * (size_t)nvars) mexErrMsgIdAndTxt("kp:desc", "gauss_centres holds fewer centres than block_count announces");
  int nfull = 0;
  if (kp_basis_desc_dims(desc, NULL, &nfull, NULL, NULL) != KP_OK) mexErrMsgIdAndTxt("kp:desc", "invalid dictionary descriptor");
  if (desc->k_pcs > 0 && (int)mxGetM(pcs) != nfull) mexErrMsgIdAndTxt("kp:desc", "pcs must have %d rows (the full dictionary)", nfull);
}
static void basis_wn(const kp_basis* b, const kp_ctx* c, int* nv, int* nf, int* N, int* W) { check(kp_basis_dims(b, nv, nf, N, W), c); }

#define ARGS int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]
#define UNUSED (void)nlhs; (void)plhs; (void)nrhs; (void)prhs
#define CTX(i) ((kp_ctx*)get_handle(prhs[i], H_CTX))

/* ---- context ----------------------------------------------------------------------------------------------------------- */
static void c_create(ARGS) {
  UNUSED;
  const int dev = nrhs > 1 ? int_arg(prhs[1], "device_id") : 0;
  if (dev < 0 || dev >= MAXDEV) mexErrMsgIdAndTxt("kp:usage", "create: device id out of range");
  if (!g_ctx[dev]) check(kp_create(dev, &g_ctx[dev]), NULL);
  ++g_refs[dev];
  plhs[0] = put_handle(g_ctx[dev], H_CTX);
}
static void c_destroy(ARGS) {
  UNUSED;
  const int slot = slot_of(CTX(1));
  if (slot < 0) mexErrMsgIdAndTxt("kp:handle", "destroy: unknown context");
  if (--g_refs[slot] <= 0) release_slot(slot);
}
static void c_device_count(ARGS) {
  UNUSED;
  int n = 0;
  check(kp_device_count(&n), NULL);
  plhs[0] = mxCreateDoubleScalar(n);
}
static void c_device_info(ARGS) {
  UNUSED;
  char name[256];
  int ncu = 0;
  int64_t hbm = 0;
  kp_ctx* c = CTX(1);
  check(kp_device_info(c, name, sizeof name, &ncu, &hbm), c);
  plhs[0] = mxCreateString(name);
  set_or_drop(nlhs, plhs, 1, mxCreateDoubleScalar(ncu));
  set_or_drop(nlhs, plhs, 2, mxCreateDoubleScalar((double)hbm));
}
static void c_last_error(ARGS) {
  UNUSED;
  plhs[0] = mxCreateString(kp_last_error(nrhs > 1 ? CTX(1) : NULL));
}
static void c_timer_get(ARGS) {
  UNUSED;
  double ms = 0;
  kp_ctx* c = CTX(1);
  check(kp_timer_get(c, int_arg(prhs[2], "which"), &ms), c);
  plhs[0] = mxCreateDoubleScalar(ms);
}
static void c_synchronize(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  check(kp_synchronize(c), c);
}

/* ---- dictionary, lift ---------------------------------------------------------------------------------------------------- */
static void c_basis_create(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis_desc desc;
  basis_desc(prhs[2], &desc);
  kp_basis* b = NULL;
  check(kp_basis_create(c, &desc, &b), c);
  plhs[0] = put_handle(b, H_BASIS);
}
static void dims4(mxArray* plhs[], const int v[4]) {
  plhs[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
  for (int i = 0; i < 4; ++i) mxGetPr(plhs[0])[i] = v[i];
}
static void c_basis_dims(ARGS) {
  UNUSED;
  int v[4];
  check(kp_basis_dims((kp_basis*)get_handle(prhs[1], H_BASIS), &v[0], &v[1], &v[2], &v[3]), NULL);
  dims4(plhs, v);
}
static void c_basis_desc_dims(ARGS) {
  UNUSED;
  kp_basis_desc desc;
  basis_desc(prhs[1], &desc);
  int v[4];
  check(kp_basis_desc_dims(&desc, &v[0], &v[1], &v[2], &v[3]), NULL);
  dims4(plhs, v);
}
static void c_basis_destroy(ARGS) { UNUSED; kp_basis_destroy((kp_basis*)take_handle(prhs[1], H_BASIS)); }
static void c_snapshots_destroy(ARGS) { UNUSED; kp_snapshots_destroy((kp_snapshots*)take_handle(prhs[1], H_SNAPS)); }
static void c_mpc_destroy(ARGS) { UNUSED; kp_mpc_destroy((kp_mpc*)take_handle(prhs[1], H_MPC)); }
static void c_sym_eig(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int n = (int)mxGetM(prhs[2]);
  if ((int)mxGetN(prhs[2]) != n) mexErrMsgIdAndTxt("kp:size", "sym_eig: square matrix expected");
  plhs[0] = mxCreateDoubleMatrix(n, n, mxREAL);
  mxArray* lam = mxCreateDoubleMatrix(n, 1, mxREAL);
  int sweeps = 0;
  check(kp_sym_eig(c, dbl(prhs[2]), n, mxGetPr(plhs[0]), mxGetPr(lam), &sweeps), c);
  set_or_drop(nlhs, plhs, 1, lam);
  set_or_drop(nlhs, plhs, 2, mxCreateDoubleScalar(sweeps));
}
static void c_lift(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  const int what = int_arg(prhs[3], "what");
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  if (what < 0 || what > 2) mexErrMsgIdAndTxt("kp:usage", "lift: what must be 0 (full), 1 (econ) or 2 (row)");
  const mwSize rows = mxGetM(prhs[4]);
  const int width = what == KP_LIFT_FULL ? nf : what == KP_LIFT_ECON ? N : W;
  const mxArray* u = nrhs > 5 ? prhs[5] : NULL;
  if (u && !mxIsEmpty(u) && mxGetM(u) != rows) mexErrMsgIdAndTxt("kp:size", "lift: zeta and u differ in rows");
  plhs[0] = mxCreateDoubleMatrix(rows, (mwSize)width, mxREAL);
  check(kp_lift(c, b, what, dbl(prhs[4]), u ? dbl(u) : NULL, (int64_t)rows, mxGetPr(plhs[0])), c);
}

/* ---- snapshots -------------------------------------------------------------------------------------------------------------- */
static void pairs_check(const mxArray* a, const mxArray* b, const mxArray* u) {
  if (mxGetM(a) != mxGetM(b) || mxGetN(a) != mxGetN(b)) mexErrMsgIdAndTxt("kp:size", "alpha and beta differ in size");
  if (!mxIsEmpty(u) && mxGetM(u) != mxGetM(a)) mexErrMsgIdAndTxt("kp:size", "alpha and u differ in rows");
}
static void c_snapshots_upload(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_snapshots* s = NULL;
  pairs_check(prhs[2], prhs[3], prhs[4]);
  check(kp_snapshots_upload(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), (int64_t)mxGetM(prhs[2]), (int)mxGetN(prhs[2]),
                            (int)mxGetN(prhs[4]), &s), c);
  plhs[0] = put_handle(s, H_SNAPS);
}
static void c_snapshots_resident(ARGS) {
  UNUSED;
  /* the context's resident object, refilled in place (kp_snapshots_update: no hipMalloc / hipFree per call, staged chunked
   * transfer); owned by the MEX file - do not destroy */
  kp_ctx* c = CTX(1);
  pairs_check(prhs[2], prhs[3], prhs[4]);
  const int nz = (int)mxGetN(prhs[2]), m_ = (int)mxGetN(prhs[4]);
  const int slot = slot_of(c);
  if (slot < 0) mexErrMsgIdAndTxt("kp:handle", "snapshots_resident: unknown context");
  if (g_snaps[slot] && (g_snaps_nz[slot] != nz || g_snaps_m[slot] != m_)) {
    reg_del(g_snaps[slot]);
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
  plhs[0] = put_handle(g_snaps[slot], H_SNAPS);
}
static void c_snapshots_update(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  pairs_check(prhs[3], prhs[4], prhs[5]);
  check(kp_snapshots_update(c, (kp_snapshots*)get_handle(prhs[2], H_SNAPS), dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), (int64_t)mxGetM(prhs[3])), c);
}

/* ---- fit ------------------------------------------------------------------------------------------------------------------------ */
static void rank_warning(const kp_ctx* c, int W) {               /* like mldivide: "Warning: Rank deficient, rank = ..." */
  int r = W;
  kp_fit_last_rank(c, &r);
  if (r >= 0 && r < W) mexWarnMsgIdAndTxt("kp:rankDeficient", "Rank deficient, rank = %d of %d: basic solution returned", r, W);
}
static void fit_common(ARGS, int sharded) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  const int nl = (int)mxGetNumberOfElements(prhs[4]);
  if (nl < 1) mexErrMsgIdAndTxt("kp:usage", "fit: at least one lasso value");
  plhs[0] = dstack(W, W, nl);
  kp_snapshots* s = (kp_snapshots*)get_handle(prhs[3], H_SNAPS);
  check(sharded ? kp_fit_sharded(c, b, s, dbl(prhs[4]), nl, mxGetPr(plhs[0])) : kp_fit(c, b, s, dbl(prhs[4]), nl, mxGetPr(plhs[0])), c);
  rank_warning(c, W);
}
static void c_fit(ARGS) { fit_common(nlhs, plhs, nrhs, prhs, 0); }
static void c_fit_sharded(ARGS) { fit_common(nlhs, plhs, nrhs, prhs, 1); }
static void c_fit_async(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  check(kp_fit(c, (kp_basis*)get_handle(prhs[2], H_BASIS), (kp_snapshots*)get_handle(prhs[3], H_SNAPS), NULL, 1, NULL), c);
}
static void c_fit_get_K(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int W = int_arg(prhs[3], "W");
  if (W < 1) mexErrMsgIdAndTxt("kp:usage", "fit_get_K: W");
  plhs[0] = mxCreateDoubleMatrix(W, W, mxREAL);
  check(kp_fit_get_K(c, int_arg(prhs[2], "index"), W, mxGetPr(plhs[0])), c);
}
static void c_fit_async_slots(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  check(kp_fit_async_slots(c, int_arg(prhs[2], "n_slots")), c);
}
static void gram_common(ARGS, int sharded) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  plhs[0] = mxCreateDoubleMatrix(W, W, mxREAL);
  mxArray* Cm = mxCreateDoubleMatrix(W, W, mxREAL);
  kp_snapshots* s = (kp_snapshots*)get_handle(prhs[3], H_SNAPS);
  check(sharded ? kp_fit_gram_sharded(c, b, s, mxGetPr(plhs[0]), mxGetPr(Cm)) : kp_fit_gram(c, b, s, mxGetPr(plhs[0]), mxGetPr(Cm)), c);
  set_or_drop(nlhs, plhs, 1, Cm);
}
static void c_fit_gram(ARGS) { gram_common(nlhs, plhs, nrhs, prhs, 0); }
static void c_fit_gram_sharded(ARGS) { gram_common(nlhs, plhs, nrhs, prhs, 1); }
static void gc_check(const mxArray* G, const mxArray* Cm) {
  if (mxGetM(G) != mxGetN(G) || mxGetM(Cm) != mxGetM(G) || mxIsEmpty(G)) mexErrMsgIdAndTxt("kp:size", "G must be square and C have as many rows");
}
static void c_fit_solve(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  gc_check(prhs[2], prhs[3]);
  const int W = (int)mxGetM(prhs[2]), nc = (int)mxGetN(prhs[3]);
  plhs[0] = mxCreateDoubleMatrix(W, nc, mxREAL);
  check(kp_fit_solve(c, dbl(prhs[2]), dbl(prhs[3]), W, nc, mxGetPr(plhs[0])), c);
  rank_warning(c, W);
}
static void c_fit_lasso(ARGS) {
  UNUSED;
  /* min 1/2 ||Px K - Py||^2 s.t. ||vec K||_1 <= t from the Grams (solve_KoopmanQP, Ksysid.m:1095-1176; the caller forms
   * t = lasso * N, :996) */
  kp_ctx* c = CTX(1);
  gc_check(prhs[2], prhs[3]);
  const int W = (int)mxGetM(prhs[2]), nc = (int)mxGetN(prhs[3]);
  int iters = 0;
  plhs[0] = mxCreateDoubleMatrix(W, nc, mxREAL);
  check(kp_fit_lasso(c, dbl(prhs[2]), dbl(prhs[3]), W, nc, mxGetScalar(prhs[4]), 20000, 1e-10, mxGetPr(plhs[0]), &iters), c);
  set_or_drop(nlhs, plhs, 1, mxCreateDoubleScalar(iters));
}
static void c_fit_lasso_batch(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  gc_check(prhs[2], prhs[3]);
  const int W = (int)mxGetM(prhs[2]), nc = (int)mxGetN(prhs[3]), nv = (int)mxGetNumberOfElements(prhs[4]);
  if (nv < 1) mexErrMsgIdAndTxt("kp:usage", "fit_lasso_batch: at least one budget");
  int* it = (int*)calloc((size_t)nv, sizeof(int));
  plhs[0] = dstack(W, nc, nv);
  const int rc = kp_fit_lasso_batch(c, dbl(prhs[2]), dbl(prhs[3]), W, nc, dbl(prhs[4]), nv, 20000, 1e-10, mxGetPr(plhs[0]), it);
  mxArray* ito = ints_out(it, (size_t)nv, (size_t)nv, 1);
  free(it);
  check(rc, c);
  set_or_drop(nlhs, plhs, 1, ito);
}
static void c_fit_refine(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  const double* K0 = dbl_n(prhs[4], (size_t)W * W, "fit_refine: K");
  plhs[0] = mxCreateDoubleMatrix(W, W, mxREAL);
  memcpy(mxGetPr(plhs[0]), K0, (size_t)W * W * sizeof(double));
  check(kp_fit_refine(c, b, (kp_snapshots*)get_handle(prhs[3], H_SNAPS), int_arg(prhs[5], "steps"), mxGetPr(plhs[0])), c);
}
static void c_last_rank(ARGS) {
  UNUSED;
  int r = -1;
  check(kp_fit_last_rank(CTX(1), &r), NULL);
  plhs[0] = mxCreateDoubleScalar(r);
}
static void c_last_pivot_ratio(ARGS) {
  UNUSED;
  double q = 0;
  check(kp_fit_last_pivot_ratio(CTX(1), &q), NULL);
  plhs[0] = mxCreateDoubleScalar(q);
}
static void c_fit_batch(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  const int nb = int_arg(prhs[4], "nb");
  if (nb < 1) mexErrMsgIdAndTxt("kp:usage", "fit_batch: nb");
  int* st = (int*)calloc((size_t)nb, sizeof(int));
  plhs[0] = dstack(W, W, nb);
  mxArray* G = dstack(W, W, nb);
  mxArray* Cm = dstack(W, W, nb);
  const int rc = kp_fit_batch(c, b, (kp_snapshots*)get_handle(prhs[3], H_SNAPS), nb, (int64_t)mxGetScalar(prhs[5]), mxGetPr(plhs[0]), mxGetPr(G), mxGetPr(Cm), st);
  mxArray* so = ints_out(st, (size_t)nb, (size_t)nb, 1);
  free(st);
  check(rc, c);
  set_or_drop(nlhs, plhs, 1, G);
  set_or_drop(nlhs, plhs, 2, Cm);
  set_or_drop(nlhs, plhs, 3, so);
}

/* ---- models ------------------------------------------------------------------------------------------------------------------------ */
static void c_model_project(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int N = int_arg(prhs[5], "N"), m = int_arg(prhs[6], "m");
  if (N < 1 || m < 0) mexErrMsgIdAndTxt("kp:usage", "model_project: N, m");
  const size_t W2 = (size_t)(N + m) * (N + m);
  const double *K = dbl_n(prhs[2], W2, "model_project: K"), *G = dbl_n(prhs[3], W2, "model_project: G"), *Cm = dbl_n(prhs[4], W2, "model_project: C");
  plhs[0] = mxCreateDoubleMatrix(N, N, mxREAL);
  mxArray* B = mxCreateDoubleMatrix(N, m, mxREAL);
  mxArray* M = mxCreateDoubleMatrix(N, N, mxREAL);
  check(kp_model_project(c, K, G, Cm, N, m, mxGetPr(plhs[0]), mxGetPr(B), mxGetPr(M)), c);
  set_or_drop(nlhs, plhs, 1, B);
  set_or_drop(nlhs, plhs, 2, M);
}
static void c_model_project_batch(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int N = int_arg(prhs[5], "N"), m = int_arg(prhs[6], "m");
  const int nb = (int)pages(prhs[2]);
  if (N < 1 || m < 0 || nb < 1) mexErrMsgIdAndTxt("kp:usage", "model_project_batch: N, m, nb");
  const size_t W2 = (size_t)(N + m) * (N + m) * nb;
  const double *K = dbl_n(prhs[2], W2, "model_project_batch: K"), *G = dbl_n(prhs[3], W2, "model_project_batch: G"), *Cm = dbl_n(prhs[4], W2, "model_project_batch: C");
  int* st = (int*)calloc((size_t)nb, sizeof(int));
  plhs[0] = dstack(N, N, nb);
  mxArray* B = dstack(N, m, nb);
  mxArray* M = dstack(N, N, nb);
  const int rc = kp_model_project_batch(c, K, G, Cm, nb, N, m, mxGetPr(plhs[0]), mxGetPr(B), mxGetPr(M), st);
  mxArray* so = ints_out(st, (size_t)nb, (size_t)nb, 1);
  free(st);
  check(rc, c);
  set_or_drop(nlhs, plhs, 1, B);
  set_or_drop(nlhs, plhs, 2, M);
  set_or_drop(nlhs, plhs, 3, so);
}
static void c_rollout(ARGS) {
  UNUSED;
  /* Y = kp_mex('rollout', h, model_type, A, B, z0, U, n_out): z0 N x batch, A N x N x batch, B N x mb x batch, U T x m x batch */
  kp_ctx* c = CTX(1);
  const int mt = int_arg(prhs[2], "model_type");
  const int N = (int)mxGetM(prhs[3]), batch = (int)mxGetN(prhs[5]);
  const int T = (int)mxGetM(prhs[6]), n_out = int_arg(prhs[7], "n_out");
  if (N < 1 || batch < 1 || T < 1) mexErrMsgIdAndTxt("kp:size", "rollout: empty argument");
  const int m = (int)(mxGetN(prhs[6]) / (size_t)batch);
  const size_t mb = mt == KP_MODEL_BILINEAR ? (size_t)N * m : (size_t)m;
  dbl_n(prhs[3], (size_t)N * N * batch, "rollout: A");
  dbl_n(prhs[4], (size_t)N * mb * batch, "rollout: B");
  dbl_n(prhs[5], (size_t)N * batch, "rollout: z0");
  dbl_n(prhs[6], (size_t)T * m * batch, "rollout: U");
  plhs[0] = dstack(T, n_out, batch);
  check(kp_rollout(c, mt, batch, dbl(prhs[3]), dbl(prhs[4]), N, m, dbl(prhs[5]), dbl(prhs[6]), T, n_out, mxGetPr(plhs[0])), c);
}
static void c_rollout_nl(ARGS) {
  UNUSED;
  /* Z = kp_mex('rollout_nl', h, b, Kf, zeta0, U): Kf nzeta x N x batch, zeta0 nzeta x batch, U T x m x batch -> Z T x nzeta x batch */
  kp_ctx* c = CTX(1);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nv, nf, N, W;
  basis_wn(b, c, &nv, &nf, &N, &W);
  const int nz = (int)mxGetM(prhs[3]), batch = (int)mxGetN(prhs[4]), T = (int)mxGetM(prhs[5]);
  if (nz < 1 || batch < 1 || T < 1) mexErrMsgIdAndTxt("kp:size", "rollout_nl: empty argument");
  dbl_n(prhs[3], (size_t)nz * N * batch, "rollout_nl: Kf");
  dbl_n(prhs[4], (size_t)nz * batch, "rollout_nl: zeta0");
  if (mxGetNumberOfElements(prhs[5]) % ((size_t)T * batch)) mexErrMsgIdAndTxt("kp:size", "rollout_nl: U must be T x m x batch");
  plhs[0] = dstack(T, nz, batch);
  check(kp_rollout_nl(c, b, batch, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), T, mxGetPr(plhs[0])), c);
}

/* ---- random-system sweep -------------------------------------------------------------------------------------------------------------- */
static void traj_shapes(const mxArray* const* a, int ntrials, int* nb, int* T, int* n, int* m, int* Tv) {
  /* a[0..3] = Y rows x n x nb, U rows x m x nb, Yv Tv x n x nb, Uv Tv x m x nb */
  const mwSize* dY = mxGetDimensions(a[0]);
  const mwSize* dU = mxGetDimensions(a[1]);
  const mwSize* dYv = mxGetDimensions(a[2]);
  const mwSize* dUv = mxGetDimensions(a[3]);
  const size_t pb = pages(a[0]);
  if (ntrials < 1 || dY[0] % (size_t)ntrials) mexErrMsgIdAndTxt("kp:size", "traj: rows of Y must be ntrials * T");
  if (dU[0] != dY[0] || pages(a[1]) != pb || pages(a[2]) != pb || pages(a[3]) != pb || dYv[1] != dY[1] || dUv[1] != dU[1] || dUv[0] != dYv[0])
    mexErrMsgIdAndTxt("kp:size", "traj: Y rows x n x nb, U rows x m x nb, Yv Tv x n x nb, Uv Tv x m x nb");
  *nb = (int)pb; *T = (int)(dY[0] / (size_t)ntrials); *n = (int)dY[1]; *m = (int)dU[1]; *Tv = (int)dYv[0];
}
static void c_traj_upload(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int ntrials = int_arg(prhs[6], "ntrials");
  int nb, T, n, m, Tv;
  traj_shapes(prhs + 2, ntrials, &nb, &T, &n, &m, &Tv);
  kp_traj* t = NULL;
  check(kp_traj_upload(c, dbl(prhs[2]), dbl(prhs[3]), nb, ntrials, T, n, m, dbl(prhs[4]), dbl(prhs[5]), Tv, &t), c);
  plhs[0] = put_handle(t, H_TRAJ);
}
static void c_traj_create(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_traj* t = NULL;
  check(kp_traj_create(c, int_arg(prhs[2], "nb"), int_arg(prhs[3], "ntrials"), int_arg(prhs[4], "T"), int_arg(prhs[5], "n"), int_arg(prhs[6], "m"),
                       int_arg(prhs[7], "Tv"), &t), c);
  plhs[0] = put_handle(t, H_TRAJ);
}
static void traj_dims(const kp_traj* t, int d[6]) { check(kp_traj_dims(t, &d[0], &d[1], &d[2], &d[3], &d[4], &d[5]), NULL); }
static void c_traj_put(ARGS) {
  UNUSED;
  kp_traj* t = (kp_traj*)get_handle(prhs[1], H_TRAJ);
  const int which = int_arg(prhs[2], "which");
  int d[6];
  traj_dims(t, d);                                    /* nb ntrials T n m Tv */
  if (which < 0 || which > 3) mexErrMsgIdAndTxt("kp:usage", "traj_put: which = 0 (Y), 1 (U), 2 (Yv), 3 (Uv)");
  const size_t rows = which < 2 ? (size_t)d[1] * d[2] : (size_t)d[5], width = (which & 1) ? (size_t)d[4] : (size_t)d[3];
  const double* blk = dbl_n(prhs[3], rows * width * d[0], "traj_put: block");
  /* the mxArray is pageable memory: the library's copy has returned from it when this call returns */
  check(kp_traj_put(t, which, blk), NULL);
}
static void c_traj_finish(ARGS) { UNUSED; check(kp_traj_finish((kp_traj*)get_handle(prhs[1], H_TRAJ)), NULL); }
static void c_traj_destroy(ARGS) { UNUSED; kp_traj_destroy((kp_traj*)take_handle(prhs[1], H_TRAJ)); }
static void c_traj_dims(ARGS) {
  UNUSED;
  int d[6];
  traj_dims((kp_traj*)get_handle(prhs[1], H_TRAJ), d);
  plhs[0] = ints_out(d, 6, 1, 6);
}
static void c_traj_scale(ARGS) {
  UNUSED;
  kp_traj* t = (kp_traj*)get_handle(prhs[1], H_TRAJ);
  int d[6];
  traj_dims(t, d);
  plhs[0] = mxCreateDoubleMatrix((mwSize)(2 * (d[3] + d[4])), (mwSize)d[0], mxREAL);
  check(kp_traj_scale(t, mxGetPr(plhs[0])), NULL);
}
static void c_sweep_eval(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_traj* t = (kp_traj*)get_handle(prhs[2], H_TRAJ);
  kp_basis* b = (kp_basis*)get_handle(prhs[3], H_BASIS);
  int d[6], nv, nf, N, W;
  traj_dims(t, d);
  basis_wn(b, c, &nv, &nf, &N, &W);
  const int nb = d[0], n = d[3];
  int* st = (int*)calloc((size_t)nb, sizeof(int));
  plhs[0] = mxCreateDoubleMatrix(n, nb, mxREAL);
  mxArray* K = nlhs > 1 ? dstack(W, W, nb) : NULL;
  const int rc = kp_sweep_eval(c, t, b, mxGetScalar(prhs[4]), mxGetPr(plhs[0]), K ? mxGetPr(K) : NULL, st);
  mxArray* so = ints_out(st, (size_t)nb, (size_t)nb, 1);
  free(st);
  check(rc, c);
  if (K) plhs[1] = K;
  set_or_drop(nlhs, plhs, 2, so);
}
static void c_sweep_eval_nested(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  kp_traj* t = (kp_traj*)get_handle(prhs[2], H_TRAJ);
  int d[6];
  traj_dims(t, d);
  const int nb = d[0], n = d[3], nd = int_arg(prhs[5], "n_deg");
  if (nd < 1) mexErrMsgIdAndTxt("kp:usage", "sweep_eval_nested: n_deg");
  int* st = (int*)calloc((size_t)nb * nd, sizeof(int));
  plhs[0] = dstack(n, nb, nd);
  const int rc = kp_sweep_eval_nested(c, t, (kp_basis*)get_handle(prhs[3], H_BASIS), mxGetScalar(prhs[4]), nd, mxGetPr(plhs[0]), st);
  mxArray* so = ints_out(st, (size_t)nb * nd, (size_t)nb, (size_t)nd);
  free(st);
  check(rc, c);
  set_or_drop(nlhs, plhs, 1, so);
}
static void c_sweep_nested_get_K(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int nb = int_arg(prhs[2], "nb"), W = int_arg(prhs[6], "W");
  if (nb < 1 || W < 1) mexErrMsgIdAndTxt("kp:usage", "sweep_nested_get_K: nb, W");
  plhs[0] = dstack(W, W, nb);
  check(kp_sweep_nested_get_K(c, nb, int_arg(prhs[3], "Wmax"), int_arg(prhs[4], "n_deg"), int_arg(prhs[5], "deg_index"), W, mxGetPr(plhs[0])), c);
}

/* ---- MPC ------------------------------------------------------------------------------------------------------------------------------------ */
typedef struct { int mt, N, m, Np, nproj; const double *A, *B, *proj, *r, *lo, *hi; double q_run, q_term, slope, smooth; } mpc_args;
static void mpc_parse(const mxArray* const* a, mpc_args* p) {       /* a[0] = model_type ... a[11] = smooth */
  p->mt = int_arg(a[0], "model_type");
  p->N = (int)mxGetM(a[1]);
  p->Np = int_arg(a[3], "Np");
  p->nproj = (int)mxGetM(a[4]);
  p->m = (int)mxGetNumberOfElements(a[7]);
  if (p->N < 1 || p->m < 1 || p->Np < 1 || p->nproj < 1) mexErrMsgIdAndTxt("kp:size", "mpc_create: empty model / horizon / projection / r");
  const size_t mb = p->mt == KP_MODEL_BILINEAR ? (size_t)p->N * p->m : (size_t)p->m;
  p->A = dbl_n(a[1], (size_t)p->N * p->N, "mpc_create: A");
  p->B = dbl_n(a[2], (size_t)p->N * mb, "mpc_create: B");
  p->proj = dbl_n(a[4], (size_t)p->nproj * p->N, "mpc_create: proj");
  p->q_run = mxGetScalar(a[5]);
  p->q_term = mxGetScalar(a[6]);
  p->r = dbl(a[7]);
  p->lo = mxIsEmpty(a[8]) ? NULL : dbl_n(a[8], (size_t)p->m, "mpc_create: lo");
  p->hi = mxIsEmpty(a[9]) ? NULL : dbl_n(a[9], (size_t)p->m, "mpc_create: hi");
  p->slope = mxIsEmpty(a[10]) ? NAN : mxGetScalar(a[10]);
  p->smooth = mxIsEmpty(a[11]) ? NAN : mxGetScalar(a[11]);
}
static void c_mpc_create(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  mpc_args p;
  mpc_parse(prhs + 2, &p);
  kp_mpc* mp = NULL;
  check(kp_mpc_create(c, p.mt, p.A, p.B, p.N, p.m, p.Np, p.proj, p.nproj, p.q_run, p.q_term, p.r, p.lo, p.hi, p.slope, p.smooth, &mp), c);
  plhs[0] = put_handle(mp, H_MPC);
}
static void c_mpc_set_state_bounds(ARGS) {
  UNUSED;
  const int n = (int)mxGetNumberOfElements(prhs[2]);
  if ((int)mxGetNumberOfElements(prhs[3]) != n) mexErrMsgIdAndTxt("kp:size", "mpc_set_state_bounds: lo and hi differ in length");
  check(kp_mpc_set_state_bounds((kp_mpc*)get_handle(prhs[1], H_MPC), n, dbl(prhs[2]), dbl(prhs[3])), NULL);
}
static void c_mpc_dims(ARGS) {
  UNUSED;
  int d[2];
  check(kp_mpc_dims((kp_mpc*)get_handle(prhs[1], H_MPC), &d[0], &d[1]), NULL);
  plhs[0] = ints_out(d, 2, 1, 2);
}
static void c_mpc_step_zeta(ARGS) {
  UNUSED;
  kp_mpc* mp = (kp_mpc*)get_handle(prhs[1], H_MPC);
  kp_basis* b = (kp_basis*)get_handle(prhs[2], H_BASIS);
  int nvar, nrows, nv, nf, N, W, status = 0;
  check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
  basis_wn(b, NULL, &nv, &nf, &N, &W);
  const int m = (int)mxGetNumberOfElements(prhs[4]);
  if (m < 1 || nvar % m) mexErrMsgIdAndTxt("kp:size", "mpc_step_zeta: u_prev must have m entries");
  const int Np = nvar / m;
  if ((int)mxGetNumberOfElements(prhs[3]) != nv) mexErrMsgIdAndTxt("kp:size", "mpc_step_zeta: zeta must have %d entries", nv);
  if (mxGetNumberOfElements(prhs[5]) % (size_t)(Np + 1)) mexErrMsgIdAndTxt("kp:size", "mpc_step_zeta: Yr must have nproj (Np + 1) entries");
  plhs[0] = mxCreateDoubleMatrix(Np, m, mxREAL);
  mxArray* z = mxCreateDoubleMatrix(N, 1, mxREAL);
  /* QP failure: U comes back NaN and the call itself succeeds - Ksim.m:220-222 tests any(isnan(U)) */
  check(kp_mpc_step_zeta(mp, b, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), nrhs > 6 ? int_arg(prhs[6], "iters") : 1, mxGetPr(plhs[0]),
                         mxGetPr(z), &status), NULL);
  set_or_drop(nlhs, plhs, 1, z);
}
static void c_mpc_step(ARGS) {
  UNUSED;
  /* the step from an already lifted state (loaded models lift with the current load estimate on the host side of the
   * boundary, Kmpc.m:347-348) */
  kp_mpc* mp = (kp_mpc*)get_handle(prhs[1], H_MPC);
  int nvar, nrows, status = 0;
  check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
  const int m = (int)mxGetNumberOfElements(prhs[3]);
  if (m < 1 || nvar % m) mexErrMsgIdAndTxt("kp:size", "mpc_step: u_prev must have m entries");
  const int Np = nvar / m;
  plhs[0] = mxCreateDoubleMatrix(Np, m, mxREAL);
  check(kp_mpc_step(mp, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), nrhs > 5 ? int_arg(prhs[5], "iters") : 1, mxGetPr(plhs[0]), &status), NULL);
  set_or_drop(nlhs, plhs, 1, mxCreateDoubleScalar(status));
}
static void batch_shapes(const mxArray* const* a, int nvar, int* nb, int* m) {      /* a = Z, Uprev, YR: columns = problems */
  *nb = (int)mxGetN(a[0]);
  *m = (int)mxGetM(a[1]);
  if (*nb < 1 || *m < 1 || nvar % *m || (int)mxGetN(a[1]) != *nb || (int)mxGetN(a[2]) != *nb)
    mexErrMsgIdAndTxt("kp:size", "mpc_step_batch: Z N x nb, Uprev m x nb, YR nproj (Np + 1) x nb");
}
static void c_mpc_step_batch(ARGS) {
  UNUSED;
  kp_mpc* mp = (kp_mpc*)get_handle(prhs[1], H_MPC);
  int nvar, nrows, nb, m;
  check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
  batch_shapes(prhs + 2, nvar, &nb, &m);
  int* st = (int*)calloc((size_t)nb, sizeof(int));
  plhs[0] = dstack(nvar / m, m, nb);
  const int rc = kp_mpc_step_batch(mp, nb, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), mxGetPr(plhs[0]), st);
  mxArray* so = ints_out(st, (size_t)nb, (size_t)nb, 1);
  free(st);
  check(rc, NULL);
  set_or_drop(nlhs, plhs, 1, so);
}
static void c_mpc_last_qp(ARGS) {
  UNUSED;
  kp_mpc* mp = (kp_mpc*)get_handle(prhs[1], H_MPC);
  int nvar, nrows;
  check(kp_mpc_dims(mp, &nvar, &nrows), NULL);
  plhs[0] = mxCreateDoubleMatrix(nvar, nvar, mxREAL);
  mxArray* f = mxCreateDoubleMatrix(nvar, 1, mxREAL);
  mxArray* Aq = mxCreateDoubleMatrix(nrows, nvar, mxREAL);
  mxArray* bq = mxCreateDoubleMatrix(nrows, 1, mxREAL);
  check(kp_mpc_last_qp(mp, mxGetPr(plhs[0]), mxGetPr(f), mxGetPr(Aq), mxGetPr(bq)), NULL);
  set_or_drop(nlhs, plhs, 1, f);
  set_or_drop(nlhs, plhs, 2, Aq);
  set_or_drop(nlhs, plhs, 3, bq);
}
static void c_mpc_last_profile(ARGS) {
  UNUSED;
  int counts[2] = {0, 0};
  plhs[0] = mxCreateDoubleMatrix(1, 6, mxREAL);
  check(kp_mpc_last_profile((kp_mpc*)get_handle(prhs[1], H_MPC), mxGetPr(plhs[0]), counts), NULL);
  set_or_drop(nlhs, plhs, 1, ints_out(counts, 2, 1, 2));
}
static void c_mpc_last_stamps(ARGS) {
  UNUSED;
  plhs[0] = mxCreateDoubleMatrix(1, 16, mxREAL);
  check(kp_mpc_last_stamps((kp_mpc*)get_handle(prhs[1], H_MPC), mxGetPr(plhs[0])), NULL);
}
static void c_qp_solve(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const int n = (int)mxGetM(prhs[2]), mr = (int)mxGetM(prhs[4]);
  if (n < 1 || (int)mxGetN(prhs[2]) != n) mexErrMsgIdAndTxt("kp:size", "qp_solve: H must be square");
  dbl_n(prhs[3], (size_t)n, "qp_solve: f");
  dbl_n(prhs[4], (size_t)mr * n, "qp_solve: A");
  dbl_n(prhs[5], (size_t)mr, "qp_solve: b");
  int status = 0;
  plhs[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
  check(kp_qp_solve(c, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), n, mr, mxGetPr(plhs[0]), &status), c);
  set_or_drop(nlhs, plhs, 1, mxCreateDoubleScalar(status));
}

/* ---- one process per GPU: RCCL ------------------------------------------------------------------------------------------------------------------ */
static void c_comm_unique_id(ARGS) {
  UNUSED;
  plhs[0] = mxCreateNumericMatrix(1, 128, mxUINT8_CLASS, mxREAL);
  check(kp_comm_unique_id(mxGetData(plhs[0])), NULL);
}
static void c_comm_create(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  if (!mxIsUint8(prhs[2]) || mxGetNumberOfElements(prhs[2]) != 128) mexErrMsgIdAndTxt("kp:type", "comm_create: id must be uint8 1 x 128");
  check(kp_comm_create(c, mxGetData(prhs[2]), int_arg(prhs[3], "rank"), int_arg(prhs[4], "world")), c);
}
static void c_comm_destroy(ARGS) { UNUSED; kp_ctx* c = CTX(1); check(kp_comm_destroy(c), c); }
static void c_comm_abandon(ARGS) { UNUSED; kp_ctx* c = CTX(1); check(kp_comm_abandon(c), c); }
static void comm_rw(kp_ctx* c, int rw[2]) { check(kp_comm_info(c, &rw[0], &rw[1]), c); }
static void c_comm_info(ARGS) {
  UNUSED;
  int rw[2];
  comm_rw(CTX(1), rw);
  plhs[0] = ints_out(rw, 2, 1, 2);
}
static void c_comm_allgather(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  int rw[2];
  comm_rw(c, rw);
  const size_t n = mxGetNumberOfElements(prhs[2]);
  plhs[0] = mxCreateDoubleMatrix(n, (mwSize)(rw[1] > 0 ? rw[1] : 1), mxREAL);
  check(kp_comm_allgather(c, dbl(prhs[2]), (int64_t)(n * sizeof(double)), mxGetPr(plhs[0])), c);
}
static void c_comm_allreduce_sum(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  const size_t n = mxGetNumberOfElements(prhs[2]);
  plhs[0] = mxCreateDoubleMatrix(mxGetM(prhs[2]), mxGetN(prhs[2]), mxREAL);
  if (n) memcpy(mxGetPr(plhs[0]), dbl(prhs[2]), n * sizeof(double));
  check(kp_comm_allreduce_sum(c, mxGetPr(plhs[0]), (int64_t)n), c);
}
static void c_comm_allgather_fit(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  int rw[2];
  comm_rw(c, rw);
  const int W = int_arg(prhs[3], "W");
  if (W < 1) mexErrMsgIdAndTxt("kp:usage", "comm_allgather_fit: W");
  plhs[0] = dstack(W, W, (size_t)(rw[1] > 0 ? rw[1] : 1));
  check(kp_comm_allgather_fit(c, int_arg(prhs[2], "index"), W, mxGetPr(plhs[0])), c);
}
static void c_comm_allgather_fits(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  int rw[2];
  comm_rw(c, rw);
  const int first = int_arg(prhs[2], "first"), count = int_arg(prhs[3], "count"), W = int_arg(prhs[4], "W");
  if (count < 1 || W < 1) mexErrMsgIdAndTxt("kp:usage", "comm_allgather_fits: count, W");
  const mwSize dims[4] = {(mwSize)W, (mwSize)W, (mwSize)count, (mwSize)(rw[1] > 0 ? rw[1] : 1)};
  plhs[0] = mxCreateNumericArray(4, dims, mxDOUBLE_CLASS, mxREAL);
  check(kp_comm_allgather_fits(c, first, count, W, mxGetPr(plhs[0])), c);
}
static void c_comm_gather_fits(ARGS) {
  UNUSED;
  kp_ctx* c = CTX(1);
  int rw[2];
  comm_rw(c, rw);
  const int root = int_arg(prhs[2], "root"), first = int_arg(prhs[3], "first"), count = int_arg(prhs[4], "count"), W = int_arg(prhs[5], "W");
  if (count < 1 || W < 1) mexErrMsgIdAndTxt("kp:usage", "comm_gather_fits: count, W");
  if (rw[0] != root) {                            /* this rank only sends */
    plhs[0] = mxCreateDoubleMatrix(0, 0, mxREAL);
    check(kp_comm_gather_fits(c, root, first, count, W, NULL), c);
    return;
  }
  const mwSize dims[4] = {(mwSize)W, (mwSize)W, (mwSize)count, (mwSize)(rw[1] > 0 ? rw[1] : 1)};
  plhs[0] = mxCreateNumericArray(4, dims, mxDOUBLE_CLASS, mxREAL);
  check(kp_comm_gather_fits(c, root, first, count, W, mxGetPr(plhs[0])), c);
}

/* ---- one caller, several GPUs ---------------------------------------------------------------------------------------------------------------------- */
static kp_multi* multi_handle(const mxArray* a) {
  kp_multi* g = (kp_multi*)get_handle(a, H_MULTI);
  for (int i = 0; i < MAXMULTI; ++i)
    if (g_multi[i] == g) return g;
  mexErrMsgIdAndTxt("kp:handle", "unknown multi-GPU object");
  return NULL;
}
static void c_multi_create(ARGS) {
  UNUSED;
  const int n = (int)mxGetNumberOfElements(prhs[1]);
  if (n < 1 || n > 64) mexErrMsgIdAndTxt("kp:usage", "multi_create: 1 to 64 device ids");
  int ids[64], slot = -1;
  const double* v = dbl(prhs[1]);
  for (int i = 0; i < n; ++i) ids[i] = (int)v[i];
  for (int i = 0; i < MAXMULTI && slot < 0; ++i)
    if (!g_multi[i]) slot = i;
  if (slot < 0) mexErrMsgIdAndTxt("kp:usage", "multi_create: too many multi-GPU objects (multi_destroy the old ones)");
  check(kp_multi_create(ids, n, &g_multi[slot]), NULL);
  plhs[0] = put_handle(g_multi[slot], H_MULTI);
}
static void c_multi_destroy(ARGS) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  for (int i = 0; i < MAXMULTI; ++i)
    if (g_multi[i] == g) g_multi[i] = NULL;
  reg_del(g);
  kp_multi_destroy(g);
}
static void c_multi_size(ARGS) {
  UNUSED;
  int n = 0;
  check(kp_multi_size(multi_handle(prhs[1]), &n), NULL);
  plhs[0] = mxCreateDoubleScalar(n);
}
static void c_multi_timers(ARGS) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  int n = 0;
  check(kp_multi_size(g, &n), NULL);
  plhs[0] = mxCreateDoubleMatrix(4, (mwSize)n, mxREAL);
  check_multi(kp_multi_timers(g, mxGetPr(plhs[0])), g);
}
static void multi_fit_common(ARGS, int sharded) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  kp_basis_desc desc;
  basis_desc(prhs[2], &desc);
  int W = 0;
  check(kp_basis_desc_dims(&desc, NULL, NULL, NULL, &W), NULL);
  pairs_check(prhs[3], prhs[4], prhs[5]);
  if ((int)mxGetN(prhs[3]) != desc.nzeta || (int)mxGetN(prhs[5]) != desc.m) mexErrMsgIdAndTxt("kp:size", "multi_fit: alpha needs nzeta columns and u m columns");
  const int nl = (int)mxGetNumberOfElements(prhs[6]);
  if (nl < 1) mexErrMsgIdAndTxt("kp:usage", "multi_fit: at least one lasso value");
  plhs[0] = dstack(W, W, nl);
  const int64_t Ns = (int64_t)mxGetM(prhs[3]);
  check_multi(sharded ? kp_multi_fit_sharded(g, &desc, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), Ns, dbl(prhs[6]), nl, mxGetPr(plhs[0]))
                      : kp_multi_fit(g, &desc, dbl(prhs[3]), dbl(prhs[4]), dbl(prhs[5]), Ns, dbl(prhs[6]), nl, mxGetPr(plhs[0])), g);
}
static void c_multi_fit(ARGS) { multi_fit_common(nlhs, plhs, nrhs, prhs, 0); }
static void c_multi_fit_sharded(ARGS) { multi_fit_common(nlhs, plhs, nrhs, prhs, 1); }
/* kp_multi_traj carries no size query: the gateway keeps what it needs beside the handle */
typedef struct { kp_multi_traj* t; int nb, n; } mtraj_rec;
static void c_multi_traj_upload(ARGS) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  const int ntrials = int_arg(prhs[6], "ntrials");
  int nb, T, n, m, Tv;
  traj_shapes(prhs + 2, ntrials, &nb, &T, &n, &m, &Tv);
  mtraj_rec* r = (mtraj_rec*)calloc(1, sizeof *r);
  const int rc = kp_multi_traj_upload(g, dbl(prhs[2]), dbl(prhs[3]), nb, ntrials, T, n, m, dbl(prhs[4]), dbl(prhs[5]), Tv, &r->t);
  if (rc) { free(r); check_multi(rc, g); }
  r->nb = nb;
  r->n = n;
  plhs[0] = put_handle(r, H_MTRAJ);
}
static void c_multi_traj_destroy(ARGS) {
  UNUSED;
  mtraj_rec* r = (mtraj_rec*)take_handle(prhs[1], H_MTRAJ);
  kp_multi_traj_destroy(r->t);
  free(r);
}
static void c_multi_sweep_eval_nested(ARGS) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  mtraj_rec* r = (mtraj_rec*)get_handle(prhs[2], H_MTRAJ);
  kp_basis_desc desc;
  basis_desc(prhs[3], &desc);
  const int nd = int_arg(prhs[5], "n_deg");
  if (nd < 1) mexErrMsgIdAndTxt("kp:usage", "multi_sweep_eval_nested: n_deg");
  int* st = (int*)calloc((size_t)r->nb * nd, sizeof(int));
  plhs[0] = dstack((size_t)r->n, (size_t)r->nb, (size_t)nd);
  const int rc = kp_multi_sweep_eval_nested(g, r->t, &desc, mxGetScalar(prhs[4]), nd, mxGetPr(plhs[0]), st);
  mxArray* so = ints_out(st, (size_t)r->nb * nd, (size_t)r->nb, (size_t)nd);
  free(st);
  check_multi(rc, g);
  set_or_drop(nlhs, plhs, 1, so);
}
typedef struct { kp_multi_mpc* p; kp_multi* g; int nvar; } mmpc_rec;
static void c_multi_mpc_create(ARGS) {
  UNUSED;
  kp_multi* g = multi_handle(prhs[1]);
  mpc_args p;
  mpc_parse(prhs + 2, &p);
  mmpc_rec* r = (mmpc_rec*)calloc(1, sizeof *r);
  const int rc = kp_multi_mpc_create(g, p.mt, p.A, p.B, p.N, p.m, p.Np, p.proj, p.nproj, p.q_run, p.q_term, p.r, p.lo, p.hi, p.slope, p.smooth, &r->p);
  if (rc) { free(r); check_multi(rc, g); }
  r->g = g;
  r->nvar = p.m * p.Np;
  plhs[0] = put_handle(r, H_MMPC);
}
static void c_multi_mpc_set_state_bounds(ARGS) {
  UNUSED;
  mmpc_rec* r = (mmpc_rec*)get_handle(prhs[1], H_MMPC);
  const int n = (int)mxGetNumberOfElements(prhs[2]);
  if ((int)mxGetNumberOfElements(prhs[3]) != n) mexErrMsgIdAndTxt("kp:size", "multi_mpc_set_state_bounds: lo and hi differ in length");
  check_multi(kp_multi_mpc_set_state_bounds(r->p, n, dbl(prhs[2]), dbl(prhs[3])), r->g);
}
static void c_multi_mpc_destroy(ARGS) {
  UNUSED;
  mmpc_rec* r = (mmpc_rec*)take_handle(prhs[1], H_MMPC);
  kp_multi_mpc_destroy(r->p);
  free(r);
}
static void c_multi_mpc_step_batch(ARGS) {
  UNUSED;
  mmpc_rec* r = (mmpc_rec*)get_handle(prhs[1], H_MMPC);
  int nb, m;
  batch_shapes(prhs + 2, r->nvar, &nb, &m);
  int* st = (int*)calloc((size_t)nb, sizeof(int));
  plhs[0] = dstack((size_t)(r->nvar / m), (size_t)m, (size_t)nb);
  const int rc = kp_multi_mpc_step_batch(r->p, nb, dbl(prhs[2]), dbl(prhs[3]), dbl(prhs[4]), mxGetPr(plhs[0]), st);
  mxArray* so = ints_out(st, (size_t)nb, (size_t)nb, 1);
  free(st);
  check_multi(rc, r->g);
  set_or_drop(nlhs, plhs, 1, so);
}

/* ---- the command table: name, arguments after the command (min, max), outputs (max), handler ------------------------------- */
static void c_commands(ARGS);
typedef struct { const char* name; int nrhs_min, nrhs_max, nlhs_max; void (*fn)(ARGS); } kp_command;
static const kp_command g_commands[] = {
  {"create", 0, 1, 1, c_create}, {"destroy", 1, 1, 0, c_destroy}, {"device_count", 0, 0, 1, c_device_count},
  {"device_info", 1, 1, 3, c_device_info}, {"last_error", 0, 1, 1, c_last_error}, {"timer_get", 2, 2, 1, c_timer_get},
  {"synchronize", 1, 1, 0, c_synchronize}, {"commands", 0, 0, 2, c_commands},
  {"basis_create", 2, 2, 1, c_basis_create}, {"basis_dims", 1, 1, 1, c_basis_dims}, {"basis_desc_dims", 1, 1, 1, c_basis_desc_dims},
  {"basis_destroy", 1, 1, 0, c_basis_destroy}, {"sym_eig", 2, 2, 3, c_sym_eig}, {"lift", 4, 5, 1, c_lift},
  {"snapshots_upload", 4, 4, 1, c_snapshots_upload}, {"snapshots_update", 5, 5, 0, c_snapshots_update},
  {"snapshots_resident", 4, 4, 1, c_snapshots_resident}, {"snapshots_destroy", 1, 1, 0, c_snapshots_destroy},
  {"fit", 4, 4, 1, c_fit}, {"fit_async", 3, 3, 0, c_fit_async}, {"fit_get_K", 3, 3, 1, c_fit_get_K},
  {"fit_async_slots", 2, 2, 0, c_fit_async_slots}, {"fit_gram", 3, 3, 2, c_fit_gram}, {"fit_solve", 3, 3, 1, c_fit_solve},
  {"fit_lasso", 4, 4, 2, c_fit_lasso}, {"fit_lasso_batch", 4, 4, 2, c_fit_lasso_batch}, {"fit_refine", 5, 5, 1, c_fit_refine},
  {"last_rank", 1, 1, 1, c_last_rank}, {"last_pivot_ratio", 1, 1, 1, c_last_pivot_ratio}, {"fit_batch", 5, 5, 4, c_fit_batch},
  {"fit_sharded", 4, 4, 1, c_fit_sharded}, {"fit_gram_sharded", 3, 3, 2, c_fit_gram_sharded},
  {"model_project", 6, 6, 3, c_model_project}, {"model_project_batch", 6, 6, 4, c_model_project_batch},
  {"rollout", 7, 7, 1, c_rollout}, {"rollout_nl", 5, 5, 1, c_rollout_nl},
  {"traj_upload", 6, 6, 1, c_traj_upload}, {"traj_create", 7, 7, 1, c_traj_create}, {"traj_put", 3, 3, 0, c_traj_put},
  {"traj_finish", 1, 1, 0, c_traj_finish}, {"traj_destroy", 1, 1, 0, c_traj_destroy}, {"traj_dims", 1, 1, 1, c_traj_dims},
  {"traj_scale", 1, 1, 1, c_traj_scale}, {"sweep_eval", 4, 4, 3, c_sweep_eval}, {"sweep_eval_nested", 5, 5, 2, c_sweep_eval_nested},
  {"sweep_nested_get_K", 6, 6, 1, c_sweep_nested_get_K},
  {"mpc_create", 13, 13, 1, c_mpc_create}, {"mpc_set_state_bounds", 3, 3, 0, c_mpc_set_state_bounds}, {"mpc_destroy", 1, 1, 0, c_mpc_destroy},
  {"mpc_dims", 1, 1, 1, c_mpc_dims}, {"mpc_step_zeta", 5, 6, 2, c_mpc_step_zeta}, {"mpc_step", 4, 5, 2, c_mpc_step},
  {"mpc_step_batch", 4, 4, 2, c_mpc_step_batch}, {"mpc_last_qp", 1, 1, 4, c_mpc_last_qp}, {"mpc_last_profile", 1, 1, 2, c_mpc_last_profile},
  {"mpc_last_stamps", 1, 1, 1, c_mpc_last_stamps},
  {"qp_solve", 5, 5, 2, c_qp_solve},
  {"comm_unique_id", 0, 0, 1, c_comm_unique_id}, {"comm_create", 4, 4, 0, c_comm_create}, {"comm_destroy", 1, 1, 0, c_comm_destroy},
  {"comm_abandon", 1, 1, 0, c_comm_abandon}, {"comm_info", 1, 1, 1, c_comm_info}, {"comm_allgather", 2, 2, 1, c_comm_allgather},
  {"comm_allreduce_sum", 2, 2, 1, c_comm_allreduce_sum}, {"comm_allgather_fit", 3, 3, 1, c_comm_allgather_fit},
  {"comm_allgather_fits", 4, 4, 1, c_comm_allgather_fits}, {"comm_gather_fits", 5, 5, 1, c_comm_gather_fits},
  {"multi_create", 1, 1, 1, c_multi_create}, {"multi_destroy", 1, 1, 0, c_multi_destroy}, {"multi_size", 1, 1, 1, c_multi_size},
  {"multi_timers", 1, 1, 1, c_multi_timers}, {"multi_fit", 6, 6, 1, c_multi_fit}, {"multi_fit_sharded", 6, 6, 1, c_multi_fit_sharded},
  {"multi_traj_upload", 6, 6, 1, c_multi_traj_upload}, {"multi_traj_destroy", 1, 1, 0, c_multi_traj_destroy},
  {"multi_sweep_eval_nested", 5, 5, 2, c_multi_sweep_eval_nested},
  {"multi_mpc_create", 13, 13, 1, c_multi_mpc_create}, {"multi_mpc_set_state_bounds", 3, 3, 0, c_multi_mpc_set_state_bounds},
  {"multi_mpc_destroy", 1, 1, 0, c_multi_mpc_destroy}, {"multi_mpc_step_batch", 4, 4, 2, c_multi_mpc_step_batch},
};
#define NCOMMANDS ((int)(sizeof g_commands / sizeof g_commands[0]))

static void c_commands(ARGS) {
  UNUSED;
  plhs[0] = mxCreateDoubleMatrix(NCOMMANDS, 3, mxREAL);
  size_t len = 1;
  for (int i = 0; i < NCOMMANDS; ++i) len += strlen(g_commands[i].name) + 1;
  char* names = (char*)calloc(len, 1);
  for (int i = 0; i < NCOMMANDS; ++i) {
    double* t = mxGetPr(plhs[0]);
    t[i] = g_commands[i].nrhs_min;
    t[i + NCOMMANDS] = g_commands[i].nrhs_max;
    t[i + 2 * NCOMMANDS] = g_commands[i].nlhs_max;
    strcat(names, g_commands[i].name);
    if (i + 1 < NCOMMANDS) strcat(names, "\n");
  }
  mxArray* s = mxCreateString(names);
  free(names);
  set_or_drop(nlhs, plhs, 1, s);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  char cmd[64];
  if (nrhs < 1 || !mxIsChar(prhs[0]) || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("kp:usage", "kp_mex(command, ...)");
  if (!mexIsLocked()) { mexLock(); mexAtExit(at_exit); }
  for (int i = 0; i < NCOMMANDS; ++i) {
    const kp_command* k = &g_commands[i];
    if (strcmp(cmd, k->name)) continue;
    const int na = nrhs - 1;
    if (na < k->nrhs_min || na > k->nrhs_max)
      mexErrMsgIdAndTxt("kp:usage", "kp_mex('%s', ...): %d to %d arguments after the command, got %d", cmd, k->nrhs_min, k->nrhs_max, na);
    if (nlhs > (k->nlhs_max > 1 ? k->nlhs_max : 1) || (k->nlhs_max == 0 && nlhs > 0))
      mexErrMsgIdAndTxt("kp:usage", "kp_mex('%s', ...): at most %d outputs", cmd, k->nlhs_max);
    k->fn(nlhs, plhs, nrhs, prhs);
    return;
  }
  mexErrMsgIdAndTxt("kp:usage", "unknown command '%s'", cmd);
}

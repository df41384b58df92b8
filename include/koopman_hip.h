/* koopman_hip.h — C ABI of libkoopman_hip.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for ONE hot path of roahmlab/koopman-realizations: the EDMD fit of
 * Ksysid.m and the per-step MPC solve of Kmpc.m.  The reference has no FFI (it is pure
 * MATLAB); the seam is its class-method signatures.  Each entry point below names the
 * reference code it replaces (file:line under the reference checkout).  A MATLAB host
 * binds these with loadlibrary/calllib or a thin MEX gateway.
 *
 * Conventions
 *   - all matrices are IEEE f64, COLUMN-MAJOR (MATLAB layout), rows = snapshots/time;
 *   - every pointer is a HOST pointer unless the name ends in _dev or the type is an
 *     opaque handle; the library copies in/out and never keeps caller pointers;
 *   - every function returns KP_OK (0) or a negative kp_status; kp_last_error() gives text;
 *   - handles own device memory and are freed by the matching *_destroy; kp_basis, kp_snapshots, kp_traj and kp_mpc
 *     handles point into the kp_ctx they were created on and must be destroyed BEFORE it (kp_destroy does not track them);
 *   - calls on one kp_ctx are serialised by the caller (MATLAB is single threaded);
 *     the library synchronises its stream before returning host results.
 */
#ifndef KOOPMAN_HIP_H
#define KOOPMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  KP_OK = 0,
  KP_ERR_ARG = -1,        /* bad argument / unsupported configuration            */
  KP_ERR_HIP = -2,        /* HIP runtime failure (no device, OOM, launch error)   */
  KP_ERR_NOT_SPD = -3,    /* Gram matrix not numerically SPD (rank deficient Psi) */
  KP_ERR_QP_FAIL = -4,    /* QP infeasible / iteration cap: caller maps to NaN U  */
  KP_ERR_NOT_CONVERGED = -5
} kp_status;

typedef struct kp_ctx kp_ctx;             /* device, stream, workspaces            */
typedef struct kp_basis kp_basis;         /* dictionary of observables on device   */
typedef struct kp_snapshots kp_snapshots; /* snapshot pairs resident in HBM        */
typedef struct kp_mpc kp_mpc;             /* condensed MPC problem on device       */

/* ---- context ------------------------------------------------------------------- */
int kp_device_count(int* count);          /* GPUs visible to this process (0 without a GPU); does not create a context */
int kp_create(int device_id, kp_ctx** ctx);
int kp_destroy(kp_ctx* ctx);
const char* kp_last_error(const kp_ctx* ctx);   /* ctx may be NULL: last global error */
/* name (<=255 chars), number of CUs, total HBM bytes of the bound device */
int kp_device_info(const kp_ctx* ctx, char* name, int name_len, int* num_cu, int64_t* hbm_bytes);
/* milliseconds spent in the device part of the most recent call of each kind
 * (HIP events on the library stream): which = 0 fused lift+Gram kernel, 1 solve, 2 mpc step, 3 lasso, 4 lift, 5 rollout, 6 Gram partial reduction.
 * After a run of pipelined fits (kp_fit with K_out == NULL) timer 0 is the MEAN duration of the last (up to 64) Gram
 * launches and which = 7 the number of launches in that mean.  which = 8: the first (widest) product G [K_1 ... K_nv] of the
 * most recent lasso batch, 9: its number of columns (kp_symm_gemm2_kernel: the FISTA iteration's product); 10: flop per snapshot pair the most recent
 * fused lift+Gram launch EXECUTES on the matrix pipe (padding and, for dim_red dictionaries, the projection included); 11: host
 * milliseconds the most recent lasso batch spent in the regularisation-path homotopy (0: the projected-gradient iteration
 * finished every value; kp_fit_lasso below). */
int kp_timer_get(const kp_ctx* ctx, int which, double* ms);
/* Device pointer + byte size of the library stream's raw handle, for profilers/benchmarks
 * that want to bracket work with their own HIP events: returns hipStream_t as void*. */
void* kp_stream(const kp_ctx* ctx);

/* ---- dictionary ------------------------------------------------------------------
 * Replaces the symbolic dictionary of Ksysid.def_observables (Ksysid.m:455-536),
 * def_polyLift (:629-677, monomial order from partitions.m:206-219), def_fourierLift
 * (:694-731), def_gaussianLift (:790-817) and the reduced basis of get_econ_observables
 * (:1435-1577, econ_full :1615, econ_full_input :1594).  matlabFunction closures cannot
 * cross a C ABI, so the dictionary is carried as data.
 *
 * fullBasis = [ v ; block_1 ; ... ; block_nb ; 1 ] over nvars variables v, where
 * nvars = nzeta for 'linear'/'bilinear' and nzeta+m for 'nonlinear' (u appended,
 * Ksysid.m:475-477).  Blocks, in obs_type order:
 *   KP_BLOCK_POLY      count = rows of exponents (nvars bytes each), i.e. the monomials
 *                      AFTER the first nvars (Ksysid.m:488); data: exponents in poly_exps
 *   KP_BLOCK_FOURIER   count = degree; (1+2deg)^nvars - 1 functions, last variable fastest
 *   KP_BLOCK_GAUSSIAN  count = number of centres; data: nvars doubles per centre
 *   KP_BLOCK_HERMITE   count = rows of orders (nvars bytes each, in poly_exps): products of
 *                      hermiteH(order_i, v_i), ALL rows of degree 1..deg (Ksysid.m:820-851)
 *   KP_BLOCK_FOURIER_SPARSER  count = rows of multipliers (2*nvars bytes each, in poly_exps):
 *                      prod sin(2 pi a_i v_i) * prod cos(2 pi b_i v_i), zero multipliers skipped
 *                      (def_fourierLift_sparser / get_sinusoid, Ksysid.m:734-786)
 * poly_exps carries the byte rows of the POLY, HERMITE and FOURIER_SPARSER blocks in block order.
 * pcs (Nfull x k, column-major) may be NULL (dim_red = false).
 */
enum { KP_MODEL_LINEAR = 0, KP_MODEL_BILINEAR = 1, KP_MODEL_NONLINEAR = 2 };
enum { KP_BLOCK_POLY = 0, KP_BLOCK_FOURIER = 1, KP_BLOCK_GAUSSIAN = 2, KP_BLOCK_HERMITE = 3, KP_BLOCK_FOURIER_SPARSER = 4 };

typedef struct {
  int32_t model_type;          /* KP_MODEL_*                                         */
  int32_t nzeta;               /* n*(nd+1)+m*nd, Ksysid.m:86                         */
  int32_t m;                   /* number of inputs                                   */
  int32_t n_blocks;
  const int32_t* block_type;   /* n_blocks                                           */
  const int32_t* block_count;  /* n_blocks                                           */
  const uint8_t* poly_exps;    /* byte rows of POLY/HERMITE/FOURIER_SPARSER blocks    */
  const double* gauss_centres; /* all GAUSSIAN blocks' centres, concatenated         */
  int32_t k_pcs;               /* 0 => no dimension reduction                        */
  const double* pcs;           /* Nfull x k_pcs column-major, or NULL                */
} kp_basis_desc;

int kp_basis_create(kp_ctx* ctx, const kp_basis_desc* desc, kp_basis** basis);
int kp_basis_destroy(kp_basis* basis);
/* Nfull = length(basis.full) (Ksysid.m:534); N = params.N after dim_red (:1512-1516);
 * W = width of Px (:1019-1028): N+m linear, N(m+1) bilinear, N nonlinear. */
int kp_basis_dims(const kp_basis* basis, int* nvars, int* nfull, int* N, int* W);
/* the same sizes from a descriptor alone (no device, no context): kp_multi_* callers size their outputs with it */
int kp_basis_desc_dims(const kp_basis_desc* desc, int* nvars, int* nfull, int* N, int* W);

/* Symmetric eigendecomposition S = V diag(lam) V' of an n x n matrix (n <= 1024, column-major, host pointers) on the
 * device (parallel cyclic Jacobi; several workgroups with a grid barrier per round for n > 40): the `pca` step of
 * get_econ_observables (Ksysid.m:1498) - principal
 * axes = eigenvectors of the covariance of the lifted snapshots, which kp_fit_gram provides without forming the lifted
 * matrix (the dictionary's constant column carries the column sums).  Eigenvalues are returned UNSORTED (lam[i]
 * belongs to column i of V); sweeps (may be NULL) receives the number of Jacobi sweeps. */
int kp_sym_eig(kp_ctx* ctx, const double* S, int n, double* V_out, double* lam_out, int* sweeps);

/* ---- lifting ----------------------------------------------------------------------
 * what = KP_LIFT_FULL : lift.full      (Ksysid.m:533; rows x Nfull)  [lift_snapshots :1417-1431]
 *        KP_LIFT_ECON : lift.econ_full (Ksysid.m:1615-1618 / :1443-1491; rows x N)
 *        KP_LIFT_ROW  : one row block of Px as built in get_Koopman (Ksysid.m:1034-1064;
 *                       rows x W): [psi,u] / psi (x) [1;u] / psi([zeta;u])
 * zeta: rows x nzeta, u: rows x m (u may be NULL for FULL/ECON of linear/bilinear).
 * out: rows x width, column-major, caller allocated.
 */
enum { KP_LIFT_FULL = 0, KP_LIFT_ECON = 1, KP_LIFT_ROW = 2 };
int kp_lift(kp_ctx* ctx, const kp_basis* basis, int what, const double* zeta, const double* u,
            int64_t rows, double* out);

/* ---- snapshot pairs resident on the device ----------------------------------------
 * snapshotPairs.alpha / .beta (Ns x nzeta) and .u (Ns x m) of Ksysid.get_snapshotPairs
 * (Ksysid.m:977-979).  The random draw (:974-975) stays on the MATLAB host. */
int kp_snapshots_upload(kp_ctx* ctx, const double* alpha, const double* beta, const double* u,
                        int64_t Ns, int nzeta, int m, kp_snapshots** snaps);
/* Refills an existing object from new host arrays of the same column counts (a caller that hands a new
 * snapshotPairs struct to every Ksysid.get_Koopman call, Ksysid.m:987, keeps two objects and fills them alternately).
 * Returns as soon as the caller's arrays have been copied into the library's pinned staging ring - they may be reused
 * or freed at once; the transfer itself runs on the context's copy stream behind every kernel already enqueued that
 * reads the object and ahead of every later one (events, no device synchronisation), so it overlaps the Gram kernel
 * of another object.  Ns may differ from the previous contents (the arrays grow when needed, which drains the
 * pipeline once).  Both entry points stage through pinned memory in chunks copied by KP_COPY_THREADS host threads
 * (default 4 on hosts with >= 16 cores) while the DMA of earlier chunks is already running. */
int kp_snapshots_update(kp_ctx* ctx, kp_snapshots* snaps, const double* alpha, const double* beta, const double* u,
                        int64_t Ns);
int kp_snapshots_destroy(kp_snapshots* snaps);

/* Page-locked host memory owned by the context (freed by kp_host_free or kp_destroy).  A host side that ASSEMBLES its
 * inputs - evaluate_rand_models.m gathers `data4sysid_all{i}.train{j}.y/.u` of every system before the sweep
 * (evaluate_rand_models.m:45-59; kp_traj_upload takes them as one block) - writes them here once: no first-touch page
 * faults in the gather, and every upload from such a buffer is a direct DMA at PCIe speed instead of a staged copy. */
int kp_host_alloc(kp_ctx* ctx, int64_t bytes, void** ptr);
int kp_host_free(kp_ctx* ctx, void* ptr);

/* ---- EDMD fit ---------------------------------------------------------------------
 * kp_fit_gram: the fused per-row lift loop of get_Koopman (Ksysid.m:1030-1065) and the
 *   accumulations PxTPx = Px'*Px (:1114), PxTPy = Px'*Py (:1125).  Px/Py are never
 *   materialised.  G, C: W x W column-major, caller allocated (either may be NULL).
 * kp_fit_solve: K = Px \ Py (Ksysid.m:1069) from the normal equations G K = C by Cholesky.
 *   ncols = columns of C.  When Psi is rank deficient (arm marker data without dim_red) MATLAB's `\`
 *   warns and returns a BASIC solution from QR with column pivoting; here: Cholesky with diagonal pivoting selects the
 *   column subset (same greedy rule in Gram space), K is the least-squares solution over those columns and zero in the
 *   other rows, the call returns KP_OK, kp_last_error() carries the warning and kp_fit_last_rank the rank (W when G was
 *   positive definite).  kp_fit does the same on its synchronous path; the asynchronous pipeline (K_out == NULL) reports
 *   KP_ERR_NOT_SPD from kp_synchronize instead.
 * kp_fit_lasso: solve_KoopmanQP (Ksysid.m:1095-1176, delays = 0):
 *   min 1/2||Px K - Py||_F^2  s.t. ||vec K||_1 <= t,  t = lasso * N (:996).
 * kp_fit: get_Koopman end to end for n_lasso values (train_models loop :1372-1387) on
 *   resident snapshots; lasso[i] >= 1e6 (or +Inf) selects the least-squares branch (:1068).
 *   K_out: n_lasso matrices W x W, column-major, back to back (may be NULL: results stay
 *   on the device; fetch with kp_fit_get_K).
 */
int kp_fit_gram(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, double* G, double* C);
int kp_fit_solve(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, double* K);
int kp_fit_last_rank(const kp_ctx* ctx, int* rank);
/* min_i L_ii^2 / G_ii of the most recent synchronous least-squares solve (kp_fit with K_out, kp_fit_solve): about
 * 1 / cond(G) = 1 / cond(Px)^2.  MATLAB's `\` (Ksysid.m:1069) is a QR solve; the normal equations lose cond(G) eps against
 * it, so a ratio above ~1e-5 means K is already at QR accuracy (1e-11) and kp_fit_refine's pass over the data is not needed. */
int kp_fit_last_pivot_ratio(const kp_ctx* ctx, double* ratio);
int kp_fit_lasso(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, double t,
                 int max_iter, double tol, double* K, int* iters);
/* nv lasso values on the same Grams at once (the train_models loop over a lasso vector, Ksysid.m:1372-1387, re-lifts
 * and re-solves per value; here the values share G, C, the least-squares solution and every G*[K_1..K_nv] product).
 * t: nv budgets; K: nv matrices W x ncols back to back; iters: nv counts (may be NULL; 0 = constraint inactive).
 * Each value ends either by convergence of the projected-gradient iteration (relative change <= tol) or, earlier, when
 * the active-set candidate built from its current support satisfies every optimality condition of the QP of
 * Ksysid.m:1126-1137 (then K is that QP's optimum to rounding), or - values still running after 24 iterations when W <= 136 or cond(G) > 1e7 (1 000 otherwise): the
 * ill-conditioned Grams of monomial dictionaries on real data - by the regularisation-path homotopy (ONE LARS-with-drops path
 * per column of K serves all of them; exact active-set optimum, |K|_1 = t to 1e-13; kp_timer_get(11) tells).
 * KP_ERR_NOT_CONVERGED: iteration cap (K still written). */
int kp_fit_lasso_batch(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, const double* t, int nv,
                       int max_iter, double tol, double* K, int* iters);
int kp_fit(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, const double* lasso,
           int n_lasso, double* K_out);
int kp_fit_get_K(kp_ctx* ctx, int index, int W, double* K);
/* Iterative refinement of a least-squares K (W x W, column-major, in/out) with the residual taken from the data:
 * `steps` times K += G^-1 Px'(Py - Px K).  MATLAB's `\` (Ksysid.m:1069) is a QR solve; the normal equations of
 * kp_fit lose cond(Px)^2 eps, which one or two of these steps recover (cond 2e5: 4e-6 -> 1e-10).  Materialises Px, Py
 * (3 Ns W doubles of device memory): the accuracy path, not the throughput path. */
int kp_fit_refine(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, int steps, double* K);
/* kp_fit with K_out == NULL and one least-squares value is ASYNCHRONOUS: it returns when the work is
 * enqueued; the solve runs on a second HIP stream so that it overlaps the fused Gram kernel of the
 * next kp_fit call (sweeps over many fits, train_models loop Ksysid.m:1372-1387 / evaluate_rand_models.m:45-144).
 * kp_synchronize waits for everything in flight and returns KP_ERR_NOT_SPD if a deferred fit failed; kp_fit_get_K,
 * kp_fit_gram, kp_fit_solve and kp_destroy synchronise implicitly.
 * The fits issued since the previous kp_synchronize form a BATCH; fit number q of the batch keeps its K in slot
 * q mod n_slots of a device ring, and kp_fit_get_K(ctx, q, W, K) fetches it after the batch (the last n_slots fits are
 * retrievable; n_slots defaults to 128, kp_fit_async_slots changes it - 8 W^2 bytes of HBM per slot; the queued [G | C]
 * pairs of up to min(128, n_slots) fits are solved by one batched launch sequence, whose padded workspace - about 32 W^2 bytes
 * per queued fit - is reserved at the first asynchronous fit).  A fit with
 * another dictionary, width or snapshot count than the fits in flight drains the pipeline first (its buffers are
 * shared), which also closes the batch. */
int kp_synchronize(kp_ctx* ctx);
int kp_fit_async_slots(kp_ctx* ctx, int n_slots);

/* Batched small fits (evaluate_rand_models.m:45-144: one Ksysid fit per random system): nb independent systems
 * with the same dictionary, W <= 16, no dimension reduction, least squares (lasso = Inf, Ksysid.m:1068-1069).
 * `snaps` holds the merged snapshot pairs of all systems, nb x Ns_each rows, system s in rows
 * [s*Ns_each, (s+1)*Ns_each).  One workgroup per system lifts its rows, accumulates Px'Px / Px'Py and solves.
 * The normal-equations solution is followed by one step of iterative refinement with the residual taken from the
 * data, K += G^-1 Px'(Py - Px K) (a second sweep over the system's snapshots): the degree-13 dictionaries of the sweep
 * have cond(Px) ~ 1e5, and the reference's `\` is a QR solve.  KP_BATCH_REFINE=0..4 overrides the number of steps.
 * K_out, G_out, C_out: nb matrices W x W, column-major, back to back (G_out/C_out may be NULL);
 * status_out[s] != 0 (may be NULL) marks a Gram matrix that is not numerically positive definite (K = NaN). */
int kp_fit_batch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, int nb, int64_t Ns_each,
                 double* K_out, double* G_out, double* C_out, int* status_out);

/* The random-system sweep with the data resident on the device (evaluate_rand_models.m:45-144).
 * kp_traj_upload: data4sysid of nb systems that share one layout (the generated / shipped rand-systems sets): per system
 *   the merged training trials (Ksysid.merge_trials :380-401: ntrials x T rows, column-major rows x n and rows x m, RAW
 *   values), and the validation trial (Tv rows).  Y: nb blocks of rows x n; U: nb blocks of rows x m; Yv / Uv likewise.
 *   The per-system scaling of get_scale (:180-229) is computed on the device (kp_traj_scale fetches it: per system
 *   [y offset (n) | y factor (n) | u offset (m) | u factor (m)]).
 * kp_sweep_eval: one (model type, dictionary) of the sweep for ALL systems in one call: scaling + snapshot pairs
 *   (get_snapshotPairs :941-978, delays = 0, all pairs, the last good one dropped) + fit (as kp_fit_batch; `lasso`
 *   >= 1e6 or Inf: least squares, else the L1 budget lasso * N of :996, evaluate_rand_models.m:122) + model
 *   extraction (get_model with the M-projection / get_BLmodel / get_NLmodel) + validation rollout (val_model /
 *   val_BLmodel / val_NLmodel) + normalised mean error (evaluate_rand_models.m:69-72).  err_out: nb x n
 *   (system-major); K_out (nb x W x W) and status_out (nb, != 0: singular Gram, err = NaN) may be NULL.
 * kp_traj_destroy: before kp_destroy of its context, like every handle (the object's device blocks return to a small pool of
 *   the context, which the next kp_traj_upload draws from and kp_destroy frees). */
typedef struct kp_traj kp_traj;
int kp_traj_upload(kp_ctx* ctx, const double* Y, const double* U, int nb, int ntrials, int T, int n, int m,
                   const double* Yv, const double* Uv, int Tv, kp_traj** traj);
int kp_traj_destroy(kp_traj* traj);
int kp_traj_scale(kp_traj* traj, double* sc_out);
/* the layout an object was created with (any pointer may be NULL): a gateway that only keeps the handle sizes its outputs with it */
int kp_traj_dims(const kp_traj* traj, int* nb, int* ntrials, int* T, int* n, int* m, int* Tv);
/* kp_traj_upload in three steps, for callers that assemble the four blocks one after the other (the reference hands
 * evaluate_rand_models.m:45-59 a cell array of data4sysid structs: thousands of small trial arrays to be gathered): every
 * block is on its way to the device while the next one is prepared.  kp_traj_create: the device blocks; kp_traj_put: block
 * `which` (0 Y, 1 U, 2 Yv, 3 Uv; layouts as kp_traj_upload) by ONE host-to-device copy enqueued on the context's stream -
 * asynchronous when `host` is page-locked (kp_host_alloc), which must then stay untouched until kp_traj_finish;
 * kp_traj_finish: scaling on the device, stream synchronised - only then may the object be used. */
int kp_traj_create(kp_ctx* ctx, int nb, int ntrials, int T, int n, int m, int Tv, kp_traj** traj);
int kp_traj_put(kp_traj* traj, int which, const double* host);
int kp_traj_finish(kp_traj* traj);
int kp_sweep_eval(kp_ctx* ctx, const kp_traj* traj, const kp_basis* basis, double lasso, double* err_out, double* K_out,
                  int* status_out);
/* kp_sweep_eval for degrees 1..n_deg of a POLYNOMIAL dictionary from ONE pass over the data (evaluate_rand_models.m
 * :47-143 loops degree by degree; def_polyLift orders monomials by total degree, Ksysid.m:645-648, so the degree-j
 * dictionary is a column subset of the degree-D one and its Grams are sub-blocks).  `basis` is the dictionary of the
 * highest degree.  The Grams are accumulated in a Chebyshev internal basis (data are scaled to [-1, 1]) and mapped back
 * exactly (K_m = T^-1 K_c T): QR-level accuracy of K without a refinement sweep.  err_out: n_deg x nb x n
 * (degree-major); status_out (n_deg x nb, may be NULL).  kp_sweep_nested_get_K fetches the K matrices (monomial basis,
 * W_j x W_j each) of one degree of the most recent call (parity checks). */
int kp_sweep_eval_nested(kp_ctx* ctx, const kp_traj* traj, const kp_basis* basis, double lasso, int n_deg, double* err_out,
                         int* status_out);
int kp_sweep_nested_get_K(kp_ctx* ctx, int nb, int Wmax, int n_deg, int deg_index, int W, double* K_out);

/* Model extraction with the M-projection of get_model (Ksysid.m:1206-1225): from K and
 * the Grams (no second pass over the data): L'L = [A B] G [A B]', L'R = [A B] C(:,1:N).
 * A_out N x N, B_out N x m (= M*A, M*B), M_out N x N.  Linear models only. */
int kp_model_project(kp_ctx* ctx, const double* K, const double* G, const double* C, int N, int m,
                     double* A_out, double* B_out, double* M_out);

/* kp_model_project for nb small linear models at once (W = N + m <= 16; evaluate_rand_models.m fits one per system,
 * :49-59 -> get_model, Ksysid.m:1206-1225): K, G, C are nb blocks W x W, column-major, back to back (as kp_fit_batch
 * returns them); A_out nb x (N x N), B_out nb x (N x m), M_out nb x (N x N) or NULL.  One workgroup per model;
 * status_out[s] != 0 (may be NULL) marks a singular L'L (A, B = NaN). */
int kp_model_project_batch(kp_ctx* ctx, const double* K, const double* G, const double* C, int nb, int N, int m,
                           double* A_out, double* B_out, double* M_out, int* status_out);

/* ---- validation rollouts ------------------------------------------------------------
 * val_model (Ksysid.m:1678-1689) z+ = A z + B u ; val_BLmodel (:1772-1787)
 * z+ = A z + B kron(I_m,z) u.  z0: N, U: T x m (rows = steps), Y: T x n_out with
 * Y(1,:) = Cz0 ... (C = first n_out entries of z, Ksysid.m:1203).  batch independent
 * models may be rolled out at once (arrays back to back). */
int kp_rollout(kp_ctx* ctx, int model_type, int batch, const double* A, const double* B, int N, int m,
               const double* z0, const double* U, int T, int n_out, double* Y);

/* val_NLmodel (Ksysid.m:1848-1863): zeta+ = F(zeta,u) = Kf * lift.econ_full([zeta;u]) with
 * Kf = K(:,1:nzeta)' (nzeta x N, Ksysid.m:1329).  basis must be a 'nonlinear' dictionary.
 * zeta0: batch x nzeta, U: T x m per rollout, Z out: T x nzeta per rollout (column-major). */
int kp_rollout_nl(kp_ctx* ctx, const kp_basis* basis, int batch, const double* Kf, const double* zeta0,
                  const double* U, int T, double* Z);

/* ---- MPC --------------------------------------------------------------------------
 * kp_mpc_create: Kmpc constructor for the linear-MPC types: get_costMatrices
 *   (Kmpc.m:157-211) / get_costMatrices_bilinear (:517-559) and
 *   get_constraintMatrices(_bilinear) (:214-326, :626-738).
 *   A: N x N; B: N x m (linear) or N x (N*m) (bilinear, blocks B_i, Ksysid.m:1259);
 *   proj: nproj x N (projmtx); q_run, q_term (cost_running/terminal, :197-198);
 *   r: m input weights = diag of eye(m).*cost_input (:201);
 *   lo, hi: m scaled-down input bounds (:247,659) or NULL; slope_lim =
 *   input_slopeConst*mean(u_factor) (:272,684) or NaN; smooth_lim (:294,706) or NaN.
 *   State bounds (:300-318, :716-730) are set afterwards with kp_mpc_set_state_bounds.
 * kp_mpc_set_state_bounds: n scaled-down lower / upper bounds (scaledown.y(state_bounds')', :313); n = 0 removes them.
 *   The reference writes the kron block of E into the first (Np+1) n columns of the stacked lifted state (:306) - not
 *   strided by N - and that is reproduced: the bounded entries are s = i n + j of [z_0; ...; z_Np].  The rows are dense
 *   in U and depend on z for bilinear models: such steps assemble the dense constraint matrix per problem and go through
 *   the generic QP kernel (one wave per problem) - single steps, kp_mpc_step_batch and iters > 1 alike; with iters > 1 the
 *   constraint matrix stays the one of the current state, as in the reference (A = get_constraintL_bilinear(zrow) is
 *   formed before the iteration loop, Kmpc.m:861), only H and f follow the predicted lifted states (:874-899).
 * kp_mpc_step: one get_mpcInput (Kmpc.m:329-387) / get_mpcInput_bilinear_iter
 *   (:817-904) call: z is the lifted state (N), u_prev = traj.u(end,:) (m), Yr the
 *   padded, vectorised reference (nproj*(Np+1), :354-365); iters as in :874.
 *   U_out: Np x m column-major (row 1 = pinned current input; Ksim applies row 2,
 *   Ksim.m:225).  On QP failure U_out is filled with NaN and *status = KP_ERR_QP_FAIL
 *   (quadprog_gurobi.m:18-23, Ksim.m:220-222); the function itself still returns KP_OK.
 * kp_mpc_step_zeta: same, with the lift z = lift.econ_full(zeta) (Kmpc.m:842) fused in
 *   front (one launch); z_out (N) may be NULL.
 * kp_mpc_step_batch: nb independent problems of the same controller (random-system
 *   sweeps, Monte-Carlo closed loops); arrays are column-major with nb columns.
 * kp_qp_solve: generic shim with the signature of quadprog_gurobi(H,f,A,b)
 *   (quadprog_gurobi.m:1; call sites Kmpc.m:382,809,882): min 1/2 x'Hx + f'x, Ax <= b.
 */
int kp_mpc_create(kp_ctx* ctx, int model_type, const double* A, const double* B, int N, int m, int Np,
                  const double* proj, int nproj, double q_run, double q_term, const double* r,
                  const double* lo, const double* hi, double slope_lim, double smooth_lim,
                  kp_mpc** mpc);
int kp_mpc_set_state_bounds(kp_mpc* mpc, int n, const double* lo, const double* hi);
int kp_mpc_destroy(kp_mpc* mpc);
int kp_mpc_dims(const kp_mpc* mpc, int* nvar, int* nrows);
int kp_mpc_step(kp_mpc* mpc, const double* z, const double* u_prev, const double* Yr, int iters,
                double* U_out, int* status);
int kp_mpc_step_zeta(kp_mpc* mpc, const kp_basis* basis, const double* zeta, const double* u_prev,
                     const double* Yr, int iters, double* U_out, double* z_out, int* status);
int kp_mpc_step_batch(kp_mpc* mpc, int nb, const double* z, const double* u_prev, const double* Yr,
                      double* U_out, int* status);
/* The QP data (2H, f, A_ineq, b) of the most recent kp_mpc_step, for parity checks against
 * the literal assembly of Kmpc.m:861-883.  Hq: nvar x nvar, Aq: nrows x nvar (column-major). */
int kp_mpc_last_qp(kp_mpc* mpc, double* Hq, double* f, double* Aq, double* bq);
/* Diagnostics (the reference only has tic/toc around the call, Ksim.m:205-217): device-side phase
 * times in microseconds of the most recent kp_mpc_step: us[0] lift + reference error, [1] Beta/S_k,
 * [2] H and f, [3] H^-1, [4] active-set iterations, [5] whole kernel; counts[0] solver
 * iterations, counts[1] active constraints at the optimum. */
int kp_mpc_last_profile(kp_mpc* mpc, double* us, int* counts);
/* All 16 device-side stamps of that step relative to its first one, in microseconds ([8], [9]: the two counts):
 * [1] tracking error formed, [2] S_k, [3] H and f, [4] solver entered, [5] solved, [10] inputs landed,
 * [11] lifted state, [6] iteration loop entered, [12] H^-1, [13] warm-start products, [14] inverse of the warm set's Schur complement. */
int kp_mpc_last_stamps(kp_mpc* mpc, double* us16);
int kp_qp_solve(kp_ctx* ctx, const double* H, const double* f, const double* A, const double* b, int n,
                int mrows, double* x, int* status);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------------------------------------
 * The reference has no parallel construct; its sweeps are serial loops over independent units:
 * lasso values (Ksysid.train_models, Ksysid.m:1372-1387), random systems x model types x degrees
 * (evaluate_rand_models.m:45-144).  Units are dealt round-robin over the ranks by the host and there is NO collective on
 * the data path; the entry points below are the final gather and, for ONE fit sharded over snapshots, the single
 * all-reduce of the Gram pair.  librccl.so is loaded on the first kp_comm_* call; single-GPU use never needs it.
 *   kp_comm_unique_id   rank 0 creates the 128-byte RCCL id and hands it to the other ranks (file, socket, MATLAB
 *                       parallel pool message - the host's business);
 *   kp_comm_create      every rank: ncclCommInitRank on the context's device; kp_destroy releases it;
 *   kp_comm_allgather   host buffers, `bytes` per rank in, world x bytes out (error tables, timing);
 *   kp_comm_allreduce_sum  host vector, in place (barrier, counters);
 *   kp_comm_allgather_fit  K of fit `index` (kp_fit_get_K numbering) of every rank, device to device, one copy out:
 *                       K_all = world matrices W x W;
 *   kp_comm_allgather_fits  the K STACK of a shard: fits first .. first + count - 1 of every rank (the lasso grid of
 *                       train_models, Ksysid.m:1372-1387, values dealt round-robin: rank r holds values r, r + world, ...),
 *                       ONE ncclAllGather of count W^2 doubles straight from the result buffer and one copy out (give it a
 *                       kp_host_alloc block: direct DMA); K_all = world x count matrices W x W, rank-major.  A rank whose
 *                       shard is shorter than `count` contributes unspecified padding slots;
 *   kp_fit_sharded / kp_fit_gram_sharded  kp_fit / kp_fit_gram on this rank's shard of the snapshot pairs with the
 *                       all-reduce of [G | C] (2 W^2 doubles) between the Gram kernel and the solve: every rank gets the
 *                       K of the whole data set. */
int kp_comm_unique_id(void* id128);
int kp_comm_create(kp_ctx* ctx, const void* id128, int rank, int world);
int kp_comm_destroy(kp_ctx* ctx);
/* the caller's watchdog gave up on a kp_comm_create that is still blocked in another thread: the context stays without a
 * communicator even if that call returns later (the launch has agreed on another backend meanwhile) */
int kp_comm_abandon(kp_ctx* ctx);
int kp_comm_info(const kp_ctx* ctx, int* rank, int* world);
int kp_comm_allgather(kp_ctx* ctx, const void* send, int64_t bytes, void* recv);
int kp_comm_allreduce_sum(kp_ctx* ctx, double* inout, int64_t count);
int kp_comm_allgather_fit(kp_ctx* ctx, int index, int W, double* K_all);
int kp_comm_allgather_fits(kp_ctx* ctx, int first, int count, int W, double* K_all);
/* kp_comm_allgather_fits to ONE rank: every other rank sends its `count` matrices to `root` (ncclSend / ncclRecv, one group)
 * and returns without a host copy; K_root (world x count matrices, rank-major) is written on the root only and may be NULL
 * elsewhere.  The reference's caller of train_models is one host (Ksysid.m:1370-1387): one rank wants the candidates. */
int kp_comm_gather_fits(kp_ctx* ctx, int root, int first, int count, int W, double* K_root);
int kp_fit_sharded(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps_local, const double* lasso,
                   int n_lasso, double* K_out);
int kp_fit_gram_sharded(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps_local, double* G, double* C);

/* ---- multi-GPU: ONE caller, several GPUs -----------------------------------------------------------------
 * The reference's host is one MATLAB interpreter running serial loops (Ksim.m:147, evaluate_rand_models.m:45,
 * Ksysid.m:1372); without the Parallel Computing Toolbox it cannot start one worker per GPU.  kp_multi_* serve that shape: the
 * library owns one context and one worker thread per listed device, a call deals its units over them and returns when every
 * device has written ITS share of the result into the caller's arrays - by direct DMA when they lie in a kp_multi_host_alloc
 * block (page-locked for every device), staged through pinned scratch otherwise.  No collective; no RCCL.  The same device may
 * be listed more than once (two contexts): that is how the fan-out is tested on a one-GPU box.
 *   kp_multi_fit           the lasso grid of train_models (Ksysid.m:1372-1387): value i goes to device i mod n_dev; every
 *                          device uploads the snapshot pairs, lifts them ONCE and solves its values as one batch (kp_fit);
 *                          K_out: n_lasso matrices W x W in value order.
 *   kp_multi_fit_sharded   ONE fit sharded over SNAPSHOTS (rows dealt in contiguous ranges): every device's fused Gram kernel
 *                          on its rows, its [G | C] (2 W^2 doubles) to device 0 by a peer copy (xGMI), summed there in device
 *                          order, one solve; lasso may be NULL (least squares).
 *   kp_multi_traj_upload / kp_multi_sweep_eval_nested   the random-system sweep (evaluate_rand_models.m:45-144): systems dealt
 *                          in contiguous chunks (layouts as kp_traj_upload / kp_sweep_eval_nested, tables in system order).
 *   kp_multi_mpc_*         kp_mpc_step_batch with the problems dealt in contiguous chunks (arguments as kp_mpc_create /
 *                          kp_mpc_set_state_bounds / kp_mpc_step_batch).
 *   kp_multi_timers        per device [upload, device work, result transfer, whole job] of the most recent call, ms (wall).
 *   kp_multi_ctx           worker i's context (kp_device_info, kp_timer_get, kp_last_error; not for concurrent use).
 * Lifetime: kp_multi_traj and kp_multi_mpc objects point into the workers' contexts - destroy them BEFORE kp_multi_destroy.  One
 * call at a time per kp_multi object: calls from several threads are serialised by it. */
typedef struct kp_multi kp_multi;
typedef struct kp_multi_traj kp_multi_traj;
typedef struct kp_multi_mpc kp_multi_mpc;
int kp_multi_create(const int* device_ids, int n_dev, kp_multi** mg);
int kp_multi_destroy(kp_multi* mg);
int kp_multi_size(const kp_multi* mg, int* n_dev);
kp_ctx* kp_multi_ctx(kp_multi* mg, int i);
const char* kp_multi_last_error(const kp_multi* mg);
int kp_multi_host_alloc(kp_multi* mg, int64_t bytes, void** ptr);
int kp_multi_host_free(kp_multi* mg, void* ptr);
int kp_multi_timers(const kp_multi* mg, double* ms);
int kp_multi_fit(kp_multi* mg, const kp_basis_desc* desc, const double* alpha, const double* beta, const double* u, int64_t Ns,
                 const double* lasso, int n_lasso, double* K_out);
int kp_multi_fit_sharded(kp_multi* mg, const kp_basis_desc* desc, const double* alpha, const double* beta, const double* u,
                         int64_t Ns, const double* lasso, int n_lasso, double* K_out);
int kp_multi_traj_upload(kp_multi* mg, const double* Y, const double* U, int nb, int ntrials, int T, int n, int m,
                         const double* Yv, const double* Uv, int Tv, kp_multi_traj** traj);
int kp_multi_traj_destroy(kp_multi_traj* traj);
int kp_multi_sweep_eval_nested(kp_multi* mg, const kp_multi_traj* traj, const kp_basis_desc* desc, double lasso, int n_deg,
                               double* err_out, int* status_out);
int kp_multi_mpc_create(kp_multi* mg, int model_type, const double* A, const double* B, int N, int m, int Np,
                        const double* proj, int nproj, double q_run, double q_term, const double* r,
                        const double* lo, const double* hi, double slope_lim, double smooth_lim, kp_multi_mpc** mpc);
int kp_multi_mpc_set_state_bounds(kp_multi_mpc* mpc, int n, const double* lo, const double* hi);
int kp_multi_mpc_destroy(kp_multi_mpc* mpc);
int kp_multi_mpc_step_batch(kp_multi_mpc* mpc, int nb, const double* z, const double* u_prev, const double* Yr,
                            double* U_out, int* status);

#ifdef __cplusplus
}
#endif
#endif /* KOOPMAN_HIP_H */

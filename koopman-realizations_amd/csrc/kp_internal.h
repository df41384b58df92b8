// Internal declarations shared by the translation units of libkoopman_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "koopman_hip.h"

#define KP_MAX_VARS 32

void kp_set_global_error(const std::string& s);
struct kp_comm_state;   // RCCL communicator of a multi-process run (kp_comm.hip)

struct kp_stage;   // kp_upload.hip: copy stream, pinned staging ring and copy threads of the host -> HBM path

struct kp_ctx {
  int device = 0;
  kp_stage* stage = nullptr;
  std::vector<void*> host_blocks;       // kp_host_alloc: page-locked host buffers handed to the caller
  // device blocks of destroyed trajectory objects, kept for the next kp_traj_upload (a sweep that uploads chunk after chunk
  // paid 5 hipMalloc + 5 hipFree per chunk: 1.3 ms of the 16.5 ms of 1024 systems); at most KP_TRAJ_POOL_MAX blocks,
  // freed by kp_destroy
  std::vector<std::pair<void*, size_t>> traj_pool;
  std::mutex host_mu;                   // ... a gathering host thread may ask for one while another thread drives the device
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t evp[6] = {nullptr};   // gram start, gram end, reduce end, solve end, spare x2
  // asynchronous fit pipeline (kp_fit with K_out == NULL): Gram on `stream`, solve on `stream2`
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_gram_done = nullptr, ev_pad_done = nullptr, ev_pad_done2 = nullptr, ev_solve0 = nullptr, ev_solve1 = nullptr;
  bool pad_pending = false, pad_pending2 = false, async_pending = false;
  int gc_flip = 0;             // asynchronous fits alternate between the two halves of GC
  // results of asynchronous fits: ring of kring_cap slots in Kres; fit number q of the current batch (the fits since
  // the last kp_synchronize) lives in slot q % kring_cap
  int kring_cap = 128;   // = the default solve batch (solve_batch_size): a full 128-fit batch needs no kp_fit_async_slots call
  int async_count = 0;         // asynchronous fits issued in the current batch
  bool batch_closed = true;    // kp_synchronize closed the batch: the next asynchronous fit starts a new one
  bool kres_is_ring = false;   // Kres currently holds an asynchronous batch (kp_fit_get_K indexes the ring)
  const void* pend_basis = nullptr;   // dictionary / snapshot count / width of the fits in flight: a change drains the pipeline
  int64_t pend_Ns = 0;
  int pend_W = 0;
  // deferred solves of the asynchronous pipeline: the Gram pairs of up to solve_batch fits wait in the [G | C] ring and are
  // factored / solved by ONE batched launch sequence (no CUs held back for a concurrent solve stream)
  int pend_solves = 0, pend_first = 0;
  bool ring_timing = false;           // Gram launchers time themselves with the event ring and record nothing else
  unsigned ring_skip = 0;
  kp_comm_state* comm = nullptr;      // set by kp_comm_create: rank / world / RCCL communicator
  std::atomic<bool> comm_abandoned{false};   // kp_comm_abandon: a bootstrap still blocked in another thread must not publish `comm`
  bool gc_preloaded = false;          // kp_multi_fit_sharded: ctx->GC already holds the summed [G | C] of all devices - kp_fit skips its Gram launch
  bool reduce_grams = false;          // kp_fit_sharded: all-reduce [G | C] over the ranks between the Gram kernel and the solve
  // set by the asynchronous kp_fit around kp_gram_dispatch: the split-partial reduction runs on this stream
  // (after an event recorded behind the main Gram kernel) and the partial buffer `part_flip` is used
  hipStream_t reduce_stream = nullptr;
  hipEvent_t ev_main_done = nullptr;
  // start / end events of the most recent pipelined Gram launches (ring of KP_RING pairs): kp_synchronize reports the
  // MEAN kernel duration over them (timer 0), not just the last launch
  hipEvent_t ring[2 * 64] = {};
  int ring_pos = 0, ring_n = 0;
  bool solve_chained = false;   // set by a Gram launch that already made the solve stream wait for it
  int part_flip = 0;
  int reduce_timed_from = 1;   // evp index that marks the start of the last partial reduction
  int* sticky_info = nullptr;       // device word: set by any deferred factorisation that hit a non-positive pivot
  bool test_hooks = false;          // KP_TEST_HOOKS was set when the context was created: kp_fit honours KP_RANK_HINT_TEST (tests only)
  void* pin_scratch = nullptr;      // growable page-locked host scratch (kp_pinned_scratch): small results read back by direct DMA
  size_t pin_scratch_bytes = 0;
  double* pin_small = nullptr;      // 64 bytes of page-locked host memory: small results (info word + pivot ratio) come back in ONE direct DMA
  int reserve_cus = 0;              // CUs left free by the Gram grid so the solve of the previous fit can run beside it
  int num_cu = 0;
  int64_t hbm_bytes = 0;
  std::string name;
  mutable std::string err;
  double timers[12] = {0};
  double gram_flops_per_pair = 0;
  // growable device workspaces
  void* ws[20] = {nullptr};     // slot 8: staging of the collectives, 9: rank-revealing solve, 10 / 11: Grams of the shadow dictionary of a dim_red fit, their half-transformed form
  size_t ws_bytes[20] = {0};   // 12 / 13: column states and results of the lasso homotopy (kp_lasso_path.hip); 14: econ-lifted rows of a dim_red fit (kp_gram3.hip)
                                // 15 / 16: lifted snapshot panels of a wide dictionary, 17: split partials of its products, 18: scratch of the blocked factorisation (kp_wide.hip, kp_fit.hip)
  int last_rank = -1;           // rank found by the most recent solve (W when the Gram matrix was positive definite)
  double last_pivot_ratio = 1.0; // min_i L_ii^2 / G_ii of the most recent synchronous least-squares solve (~1 / cond(G))
  // results of the last kp_fit
  double* Kres = nullptr;   // n_lasso x W x W
  size_t Kres_bytes = 0;
  int Kres_W = 0, Kres_n = 0;
  // last gram (device): G then C, W x W each, column-major
  double* GC = nullptr;
  size_t GC_bytes = 0;
  int GC_W = 0;

  int fail(int code, const std::string& s) const {
    err = s;
    kp_set_global_error(s);
    return code;
  }
  void* workspace(int slot, size_t bytes);  // returns nullptr on failure
};

#define KP_HIP(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return (ctx)->fail(KP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

// Column kinds of the full basis (device table, one entry per full-basis column)
enum : int { COL_VAR = 0, COL_MONO = 1, COL_FOURIER = 2, COL_GAUSS = 3, COL_CONST = 4, COL_HERMITE = 5, COL_FSPARSE = 6 };

struct ColDesc {
  int32_t kind;
  int32_t arg;  // VAR: variable index; MONO/HERMITE: row in exps; FSPARSE: first of two rows in exps (sin, cos
                // multipliers); FOURIER: mixed-radix index (>=1); GAUSS: centre
  int32_t aux;  // FOURIER: degree; MONO with <= 8 variables (kp_basis_create): exponent bytes of variables 0-3
  int32_t pad;  // MONO with <= 8 variables: exponent bytes of variables 4-7; FOURIER (<= 8 variables, degree <= 7): the digits of the
                // mixed-radix index, four bits per variable (0: not packed)
};

// Device view of a dictionary, passed by value to kernels.
struct BasisDev {
  int model_type, nzeta, m, nvars;
  int nfull, N, W, k_pcs;
  const ColDesc* cols;    // nfull
  const uint8_t* exps;    // n_mono x nvars
  const double* centres;  // n_gauss x nvars (centre-major)
  const double* pcs;      // nfull x k_pcs column-major
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: one cache entry per (call site, device), so a
// second context on another GPU of the same process gets its own attribute call
struct KpLdsCache { size_t set[32] = {}; };
inline hipError_t kp_ensure_lds(KpLdsCache& c, const void* func, size_t lds) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (lds > c.set[dev]) {
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    c.set[dev] = lds;
  }
  return hipSuccess;
}

struct kp_gram_plan;
void kp_gram_plan_free(kp_gram_plan* p);
struct kp_gram2_plan;
void kp_gram2_plan_free(kp_gram2_plan* p);
struct kp_gram3_plan;
void kp_gram3_plan_free(kp_gram3_plan* p);
struct kp_gram5_plan;
void kp_gram5_plan_free(kp_gram5_plan* p);

struct kp_basis {
  kp_ctx* ctx = nullptr;
  BasisDev dev{};
  void* d_cols = nullptr;
  void* d_exps = nullptr;
  void* d_centres = nullptr;
  void* d_pcs = nullptr;
  void* d_recipes = nullptr;   // [nfull] uint32: 4 x 8-bit power-table ids (255 = 1.0)
  void* d_pcsT = nullptr;      // dim_red: pcs as [full column][32 components], zero padded (kp_gram3_prelift_kernel; built on first use)
  int max_degree = 0;
  mutable int rank_hint = 0;   // rank found by the last synchronous fit of this dictionary when its Gram matrix was rank deficient (0: it was not):
                               // only the ORDER of the next fit's work depends on it (kp_fit), never its result
  bool pure_fourier = false;   // every column behind the variables is the constant or a fourier function of fourier_degree with packed digits (ColDesc::pad)
  int fourier_degree = 0;      // degree of the dictionary's fourier block (0: none, or blocks of different degrees): kp_lift_kernel's harmonic table
  int pow_depth = 1;           // largest single-variable exponent
  bool fast = false;           // every column is a product of <= 4 single-variable powers
  int max_factors = 1;         // largest number of single-variable powers in one column (valid if fast)
  kp_gram_plan* plan = nullptr;  // tile->wave plan of the fused Gram kernel (built on first use)
  kp_gram2_plan* plan2 = nullptr;  // plan of the 4x4x4-MFMA Gram kernel (monomial dictionaries)
  kp_gram3_plan* plan3 = nullptr;  // plan of the Kronecker (bilinear) Gram kernel
  kp_gram5_plan* plan5 = nullptr;  // plan of the dense 4x4x4 Gram kernel (linear / nonlinear monomial dictionaries)
  kp_basis* shadow_bil = nullptr;  // linear dim_red dictionaries: the same dictionary as a BILINEAR one (shares every device array; kp_gram3.hip)
  kp_basis* shadow_full = nullptr; // linear / nonlinear dim_red dictionaries: the same dictionary WITHOUT the projection (kp_gram3.hip)
  std::vector<uint32_t> h_recipes; // host copy of the recipes (valid if fast)
  // extended recipes of the Kronecker Gram kernel (kp_gram3): besides powers, the per-variable table may hold the
  // harmonics cos / sin(2 pi j x) of fourier blocks (Ksysid.m:694-731) and, behind the per-variable entries, one entry per
  // gaussian centre (Ksysid.m:790-817).  Entry layout per variable: x^1 .. x^Dp | cos(2 pi x), sin(2 pi x), ... (df pairs);
  // recipe ids: < 128 variable entries (v * D + e, D = Dp + 2 df), 128 + c gaussian centre c, 255 the constant
  bool fast_ext = false;
  int ext_Dp = 1, ext_df = 0, ext_ng = 0, ext_max_factors = 1;
  void* d_recipes_ext = nullptr;
};

struct kp_snapshots {
  kp_ctx* ctx = nullptr;
  int64_t Ns = 0;
  int nzeta = 0, m = 0;
  double* alpha = nullptr;  // Ns x nzeta col-major
  double* beta = nullptr;
  double* u = nullptr;      // Ns x m
  int64_t cap_rows = 0;     // rows the arrays were allocated for (kp_snapshots_update refills them in place)
  // kp_snapshots_update (kp_upload.hip): the refill runs on the copy stream; `ev_ready` follows its last DMA, `ev_read`
  // the last kernel that read the arrays (recorded only once the object has been refilled: `streaming`)
  hipEvent_t ev_ready = nullptr, ev_read = nullptr;
  mutable bool dma_pending = false, read_pending = false;
  bool streaming = false;
};
// page-locked host scratch of at least `bytes` (contents not preserved across calls that grow it); nullptr on failure
void* kp_pinned_scratch(kp_ctx* ctx, size_t bytes);
// Timing-only ablation switches (KP_WIDE_NOWEIGHT, KP_PIV_ABL, KP_PM_ABL: they make results WRONG by design) exist only in builds
// with -DKP_ABLATIONS (tools/*_abl*.sh); the shipped library does not look at those variables.
#ifdef KP_ABLATIONS
static inline int kp_abl_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
static inline int kp_abl_int(const char*) { return 0; }
#endif
void kp_stage_destroy(kp_ctx* ctx);
void kp_host_free_all(kp_ctx* ctx);
void kp_traj_pool_free(kp_ctx* ctx);
// Around every launch sequence that reads a snapshot object.  All readers run on ctx->stream, so one wait orders the
// later ones too.
inline hipError_t kp_snaps_acquire(const kp_snapshots* s, hipStream_t st) {
  if (!s->dma_pending) return hipSuccess;
  s->dma_pending = false;
  return hipStreamWaitEvent(st, s->ev_ready, 0);
}
inline hipError_t kp_snaps_release(const kp_snapshots* s, hipStream_t st) {
  if (!s->streaming) return hipSuccess;
  s->read_pending = true;
  return hipEventRecord(s->ev_read, st);
}

// --- device helpers shared by the lift and gram kernels -----------------------------

// Value of full-basis column `c` at the point whose variables are v[0..nvars) (stride vs).
__device__ __forceinline__ double kp_eval_col(const BasisDev& b, const ColDesc c, const double* v, int vs) {
  switch (c.kind) {
    case COL_VAR:
      return v[c.arg * vs];
    case COL_MONO: {
      const uint8_t* e = b.exps + (size_t)c.arg * b.nvars;
      double p = 1.0;
      for (int i = 0; i < b.nvars; ++i) {
        int ei = e[i];
        if (ei) {
          double x = v[i * vs];
          for (int k = 0; k < ei; ++k) p *= x;
        }
      }
      return p;
    }
    case COL_FOURIER: {
      // index = sum_i digit_i * radix^(nvars-1-i); digit 0 -> 1, 2j-1 -> cos(2 pi j x), 2j -> sin(2 pi j x)
      int radix = 2 * c.aux + 1;
      int idx = c.arg;
      double p = 1.0;
      for (int i = b.nvars - 1; i >= 0; --i) {
        int d = idx % radix;
        idx /= radix;
        if (d) {
          int j = (d + 1) >> 1;
          double a = 2.0 * 3.14159265358979323846 * (double)j * v[i * vs];
          p *= (d & 1) ? cos(a) : sin(a);
        }
      }
      return p;
    }
    case COL_GAUSS: {
      const double* ctr = b.centres + (size_t)c.arg * b.nvars;
      double r2 = 0.0;
      for (int i = 0; i < b.nvars; ++i) {
        double d = v[i * vs] - ctr[i];
        r2 += d * d;
      }
      return exp(-r2);
    }
    case COL_HERMITE: {
      // product of physicists' Hermite polynomials: H0 = 1, H1 = 2x, H_{k+1} = 2x H_k - 2k H_{k-1}
      const uint8_t* e = b.exps + (size_t)c.arg * b.nvars;
      double p = 1.0;
      for (int i = 0; i < b.nvars; ++i) {
        int n = e[i];
        if (n) {
          double x = v[i * vs], h0 = 1.0, h1 = 2.0 * x;
          for (int k = 1; k < n; ++k) {
            double h2 = 2.0 * x * h1 - 2.0 * (double)k * h0;
            h0 = h1;
            h1 = h2;
          }
          p *= h1;
        }
      }
      return p;
    }
    case COL_FSPARSE: {
      const uint8_t* sm = b.exps + (size_t)c.arg * b.nvars;  // sine multipliers, then cosine multipliers
      const uint8_t* cm = sm + b.nvars;
      double p = 1.0;
      for (int i = 0; i < b.nvars; ++i)
        if (sm[i]) p *= sin(2.0 * 3.14159265358979323846 * (double)sm[i] * v[i * vs]);
      for (int i = 0; i < b.nvars; ++i)
        if (cm[i]) p *= cos(2.0 * 3.14159265358979323846 * (double)cm[i] * v[i * vs]);
      return p;
    }
    default:
      return 1.0;
  }
}

// host-side launchers implemented in the .hip files
int kp_gram_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
bool kp_gram2_applicable(const kp_basis* basis);
int kp_gram2_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
// picks the 4x4x4-MFMA kernel when the dictionary allows it, else the general kernel
bool kp_gram3_applicable(const kp_basis* basis);
int kp_gram3_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
bool kp_gram_congruence_applicable(const kp_basis* basis);
int kp_gram_congruence_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
bool kp_gram3_linear_applicable(const kp_basis* basis);
int kp_gram3_linear_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
void kp_gram3_shadow_free(kp_basis* basis);
bool kp_gram5_applicable(const kp_basis* basis);
int kp_gram5_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
inline int kp_gram_dispatch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev) {
  if (s->Ns > 0 && (!s->alpha || !s->beta || (s->m > 0 && !s->u)))    // an object whose refill failed part-way (kp_snapshots_update)
    return ctx->fail(KP_ERR_ARG, "kp_fit: the snapshot object holds no device arrays (a failed kp_snapshots_update?)");
  ctx->reduce_timed_from = 1;
  KP_HIP(ctx, kp_snaps_acquire(s, ctx->stream));
  const int rc = kp_gram3_applicable(basis)   ? kp_gram3_launch(ctx, basis, s, GC_dev)
                 : kp_gram_congruence_applicable(basis) ? kp_gram_congruence_launch(ctx, basis, s, GC_dev)
                 : kp_gram3_linear_applicable(basis) ? kp_gram3_linear_launch(ctx, basis, s, GC_dev)
                 : kp_gram5_applicable(basis) ? kp_gram5_launch(ctx, basis, s, GC_dev)
                 : kp_gram2_applicable(basis) ? kp_gram2_launch(ctx, basis, s, GC_dev)
                                              : kp_gram_launch(ctx, basis, s, GC_dev);
  if (rc) return rc;
  KP_HIP(ctx, kp_snaps_release(s, ctx->stream));
  return KP_OK;
}
// kp_chol_ll.hip: left-looking single-workgroup Cholesky for n <= 352 (in place, lower triangle; `info` as kp_chol_kernel)
bool kp_chol_ll_applicable(int n);
hipError_t kp_chol_ll_launch(double* Gp, int n, int nb, int* info, int* sticky, int prof, hipStream_t st, const double* thr = nullptr);
// substitution with a factor the caller already holds (kp_fit.hip)
int kp_factor_substitute_dev(kp_ctx* ctx, double* Lp, int n, double* Cp, int ncp, double* Dinv, hipStream_t st);
int kp_copy_to_host_async(kp_ctx* ctx, const void* src_dev, void* dst_host, size_t bytes, int mapped, hipStream_t s);
int kp_pivchol_solve_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double* K_dev, int* rank, int rank_hint = 0,
                         void* k_host = nullptr, size_t k_bytes = 0, hipEvent_t ev_solved = nullptr, int k_host_mapped = 0);
int kp_comm_allreduce_dev(kp_ctx* ctx, double* buf_dev, size_t count, hipStream_t s);
// queued (deferred) solves of the asynchronous pipeline are launched; nothing is waited for, no status is consumed (kp_fit.hip)
int kp_flush_pending(kp_ctx* ctx);
int kp_snapshots_update_rows(kp_ctx* ctx, kp_snapshots* s, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t ld);
int kp_ensure_gc(kp_ctx* ctx, int W);   // the context's [G | C] buffer for width W (kp_fit.hip)
int kp_lift_dev(kp_ctx* ctx, const kp_basis* basis, int what, const double* dz, const double* du, int64_t rows, double* dout);
// the same with leading dimensions: `rows` rows of input columns ldi apart into output columns ldo apart
int kp_lift_dev_ld(kp_ctx* ctx, const kp_basis* basis, int what, const double* dz, const double* du, int64_t rows, int64_t ldi, double* dout, int64_t ldo);
// dictionaries too wide for the LDS-staged Gram kernels: lifted panels in HBM + TN products on the matrix pipe (kp_wide.hip)
int kp_gram_wide_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev);
int kp_chol_solve_dev(kp_ctx* ctx, double* G_dev, double* C_dev, int W, int ncols, double* K_dev, hipStream_t st = nullptr,
                      hipEvent_t pad_done = nullptr, int* sticky = nullptr);
// Where kp_chol_solve_dev leaves its info word (non-zero: a non-positive pivot) in workspace 5, behind the padded copies of
// G (np x np), C (np x ncp) and the inverted diagonal blocks: the ONE place that knows the layout (kp_fit.hip, kp_lasso.hip and
// kp_more.hip read the word through it).
inline size_t kp_chol_info_offset(int W, int ncols) {
  const size_t np = (size_t)(W + 15) / 16 * 16, ncp = (size_t)(ncols + 15) / 16 * 16;
  return np * np * 8 + np * ncp * 8 + (np / 16) * 256 * 8;
}
int kp_chol_solve_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, int nb, size_t gc_stride, double* K_dev,
                            int k_first, int k_cap, hipStream_t st, hipEvent_t pad_done, int* sticky);
// least-squares solution, PSD guard and Lipschitz constant shared by all lasso values of one fit (kp_lasso.hip)
struct kp_lasso_prep {
  bool ready = false;
  double L = 0.0, l1_ls = 0.0;
  int bad = 0;
  bool guarded = false;    // the 1e-6 PSD guard of Ksysid.m:1117-1120 was applied (cond(Gw) >= lambda_max / 1e-6)
  double pivot_ratio = 1.0; // min_i L_ii^2 / G_ii of the least-squares factorisation (~1 / cond(G))
  double* Kls = nullptr;   // device, W x ncols
  double* Gw = nullptr;    // device copy of G (with the PSD guard applied when needed)
};
int kp_lasso_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double t, int max_iter, double tol,
                 double* K_dev, int* iters, kp_lasso_prep* prep = nullptr);
// the lasso values of one fit by the regularisation-path homotopy (kp_lasso_path.hip); stats: steps, largest support, ms, inverse in memory
int kp_lasso_path_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv, double* const* K_dev,
                            double* stats, bool known_active = false);
// all lasso values of one fit at once (one wide G*[K_1..K_nv] product per FISTA iteration, active-set polish)
int kp_lasso_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv,
                       int max_iter, double tol, double* const* K_dev, int* iters, kp_lasso_prep* prep = nullptr);

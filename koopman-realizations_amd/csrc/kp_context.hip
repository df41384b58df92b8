// Context, dictionary upload, snapshot upload and the standalone lift kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

#include "kp_internal.h"

static std::mutex g_err_mu;
static std::string g_err;

void kp_set_global_error(const std::string& s) {
  std::lock_guard<std::mutex> l(g_err_mu);
  g_err = s;
}

void* kp_ctx::workspace(int slot, size_t bytes) {
  if (ws_bytes[slot] >= bytes && ws[slot]) return ws[slot];
  if (ws[slot]) (void)hipFree(ws[slot]);
  ws[slot] = nullptr;
  ws_bytes[slot] = 0;
  size_t want = bytes + bytes / 4 + 256;
  if (hipMalloc(&ws[slot], want) != hipSuccess) return nullptr;
  ws_bytes[slot] = want;
  return ws[slot];
}

extern "C" int kp_create(int device_id, kp_ctx** out) {
  if (!out) return KP_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    kp_set_global_error(std::string("kp_create: no HIP device: ") + hipGetErrorString(e));
    return KP_ERR_HIP;
  }
  if (device_id < 0 || device_id >= ndev) {
    kp_set_global_error("kp_create: device_id out of range");
    return KP_ERR_ARG;
  }
  kp_ctx* c = new kp_ctx();
  c->device = device_id;
  c->test_hooks = getenv("KP_TEST_HOOKS") != nullptr;
  // Events order work between this context's two streams and time kernels; neither needs the system-scope release
  // (L2 write-back towards the host) that a default event performs after every kernel it follows - results reach the
  // host through explicit copies.  KP_EVENT_SYSTEM_FENCE=1 restores the default.
  const unsigned evf = getenv("KP_EVENT_SYSTEM_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev0, evf)) != hipSuccess || (e = hipEventCreateWithFlags(&c->ev1, evf)) != hipSuccess) {
    kp_set_global_error(std::string("kp_create: ") + hipGetErrorString(e));
    delete c;
    return KP_ERR_HIP;
  }
  for (int i = 0; i < 6; ++i) (void)hipEventCreateWithFlags(&c->evp[i], evf);
  for (int i = 0; i < 2 * 64; ++i) (void)hipEventCreateWithFlags(&c->ring[i], evf);
  (void)hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&c->ev_gram_done, hipEventDisableTiming | evf);
  (void)hipEventCreateWithFlags(&c->ev_pad_done, hipEventDisableTiming | evf);
  (void)hipEventCreateWithFlags(&c->ev_pad_done2, hipEventDisableTiming | evf);
  (void)hipEventCreateWithFlags(&c->ev_main_done, hipEventDisableTiming | evf);
  (void)hipEventCreateWithFlags(&c->ev_solve0, evf);
  (void)hipEventCreateWithFlags(&c->ev_solve1, evf);
  if (hipMalloc((void**)&c->sticky_info, sizeof(int)) == hipSuccess) (void)hipMemset(c->sticky_info, 0, sizeof(int));
  if (hipHostMalloc((void**)&c->pin_small, 64, hipHostMallocDefault) != hipSuccess) c->pin_small = nullptr;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device_id) == hipSuccess) {
    c->num_cu = p.multiProcessorCount;
    c->hbm_bytes = (int64_t)p.totalGlobalMem;
    c->name = std::string(p.name) + " (" + p.gcnArchName + ")";
  }
  *out = c;
  return KP_OK;
}

extern "C" int kp_destroy(kp_ctx* c) {
  if (!c) return KP_OK;
  (void)kp_comm_destroy(c);
  (void)hipSetDevice(c->device);
  kp_stage_destroy(c);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (int i = 0; i < 20; ++i)
    if (c->ws[i]) (void)hipFree(c->ws[i]);
  if (c->sticky_info) (void)hipFree(c->sticky_info);
  if (c->pin_small) (void)hipHostFree(c->pin_small);
  if (c->pin_scratch) (void)hipHostFree(c->pin_scratch);
  if (c->ev_gram_done) (void)hipEventDestroy(c->ev_gram_done);
  if (c->ev_pad_done) (void)hipEventDestroy(c->ev_pad_done);
  if (c->ev_pad_done2) (void)hipEventDestroy(c->ev_pad_done2);
  if (c->ev_main_done) (void)hipEventDestroy(c->ev_main_done);
  if (c->ev_solve0) (void)hipEventDestroy(c->ev_solve0);
  if (c->ev_solve1) (void)hipEventDestroy(c->ev_solve1);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->Kres) (void)hipFree(c->Kres);
  if (c->GC) (void)hipFree(c->GC);
  for (int i = 0; i < 6; ++i)
    if (c->evp[i]) (void)hipEventDestroy(c->evp[i]);
  for (int i = 0; i < 2 * 64; ++i)
    if (c->ring[i]) (void)hipEventDestroy(c->ring[i]);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
  kp_traj_pool_free(c);
  kp_host_free_all(c);
  delete c;
  return KP_OK;
}

extern "C" int kp_device_count(int* count) {
  if (!count) return KP_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return KP_OK;
}

extern "C" const char* kp_last_error(const kp_ctx* c) {
  if (c) return c->err.c_str();
  std::lock_guard<std::mutex> l(g_err_mu);
  static thread_local std::string copy;
  copy = g_err;
  return copy.c_str();
}

extern "C" int kp_device_info(const kp_ctx* c, char* name, int name_len, int* num_cu, int64_t* hbm) {
  if (!c) return KP_ERR_ARG;
  if (name && name_len > 0) {
    std::strncpy(name, c->name.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (num_cu) *num_cu = c->num_cu;
  if (hbm) *hbm = c->hbm_bytes;
  return KP_OK;
}

void* kp_pinned_scratch(kp_ctx* ctx, size_t bytes) {
  if (ctx->pin_scratch_bytes >= bytes) return ctx->pin_scratch;
  if (ctx->pin_scratch) {
    (void)hipStreamSynchronize(ctx->stream);            // no DMA into the old block may still be in flight
    (void)hipHostFree(ctx->pin_scratch);
  }
  ctx->pin_scratch = nullptr;
  ctx->pin_scratch_bytes = 0;
  const size_t cap = (bytes + 4095) & ~(size_t)4095;
  if (hipHostMalloc(&ctx->pin_scratch, cap, hipHostMallocDefault) != hipSuccess) {
    ctx->pin_scratch = nullptr;
    return nullptr;
  }
  ctx->pin_scratch_bytes = cap;
  return ctx->pin_scratch;
}

extern "C" int kp_timer_get(const kp_ctx* c, int which, double* ms) {
  if (!c || !ms || which < 0 || which >= 12) return KP_ERR_ARG;
  *ms = c->timers[which];
  return KP_OK;
}

extern "C" void* kp_stream(const kp_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ------------------------------------------------------------------------------------
// dictionary
// ------------------------------------------------------------------------------------

static int upload(kp_ctx* ctx, void** dst, const void* src, size_t bytes) {
  *dst = nullptr;
  if (bytes == 0) return KP_OK;
  KP_HIP(ctx, hipMalloc(dst, bytes));
  KP_HIP(ctx, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return KP_OK;
}

// the sizes kp_basis_create would give this descriptor, without a device (callers of kp_multi_* size their outputs with it)
extern "C" int kp_basis_desc_dims(const kp_basis_desc* d, int* nvars_out, int* nfull_out, int* N_out, int* W_out) {
  if (!d || d->nzeta < 1 || d->m < 0 || d->model_type < 0 || d->model_type > 2 || d->n_blocks < 0 ||
      (d->n_blocks && (!d->block_type || !d->block_count)))
    return KP_ERR_ARG;
  const int nvars = d->nzeta + (d->model_type == KP_MODEL_NONLINEAR ? d->m : 0);
  double nf = nvars + 1.0;
  for (int b = 0; b < d->n_blocks; ++b) {
    const int c = d->block_count[b];
    if (c < 0) return KP_ERR_ARG;
    switch (d->block_type[b]) {
      case KP_BLOCK_POLY: case KP_BLOCK_HERMITE: case KP_BLOCK_FOURIER_SPARSER: case KP_BLOCK_GAUSSIAN: nf += c; break;
      case KP_BLOCK_FOURIER: nf += std::pow((double)(2 * c + 1), (double)nvars) - 1.0; break;
      default: return KP_ERR_ARG;
    }
    if (nf > 1e6) return KP_ERR_ARG;
  }
  const int nfull = (int)std::llround(nf);
  const int N = d->k_pcs > 0 ? d->k_pcs + nvars + 1 : nfull;
  if (nvars_out) *nvars_out = nvars;
  if (nfull_out) *nfull_out = nfull;
  if (N_out) *N_out = N;
  if (W_out) *W_out = d->model_type == KP_MODEL_BILINEAR ? N * (d->m + 1) : d->model_type == KP_MODEL_LINEAR ? N + d->m : N;
  return KP_OK;
}

extern "C" int kp_basis_create(kp_ctx* ctx, const kp_basis_desc* d, kp_basis** out) {
  if (!ctx || !d || !out) return KP_ERR_ARG;
  *out = nullptr;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (d->model_type < 0 || d->model_type > 2) return ctx->fail(KP_ERR_ARG, "kp_basis_create: model_type");
  if (d->nzeta < 1 || d->m < 0) return ctx->fail(KP_ERR_ARG, "kp_basis_create: nzeta/m");
  int nvars = d->nzeta + (d->model_type == KP_MODEL_NONLINEAR ? d->m : 0);
  if (nvars > KP_MAX_VARS) return ctx->fail(KP_ERR_ARG, "kp_basis_create: too many variables (max 32)");
  std::vector<ColDesc> cols;
  for (int i = 0; i < nvars; ++i) cols.push_back({COL_VAR, i, 0, 0});
  int n_mono = 0, n_gauss = 0, max_deg = 1, fourier_deg = 0;
  for (int b = 0; b < d->n_blocks; ++b) {
    int cnt = d->block_count[b];
    if (cnt < 0) return ctx->fail(KP_ERR_ARG, "kp_basis_create: negative block count");
    switch (d->block_type[b]) {
      case KP_BLOCK_POLY:
        if (cnt && !d->poly_exps) return ctx->fail(KP_ERR_ARG, "kp_basis_create: poly_exps is NULL");
        for (int i = 0; i < cnt; ++i) {
          int deg = 0;
          for (int v = 0; v < nvars; ++v) deg += d->poly_exps[(size_t)(n_mono + i) * nvars + v];
          if (deg > max_deg) max_deg = deg;
          // up to 8 variables: the exponent bytes ride in the descriptor itself (aux = variables 0-3, pad = 4-7), so the
          // fused lift of an MPC step needs one load per column instead of two dependent ones
          uint32_t pk[2] = {0, 0};
          if (nvars <= 8)
            for (int v = 0; v < nvars; ++v) pk[v >> 2] |= (uint32_t)d->poly_exps[(size_t)(n_mono + i) * nvars + v] << (8 * (v & 3));
          cols.push_back({COL_MONO, n_mono + i, (int32_t)pk[0], (int32_t)pk[1]});
        }
        n_mono += cnt;
        break;
      case KP_BLOCK_FOURIER: {
        if (cnt < 1) return ctx->fail(KP_ERR_ARG, "kp_basis_create: fourier degree < 1");
        double nf = std::pow((double)(2 * cnt + 1), (double)nvars);
        if (nf > 1e6) return ctx->fail(KP_ERR_ARG, "kp_basis_create: fourier block too large");
        int total = (int)std::llround(nf);
        for (int i = 1; i < total; ++i) {
          // the digits of the mixed-radix index (digit of variable v = harmonic of x_v: 0 none, 2j - 1 cos, 2j sin) packed four
          // bits each into `pad` when they fit: the lift kernel then shifts instead of dividing - twelve 32-bit integer
          // divisions per function and point were ~200 of the ~300 instructions of a fourier value
          uint32_t pk = 0;
          if (nvars <= 8 && 2 * cnt + 1 <= 16) {
            int idx = i;
            for (int v = nvars - 1; v >= 0; --v) {
              pk |= (uint32_t)(idx % (2 * cnt + 1)) << (4 * v);
              idx /= 2 * cnt + 1;
            }
          }
          cols.push_back({COL_FOURIER, i, cnt, (int32_t)pk});
        }
        fourier_deg = fourier_deg == 0 ? cnt : (fourier_deg == cnt ? cnt : -1);   // (blocks of different degrees: generic evaluation)
        break;
      }
      case KP_BLOCK_HERMITE:
        if (cnt && !d->poly_exps) return ctx->fail(KP_ERR_ARG, "kp_basis_create: poly_exps is NULL");
        for (int i = 0; i < cnt; ++i) cols.push_back({COL_HERMITE, n_mono + i, 0, 0});
        n_mono += cnt;
        break;
      case KP_BLOCK_FOURIER_SPARSER:
        if (cnt && !d->poly_exps) return ctx->fail(KP_ERR_ARG, "kp_basis_create: poly_exps is NULL");
        for (int i = 0; i < cnt; ++i) cols.push_back({COL_FSPARSE, n_mono + 2 * i, 0, 0});
        n_mono += 2 * cnt;   // two table rows (sine, cosine multipliers) per function
        break;
      case KP_BLOCK_GAUSSIAN:
        if (cnt && !d->gauss_centres) return ctx->fail(KP_ERR_ARG, "kp_basis_create: gauss_centres is NULL");
        for (int i = 0; i < cnt; ++i) cols.push_back({COL_GAUSS, n_gauss + i, 0, 0});
        n_gauss += cnt;
        break;
      default:
        return ctx->fail(KP_ERR_ARG, "kp_basis_create: unknown block type");
    }
  }
  cols.push_back({COL_CONST, 0, 0, 0});
  kp_basis* b = new kp_basis();
  b->ctx = ctx;
  b->max_degree = max_deg;
  b->fourier_degree = (fourier_deg > 0 && fourier_deg <= 8) ? fourier_deg : 0;
  b->pure_fourier = b->fourier_degree > 0 && (int)cols.size() > nvars;
  for (size_t c = (size_t)nvars; c < cols.size() && b->pure_fourier; ++c)
    b->pure_fourier = cols[c].kind == COL_CONST || (cols[c].kind == COL_FOURIER && cols[c].pad != 0 && cols[c].aux == b->fourier_degree);
  // recipes of the fused Gram kernel's fast lift: column = product of <= 4 entries x_v^e of a
  // power table, id = v*D + (e-1), 255 = the constant 1
  {
    int D = 1;
    for (size_t i = 0; i < (size_t)n_mono * nvars; ++i) D = std::max(D, (int)d->poly_exps[i]);
    bool fast = (nvars * D <= 254);
    int max_nf = 1;
    std::vector<uint32_t> rec(cols.size(), 0xffffffffu);
    for (size_t c = 0; c < cols.size() && fast; ++c) {
      uint32_t r = 0xffffffffu;
      int nf = 0;
      if (cols[c].kind == COL_VAR) {
        r = (r & ~0xffu) | (uint32_t)(cols[c].arg * D);
      } else if (cols[c].kind == COL_MONO) {
        const uint8_t* e = d->poly_exps + (size_t)cols[c].arg * nvars;
        for (int v = 0; v < nvars; ++v)
          if (e[v]) {
            if (nf == 4) { fast = false; break; }
            if (nf + 1 > max_nf) max_nf = nf + 1;
            r = (r & ~(0xffu << (8 * nf))) | ((uint32_t)(v * D + e[v] - 1) << (8 * nf));
            ++nf;
          }
      } else if (cols[c].kind != COL_CONST) {
        fast = false;
      }
      rec[c] = r;
    }
    b->fast = fast;
    b->max_factors = max_nf;
    b->pow_depth = D;
    if (fast) {
      b->h_recipes = rec;
      int rc0 = upload(ctx, &b->d_recipes, rec.data(), rec.size() * 4);
      if (rc0) { delete b; return rc0; }
    }
  }
  // extended recipes (kp_gram3): monomials, fourier products and gaussians as products of <= 3 table entries
  {
    int Dp = 1, df = 0;
    for (size_t i = 0; i < (size_t)n_mono * nvars; ++i) Dp = std::max(Dp, (int)d->poly_exps[i]);
    for (const ColDesc& c : cols)
      if (c.kind == COL_FOURIER) df = std::max(df, c.aux);
    const int D = Dp + 2 * df;
    bool ok = nvars * D < 128 && n_gauss <= 64;
    int max_nf = 1;
    std::vector<uint32_t> rec(cols.size(), 0xffffffffu);
    for (size_t c = 0; c < cols.size() && ok; ++c) {
      uint32_t r = 0xffffffffu;
      int nf = 0;
      auto push = [&](int id) {
        if (nf == 3) { ok = false; return; }
        r = (r & ~(0xffu << (8 * nf))) | ((uint32_t)id << (8 * nf));
        ++nf;
        max_nf = std::max(max_nf, nf);
      };
      switch (cols[c].kind) {
        case COL_VAR: push(cols[c].arg * D); break;
        case COL_MONO: {
          const uint8_t* e = d->poly_exps + (size_t)cols[c].arg * nvars;
          for (int vv = 0; vv < nvars && ok; ++vv)
            if (e[vv]) push(vv * D + e[vv] - 1);
          break;
        }
        case COL_FOURIER: {     // mixed-radix digits, last variable fastest (kp_eval_col): 0 -> 1, 2j-1 -> cos, 2j -> sin(2 pi j x)
          const int radix = 2 * cols[c].aux + 1;
          int idx = cols[c].arg;
          for (int vv = nvars - 1; vv >= 0 && ok; --vv) {
            const int dg = idx % radix;
            idx /= radix;
            if (dg) push(vv * D + Dp + 2 * (((dg + 1) >> 1) - 1) + ((dg & 1) ? 0 : 1));
          }
          break;
        }
        case COL_GAUSS: push(128 + cols[c].arg); break;
        case COL_CONST: break;
        default: ok = false;
      }
      rec[c] = r;
    }
    b->fast_ext = ok;
    b->ext_Dp = Dp;
    b->ext_df = df;
    b->ext_ng = n_gauss;
    b->ext_max_factors = max_nf;
    if (ok) {
      int rc0 = upload(ctx, &b->d_recipes_ext, rec.data(), rec.size() * 4);
      if (rc0) { kp_basis_destroy(b); return rc0; }
    }
  }
  BasisDev& v = b->dev;
  v.model_type = d->model_type;
  v.nzeta = d->nzeta;
  v.m = d->m;
  v.nvars = nvars;
  v.nfull = (int)cols.size();
  v.k_pcs = d->k_pcs > 0 ? d->k_pcs : 0;
  if (v.k_pcs && !d->pcs) {
    delete b;
    return ctx->fail(KP_ERR_ARG, "kp_basis_create: pcs is NULL");
  }
  // params.N: Ksysid.m:534 without dim_red; :1512-1516 with (nvars already holds nzeta(+m))
  v.N = v.k_pcs ? v.k_pcs + nvars + 1 : v.nfull;
  v.W = d->model_type == KP_MODEL_BILINEAR ? v.N * (d->m + 1) : d->model_type == KP_MODEL_LINEAR ? v.N + d->m : v.N;
  int rc = upload(ctx, &b->d_cols, cols.data(), cols.size() * sizeof(ColDesc));
  if (!rc) rc = upload(ctx, &b->d_exps, d->poly_exps, (size_t)n_mono * nvars);
  if (!rc) rc = upload(ctx, &b->d_centres, d->gauss_centres, (size_t)n_gauss * nvars * sizeof(double));
  if (!rc) rc = upload(ctx, &b->d_pcs, d->pcs, (size_t)v.nfull * v.k_pcs * sizeof(double));
  if (rc) {
    kp_basis_destroy(b);
    return rc;
  }
  v.cols = (const ColDesc*)b->d_cols;
  v.exps = (const uint8_t*)b->d_exps;
  v.centres = (const double*)b->d_centres;
  v.pcs = (const double*)b->d_pcs;
  *out = b;
  return KP_OK;
}

extern "C" int kp_basis_destroy(kp_basis* b) {
  if (!b) return KP_OK;
  (void)hipSetDevice(b->ctx->device);
  if (b->d_cols) (void)hipFree(b->d_cols);
  if (b->d_exps) (void)hipFree(b->d_exps);
  if (b->d_centres) (void)hipFree(b->d_centres);
  if (b->d_pcs) (void)hipFree(b->d_pcs);
  if (b->d_recipes) (void)hipFree(b->d_recipes);
  if (b->d_pcsT) (void)hipFree(b->d_pcsT);
  if (b->d_recipes_ext) (void)hipFree(b->d_recipes_ext);
  kp_gram_plan_free(b->plan);
  kp_gram2_plan_free(b->plan2);
  kp_gram3_plan_free(b->plan3);
  kp_gram5_plan_free(b->plan5);
  kp_gram3_shadow_free(b);
  delete b;
  return KP_OK;
}

extern "C" int kp_basis_dims(const kp_basis* b, int* nvars, int* nfull, int* N, int* W) {
  if (!b) return KP_ERR_ARG;
  if (nvars) *nvars = b->dev.nvars;
  if (nfull) *nfull = b->dev.nfull;
  if (N) *N = b->dev.N;
  if (W) *W = b->dev.W;
  return KP_OK;
}

// ------------------------------------------------------------------------------------
// snapshots
// ------------------------------------------------------------------------------------

// kp_snapshots_upload / kp_snapshots_update / kp_snapshots_destroy: kp_upload.hip (staged, chunked host -> HBM copy)

// ------------------------------------------------------------------------------------
// standalone lift kernel: lift.full / lift.econ_full / Px rows for a batch of points
// One workgroup handles LT points; the full lift goes through LDS when a pcs projection
// follows.  HBM-bound on the output (rows x width x 8 B); outputs are column-major so a
// wave writes 64 consecutive rows of one column (coalesced).
// ------------------------------------------------------------------------------------

// LT points per workgroup (64; 16 when the full lift of a projected dictionary would not fit the LDS at 64).  ldi / ldo: leading
// dimensions of the input columns and of the output (the wide Gram path, kp_wide.hip, lifts a row range of the snapshot arrays
// into a panel of its own).
// fdeg > 0: the dictionary has a fourier block of that degree (def_fourierLift, Ksysid.m:694-731): cos / sin(2 pi j x_v), j = 1..fdeg,
// are formed ONCE per (point, variable) into `trig` and a function is a product of table entries - the generic column
// evaluation called sin / cos once per non-trivial factor of every function, 3 000 of them per point for 728 functions on six
// states (the lift was half of the wide Gram pass at W = 738).
template <int LT>
__device__ __forceinline__ double kp_lift_fourier_col(const BasisDev& b, const ColDesc c, const double* trig, int fdeg, int p) {
  double v = 1.0;
  if (c.pad) {                                       // digits packed by kp_basis_create (index >= 1: never all zero)
    // all (at most eight) table reads requested together, no branch and no select per factor: digit 0 reads the row of ones
    // that heads every variable's harmonics (digits beyond nvars are zero: the last variable's ones)
    const uint32_t pk = (uint32_t)c.pad;
    const int tw = 2 * fdeg + 1, vlast = b.nvars - 1;
    double t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = trig[((i < vlast ? i : vlast) * tw + ((pk >> (4 * i)) & 15)) * LT + p];
#pragma unroll
    for (int i = 7; i >= 0; --i) v *= t[i];
    return v;
  }
  const int radix = 2 * c.aux + 1;
  int idx = c.arg;
  for (int i = b.nvars - 1; i >= 0; --i) {
    const int d = idx % radix;
    idx /= radix;
    v *= trig[(i * (2 * fdeg + 1) + d) * LT + p];               // digit 0: 1, 2j - 1: cos(2 pi j x), 2j: sin(2 pi j x)
  }
  return v;
}

template <int LT>
__global__ __launch_bounds__(256) void kp_lift_kernel(BasisDev b, int what, const double* __restrict__ zeta,
                                                      const double* __restrict__ u, int64_t rows, int64_t ldi, int64_t ldo,
                                                      double* __restrict__ out, int fdeg, int stage_cols) {
  extern __shared__ double sm[];
  // layout: vars[nvars][LT] | um[m][LT] | trig[nvars][1 + 2 fdeg][LT] (fourier blocks: ones, cos, sin, cos 2, ...) | full[nfull][LT] (only when k_pcs) | the column
  // descriptors (stage_cols: a value used to start with a 16-byte load from L2 that nothing could hide - ~600 cycles each)
  double* vars = sm;
  double* um = vars + b.nvars * LT;
  double* trig = um + (b.m > 0 ? b.m : 1) * LT;
  double* full = trig + b.nvars * (fdeg > 0 ? 2 * fdeg + 1 : 0) * LT;
  // (two instantiations of the loops below, one per address space: through a generic pointer the descriptor would be a FLAT load,
  //  whose wait also waits - the vector-memory counter retires in order - for the previous value's store to be acknowledged)
  ColDesc* const lc = reinterpret_cast<ColDesc*>(full + (b.k_pcs ? b.nfull * LT : 0));
  if (stage_cols & 1)
    for (int c = threadIdx.x; c < b.nfull; c += 256) lc[c] = b.cols[c];      // (published by the barrier behind the variables below)
  const int64_t r0 = (int64_t)blockIdx.x * LT;
  const int tid = threadIdx.x;
  const int nl = (int)min((int64_t)LT, rows - r0);
  for (int e = tid; e < (b.nvars + b.m) * LT; e += 256) {
    int v = e / LT, p = e % LT;
    double x = 0.0;
    if (p < nl) {
      if (v < b.nzeta)
        x = zeta[(int64_t)v * ldi + r0 + p];
      else if (v < b.nvars)  // nonlinear: u appended to zeta
        x = u[(int64_t)(v - b.nzeta) * ldi + r0 + p];
      else if (u)
        x = u[(int64_t)(v - b.nvars) * ldi + r0 + p];
    }
    if (v < b.nvars)
      vars[v * LT + p] = x;
    else
      um[(v - b.nvars) * LT + p] = x;
  }
  __syncthreads();
  if (fdeg > 0) {
    for (int e = tid; e < b.nvars * fdeg * LT; e += 256) {
      const int p = e % LT, vj = e / LT, v = vj / fdeg, j = vj % fdeg + 1;
      double sn, cs;
      sincos(2.0 * 3.14159265358979323846 * (double)j * vars[v * LT + p], &sn, &cs);
      trig[(v * (2 * fdeg + 1) + 2 * j - 1) * LT + p] = cs;
      trig[(v * (2 * fdeg + 1) + 2 * j) * LT + p] = sn;
      if (j == 1) trig[v * (2 * fdeg + 1) * LT + p] = 1.0;      // row 0 of a variable: digit 0 = no harmonic of it
    }
    __syncthreads();
  }
  const bool econ = (b.k_pcs > 0) && what != KP_LIFT_FULL;
  // Dictionaries that are the variables + ONE fourier block (the reference's fourier dictionaries): a loop with nothing but the
  // packed digits (4 bytes per function, staged in LDS), the table reads and the product.  The general loop below carries every
  // column kind's code behind a chain of tests - ~1 000 cycles per value, most of them taken branches - and costs the wide
  // Gram pass at W = 738 as much as its products do at the arm data's 11 999 pairs.
  const bool fast_fourier = (stage_cols & 2) && !econ && !(what == KP_LIFT_ROW && b.model_type == KP_MODEL_BILINEAR);
  if (fast_fourier) {
    const uint32_t* lp = reinterpret_cast<const uint32_t*>(lc);       // [c].pad of the staged descriptors
    for (int e = tid; e < b.nfull * LT; e += 256) {
      const int c = e / LT, p = e % LT;
      if (p >= nl) continue;
      double val;
      if (c < b.nvars) {
        val = vars[c * LT + p];
      } else {
        const uint32_t pk = lp[4 * c] == (uint32_t)COL_CONST ? 0u : lp[4 * c + 3];      // (the constant: no digit, product of ones)
        const int tw = 2 * fdeg + 1, vlast = b.nvars - 1;
        double t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = trig[((i < vlast ? i : vlast) * tw + ((pk >> (4 * i)) & 15)) * LT + p];
        val = 1.0;
#pragma unroll
        for (int i = 7; i >= 0; --i) val *= t[i];
      }
      out[(int64_t)c * ldo + r0 + p] = val;
    }
  }
  auto eval_full = [&](auto colsp) {
    if (fast_fourier) return;
    if (!econ) {
      // column c of the full basis is also column c of psi
      for (int e = tid; e < b.nfull * LT; e += 256) {
        int c = e / LT, p = e % LT;
        if (p >= nl) continue;
        const ColDesc cd = colsp[c];
        double val = (fdeg > 0 && cd.kind == COL_FOURIER && cd.aux == fdeg) ? kp_lift_fourier_col<LT>(b, cd, trig, fdeg, p) : kp_eval_col(b, cd, vars + p, LT);
        int64_t r = r0 + p;
        if (what == KP_LIFT_ROW && b.model_type == KP_MODEL_BILINEAR) {
          out[(int64_t)c * ldo + r] = val;
          for (int i = 0; i < b.m; ++i) out[(int64_t)((i + 1) * b.N + c) * ldo + r] = val * um[i * LT + p];
        } else {
          out[(int64_t)c * ldo + r] = val;
        }
      }
    } else {
      for (int e = tid; e < b.nfull * LT; e += 256) {
        int c = e / LT, p = e % LT;
        const ColDesc cd = colsp[c];
        full[c * LT + p] = p >= nl ? 0.0
                           : (fdeg > 0 && cd.kind == COL_FOURIER && cd.aux == fdeg) ? kp_lift_fourier_col<LT>(b, cd, trig, fdeg, p) : kp_eval_col(b, cd, vars + p, LT);
      }
    }
  };
  if (stage_cols & 1) eval_full(lc);
  else eval_full(b.cols);
  if (econ) {
    __syncthreads();
    // econ = [ v ; pcs' * full ; 1 ]   (Ksysid.m:1615-1618)
    for (int e = tid; e < b.N * LT; e += 256) {
      int c = e / LT, p = e % LT;
      if (p >= nl) continue;
      double val;
      if (c < b.nvars)
        val = vars[c * LT + p];
      else if (c < b.nvars + b.k_pcs) {
        const double* pc = b.pcs + (size_t)(c - b.nvars) * b.nfull;
        val = 0.0;
        for (int i = 0; i < b.nfull; ++i) val += pc[i] * full[i * LT + p];
      } else
        val = 1.0;
      int64_t r = r0 + p;
      out[(int64_t)c * ldo + r] = val;
      if (what == KP_LIFT_ROW && b.model_type == KP_MODEL_BILINEAR)
        for (int i = 0; i < b.m; ++i) out[(int64_t)((i + 1) * b.N + c) * ldo + r] = val * um[i * LT + p];
    }
  }
  if (what == KP_LIFT_ROW && b.model_type == KP_MODEL_LINEAR) {  // [psi , u]  Ksysid.m:1062
    for (int e = tid; e < b.m * LT; e += 256) {
      int i = e / LT, p = e % LT;
      if (p < nl) out[(int64_t)(b.N + i) * ldo + r0 + p] = um[i * LT + p];
    }
  }
}

// device-to-device lift (used by kp_lift, kp_fit_refine and the wide Gram path): zeta, u, out are device pointers; ldi / ldo as
// in the kernel
int kp_lift_dev_ld(kp_ctx* ctx, const kp_basis* basis, int what, const double* dz, const double* du, int64_t rows, int64_t ldi, double* dout,
                   int64_t ldo) {
  const BasisDev& b = basis->dev;
  const int fdeg = basis->fourier_degree;
  const size_t per_point = (size_t)(b.nvars + (b.m > 0 ? b.m : 1) + b.nvars * (fdeg > 0 ? 2 * fdeg + 1 : 0) + (b.k_pcs ? b.nfull : 0)) * 8;
  // 64 points per workgroup when that still gives every CU several workgroups (the kernel is bound by instruction issue: one
  // workgroup per CU is one wave per SIMD, a quarter of what a SIMD can issue), else 16
  const int ncu_ = ctx->num_cu > 0 ? ctx->num_cu : 256;
  const int lt = (per_point * 64 <= 160 * 1024 && rows >= (int64_t)64 * 4 * ncu_) ? 64 : 16;
  size_t lds = per_point * lt;
  if (lds > 160 * 1024) return ctx->fail(KP_ERR_ARG, "kp_lift: dictionary too large for the LDS staging of the pcs projection");
  int stage_cols = lds + (size_t)b.nfull * sizeof(ColDesc) <= 160 * 1024 ? 1 : 0;
  if (stage_cols) lds += (size_t)b.nfull * sizeof(ColDesc);
  if (stage_cols && basis->pure_fourier && fdeg > 0 && b.nvars <= 8) stage_cols |= 2;      // bit 1: the fourier-only loop of the kernel
  static KpLdsCache c64, c16;
  const int64_t nblk = (rows + lt - 1) / lt;
  if (lt == 64) {
    KP_HIP(ctx, kp_ensure_lds(c64, (const void*)kp_lift_kernel<64>, lds));
    hipLaunchKernelGGL(kp_lift_kernel<64>, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, b, what, dz, du, rows, ldi, ldo, dout, fdeg, stage_cols);
  } else {
    KP_HIP(ctx, kp_ensure_lds(c16, (const void*)kp_lift_kernel<16>, lds));
    hipLaunchKernelGGL(kp_lift_kernel<16>, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, b, what, dz, du, rows, ldi, ldo, dout, fdeg, stage_cols);
  }
  KP_HIP(ctx, hipGetLastError());
  return KP_OK;
}

int kp_lift_dev(kp_ctx* ctx, const kp_basis* basis, int what, const double* dz, const double* du, int64_t rows, double* dout) {
  return kp_lift_dev_ld(ctx, basis, what, dz, du, rows, rows, dout, rows);
}

extern "C" int kp_lift(kp_ctx* ctx, const kp_basis* basis, int what, const double* zeta, const double* u, int64_t rows,
                       double* out) {
  if (!ctx || !basis || what < 0 || what > 2 || rows < 0) return ctx ? ctx->fail(KP_ERR_ARG, "kp_lift: bad argument") : KP_ERR_ARG;
  if (rows == 0) return KP_OK;
  const BasisDev& b = basis->dev;
  if (!zeta || !out) return ctx->fail(KP_ERR_ARG, "kp_lift: NULL pointer");
  bool need_u = b.model_type == KP_MODEL_NONLINEAR || (what == KP_LIFT_ROW && b.m > 0);
  if (need_u && !u) return ctx->fail(KP_ERR_ARG, "kp_lift: u required");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  int width = what == KP_LIFT_FULL ? b.nfull : what == KP_LIFT_ECON ? b.N : b.W;
  size_t bz = (size_t)rows * b.nzeta * 8, bu = (size_t)rows * b.m * 8, bo = (size_t)rows * width * 8;
  double* dz = (double*)ctx->workspace(0, bz);
  double* du = (double*)ctx->workspace(1, bu ? bu : 8);
  double* dout = (double*)ctx->workspace(2, bo);
  if (!dz || !du || !dout) return ctx->fail(KP_ERR_HIP, "kp_lift: out of device memory");
  KP_HIP(ctx, hipMemcpyAsync(dz, zeta, bz, hipMemcpyHostToDevice, ctx->stream));
  if (u && bu) KP_HIP(ctx, hipMemcpyAsync(du, u, bu, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = kp_lift_dev_ld(ctx, basis, what, dz, (u && bu) ? du : nullptr, rows, rows, dout, rows);
  if (rc) return rc;
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(out, dout, bo, hipMemcpyDeviceToHost, ctx->stream));
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[4] = ms;
  return KP_OK;
}

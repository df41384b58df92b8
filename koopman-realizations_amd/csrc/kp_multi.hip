// kp_multi_*: ONE caller, several GPUs.
//
// The reference's host is a single MATLAB interpreter that runs its sweeps as serial loops over independent units - lasso
// values (Ksysid.train_models, Ksysid.m:1372-1387), random systems x model types x degrees (evaluate_rand_models.m:45-144),
// MPC problems - and has no parallel construct.  The kp_comm_* entry points serve a launch with one PROCESS per GPU; these
// serve the reference's own shape: one process, one thread calling in.  The library owns one context and one worker thread
// per listed device; a call deals its units over the workers, every worker drives its own GPU on its own HIP stream, and
// each device writes ITS share of the result straight into the caller's arrays (direct DMA when they are page-locked -
// kp_multi_host_alloc -, staged through the context's pinned scratch otherwise).  No collective on the data path; the one
// exchange step - a single fit sharded over SNAPSHOTS - moves every device's [G | C] to device 0 by a peer copy over xGMI
// and sums them there in a fixed order (bitwise reproducible for a given device list).
// The same device may be listed more than once (two contexts, two workers): how the fan-out, the packing of ragged shards
// and the exchange are tested on a one-GPU box.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include "kp_internal.h"

namespace {

struct Worker {
  int index = 0, device = 0;
  kp_ctx* ctx = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int rc = KP_OK;
  // the dictionary and the snapshot object of the most recent call stay on the device: a sweep calls again with the same ones
  std::vector<uint8_t> desc_key;
  kp_basis* basis = nullptr;
  kp_snapshots* snaps = nullptr;
  double ms[4] = {0, 0, 0, 0};        // upload, device work, result transfer, whole job (wall, this worker)
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void worker_main(Worker* w) {
  (void)hipSetDevice(w->device);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    w->has_job = false;
    lk.unlock();
    const double t0 = now_ms();
    int rc = KP_ERR_HIP;
    try {
      rc = job();
    } catch (const std::exception& e) {
      w->ctx->fail(KP_ERR_HIP, std::string("kp_multi worker: ") + e.what());
    }
    w->ms[3] = now_ms() - t0;
    lk.lock();
    w->rc = rc;
    w->done = true;
    w->cv.notify_all();
  }
}

// bytes that identify a dictionary descriptor (a sweep hands the same one to call after call)
int desc_counts(const kp_basis_desc* d, size_t* n_bytes, size_t* n_centres, int* nfull) {
  if (!d || d->nzeta < 1 || d->m < 0 || d->model_type < 0 || d->model_type > 2 || d->n_blocks < 0 || (d->n_blocks && (!d->block_type || !d->block_count)))
    return KP_ERR_ARG;
  const int nvars = d->nzeta + (d->model_type == KP_MODEL_NONLINEAR ? d->m : 0);
  size_t rows = 0, cen = 0;
  double nf = nvars + 1.0;
  for (int b = 0; b < d->n_blocks; ++b) {
    const int c = d->block_count[b];
    if (c < 0) return KP_ERR_ARG;
    switch (d->block_type[b]) {
      case KP_BLOCK_POLY: case KP_BLOCK_HERMITE: rows += (size_t)c; nf += c; break;
      case KP_BLOCK_FOURIER_SPARSER: rows += 2 * (size_t)c; nf += c; break;
      case KP_BLOCK_GAUSSIAN: cen += (size_t)c; nf += c; break;
      case KP_BLOCK_FOURIER: { double t = 1.0; for (int v = 0; v < nvars; ++v) t *= 2.0 * c + 1.0; nf += t - 1.0; break; }
      default: return KP_ERR_ARG;
    }
  }
  if (nf > 1e6) return KP_ERR_ARG;
  *n_bytes = rows * (size_t)nvars;
  *n_centres = cen * (size_t)nvars;
  *nfull = (int)nf;
  return KP_OK;
}

int desc_key(const kp_basis_desc* d, std::vector<uint8_t>* key) {
  size_t nb = 0, nc = 0;
  int nfull = 0;
  int rc = desc_counts(d, &nb, &nc, &nfull);
  if (rc) return rc;
  if ((nb && !d->poly_exps) || (nc && !d->gauss_centres) || (d->k_pcs > 0 && !d->pcs)) return KP_ERR_ARG;
  key->clear();
  auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; key->insert(key->end(), b, b + n); };
  const int32_t head[5] = {d->model_type, d->nzeta, d->m, d->n_blocks, d->k_pcs > 0 ? d->k_pcs : 0};
  put(head, sizeof head);
  put(d->block_type, (size_t)d->n_blocks * 4);
  put(d->block_count, (size_t)d->n_blocks * 4);
  put(d->poly_exps, nb);
  put(d->gauss_centres, nc * 8);
  if (d->k_pcs > 0) put(d->pcs, (size_t)nfull * d->k_pcs * 8);
  return KP_OK;
}

}  // namespace

struct kp_multi {
  std::vector<std::unique_ptr<Worker>> w;
  mutable std::string err;
  std::vector<std::pair<char*, size_t>> host_blocks;     // kp_multi_host_alloc: page-locked, visible to every device
  std::mutex mu;                                         // one call at a time
  int fail(int code, const std::string& s) const {
    err = s;
    kp_set_global_error(s);
    return code;
  }
  bool pinned(const void* p, size_t bytes) const {
    for (auto& b : host_blocks)
      if ((const char*)p >= b.first && (const char*)p + bytes <= b.first + b.second) return true;
    return false;
  }
};
struct kp_multi_traj {
  kp_multi* mg = nullptr;
  int nb = 0, n = 0, m = 0;
  std::vector<kp_traj*> shard;        // per worker (nullptr: no systems)
  std::vector<int> first, count;
};
struct kp_multi_mpc {
  kp_multi* mg = nullptr;
  int N = 0, m = 0, Np = 0, nproj = 0, nvar = 0;
  std::vector<kp_mpc*> mpc;
};

namespace {

// jobs[r] runs on worker r (an empty function: nothing to do); returns the first failure, its text in mg->err
int run_all(kp_multi* mg, std::vector<std::function<int()>>& jobs) {
  const int n = (int)mg->w.size();
  for (int r = 0; r < n; ++r) {
    Worker* w = mg->w[r].get();
    if (!jobs[r]) continue;
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = std::move(jobs[r]);
    w->has_job = true;
    w->done = false;
    w->rc = KP_OK;
    w->cv.notify_all();
  }
  int rc = KP_OK;
  for (int r = 0; r < n; ++r) {
    Worker* w = mg->w[r].get();
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc != KP_OK && rc == KP_OK) {
      rc = w->rc;
      mg->fail(rc, "device " + std::to_string(w->device) + " (worker " + std::to_string(r) + "): " + kp_last_error(w->ctx));
    }
  }
  return rc;
}

// the worker's dictionary for this descriptor (kept from the previous call when the descriptor is the same)
int worker_basis(Worker* w, const kp_basis_desc* d, const std::vector<uint8_t>& key) {
  if (w->basis && w->desc_key == key) return KP_OK;
  if (w->basis) kp_basis_destroy(w->basis);
  w->basis = nullptr;
  w->desc_key.clear();
  int rc = kp_basis_create(w->ctx, d, &w->basis);
  if (rc) return rc;
  w->desc_key = key;
  return KP_OK;
}

// the worker's resident snapshot object, refilled with rows [r0, r0 + nr) of the caller's arrays (ld rows each)
int worker_snapshots(Worker* w, const double* alpha, const double* beta, const double* u, int64_t ld, int64_t r0, int64_t nr, int nzeta, int m) {
  if (w->snaps && (w->snaps->nzeta != nzeta || w->snaps->m != m)) {
    kp_snapshots_destroy(w->snaps);
    w->snaps = nullptr;
  }
  if (!w->snaps) {
    int rc = kp_snapshots_upload(w->ctx, nullptr, nullptr, nullptr, 0, nzeta, m, &w->snaps);
    if (rc) return rc;
  }
  return kp_snapshots_update_rows(w->ctx, w->snaps, alpha + r0, beta + r0, m > 0 ? u + r0 : nullptr, nr, ld);
}

__global__ __launch_bounds__(256) void kp_multi_sum_kernel(const double* __restrict__ slots, int n_slots, size_t count, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = slots[i];
  for (int k = 1; k < n_slots; ++k) s += slots[(size_t)k * count + i];   // fixed order: reproducible for a given device list
  out[i] = s;
}

}  // namespace

extern "C" int kp_multi_create(const int* device_ids, int n_dev, kp_multi** out) {
  if (!device_ids || n_dev < 1 || n_dev > 64 || !out) {
    kp_set_global_error("kp_multi_create: bad argument");
    return KP_ERR_ARG;
  }
  *out = nullptr;
  std::unique_ptr<kp_multi> mg(new kp_multi());
  for (int r = 0; r < n_dev; ++r) {
    std::unique_ptr<Worker> w(new Worker());
    w->index = r;
    w->device = device_ids[r];
    int rc = kp_create(w->device, &w->ctx);
    if (rc) {
      for (auto& q : mg->w) kp_destroy(q->ctx);
      return rc;
    }
    mg->w.push_back(std::move(w));
  }
  // peer access between the devices (the exchange of kp_multi_fit_sharded is a peer copy); refusals are not errors: the
  // copy then goes through the host
  int dev_before = -1;                                    // the caller's current device is left as it was
  if (hipGetDevice(&dev_before) != hipSuccess) dev_before = -1;
  for (int a = 0; a < n_dev; ++a)
    for (int b = 0; b < n_dev; ++b) {
      const int da = device_ids[a], db = device_ids[b];
      int can = 0;
      if (da == db || hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) continue;
      if (hipSetDevice(da) == hipSuccess) (void)hipDeviceEnablePeerAccess(db, 0);
      (void)hipGetLastError();
    }
  if (dev_before >= 0) (void)hipSetDevice(dev_before);
  for (auto& w : mg->w) w->th = std::thread(worker_main, w.get());
  *out = mg.release();
  return KP_OK;
}

extern "C" int kp_multi_destroy(kp_multi* mg) {
  if (!mg) return KP_OK;
  for (auto& w : mg->w) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->quit = true;
      w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();
    if (w->snaps) kp_snapshots_destroy(w->snaps);
    if (w->basis) kp_basis_destroy(w->basis);
  }
  for (auto& b : mg->host_blocks) (void)hipHostFree(b.first);
  for (auto& w : mg->w) kp_destroy(w->ctx);
  delete mg;
  return KP_OK;
}

extern "C" int kp_multi_size(const kp_multi* mg, int* n_dev) {
  if (!mg || !n_dev) return KP_ERR_ARG;
  *n_dev = (int)mg->w.size();
  return KP_OK;
}

extern "C" kp_ctx* kp_multi_ctx(kp_multi* mg, int i) {
  if (!mg || i < 0 || i >= (int)mg->w.size()) return nullptr;
  return mg->w[i]->ctx;
}

extern "C" const char* kp_multi_last_error(const kp_multi* mg) { return mg ? mg->err.c_str() : kp_last_error(nullptr); }

extern "C" int kp_multi_host_alloc(kp_multi* mg, int64_t bytes, void** ptr) {
  if (!mg || !ptr || bytes < 1) return mg ? mg->fail(KP_ERR_ARG, "kp_multi_host_alloc: bad argument") : KP_ERR_ARG;
  *ptr = nullptr;
  void* p = nullptr;
  int dev_before = -1;                                    // the caller's current device is left as it was
  if (hipGetDevice(&dev_before) != hipSuccess) dev_before = -1;
  (void)hipSetDevice(mg->w[0]->device);
  hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable);      // page-locked for EVERY device of the process
  if (dev_before >= 0) (void)hipSetDevice(dev_before);
  if (e != hipSuccess) return mg->fail(KP_ERR_HIP, std::string("kp_multi_host_alloc: ") + hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(mg->mu);
  mg->host_blocks.push_back({(char*)p, (size_t)bytes});
  *ptr = p;
  return KP_OK;
}

extern "C" int kp_multi_host_free(kp_multi* mg, void* ptr) {
  if (!mg) return KP_ERR_ARG;
  if (!ptr) return KP_OK;
  std::lock_guard<std::mutex> lk(mg->mu);
  for (size_t i = 0; i < mg->host_blocks.size(); ++i)
    if (mg->host_blocks[i].first == (char*)ptr) {
      mg->host_blocks.erase(mg->host_blocks.begin() + (long)i);
      (void)hipHostFree(ptr);      // every multi call has synchronised its streams before returning: nothing is in flight
      return KP_OK;
    }
  return mg->fail(KP_ERR_ARG, "kp_multi_host_free: not a block of this object");
}

extern "C" int kp_multi_timers(const kp_multi* mg, double* ms) {
  if (!mg || !ms) return KP_ERR_ARG;
  for (size_t r = 0; r < mg->w.size(); ++r)
    for (int k = 0; k < 4; ++k) ms[r * 4 + k] = mg->w[r]->ms[k];
  return KP_OK;
}

// K matrices `count` of the worker's result buffer (slots 0..count-1) to their places idx[j] of the caller's stack
static int scatter_results(kp_multi* mg, Worker* w, int W, const std::vector<int>& idx, double* K_out) {
  kp_ctx* ctx = w->ctx;
  const size_t kb = (size_t)W * W * 8;
  const int cnt = (int)idx.size();
  bool direct = true;
  for (int j = 0; j < cnt; ++j) direct &= mg->pinned(K_out + (size_t)idx[j] * W * W, kb);
  if (direct) {                                           // every slice by its own DMA, straight into the caller's block
    for (int j = 0; j < cnt; ++j)
      KP_HIP(ctx, hipMemcpyAsync(K_out + (size_t)idx[j] * W * W, ctx->Kres + (size_t)j * W * W, kb, hipMemcpyDeviceToHost, ctx->stream));
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KP_OK;
  }
  // pageable destination: one DMA into the context's page-locked scratch (chunks of <= 64 MB), memcpy from there
  const int per = std::max(1, (int)(((size_t)64 << 20) / kb));
  for (int j0 = 0; j0 < cnt; j0 += per) {
    const int c = std::min(per, cnt - j0);
    double* pin = (double*)kp_pinned_scratch(ctx, (size_t)c * kb);
    if (!pin) return ctx->fail(KP_ERR_HIP, "kp_multi_fit: no page-locked scratch");
    KP_HIP(ctx, hipMemcpyAsync(pin, ctx->Kres + (size_t)j0 * W * W, (size_t)c * kb, hipMemcpyDeviceToHost, ctx->stream));
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < c; ++j) std::memcpy(K_out + (size_t)idx[j0 + j] * W * W, pin + (size_t)j * W * W, kb);
  }
  return KP_OK;
}

// The lasso grid of train_models (Ksysid.m:1372-1387): value i on device i mod n.  The Grams do not depend on the value:
// device 0 uploads the snapshot matrix and runs the fused lift + Gram kernel ONCE, every other device receives [G | C]
// (2 W^2 doubles, 1.8 MB at W = 336) by a peer copy over xGMI and only factorises and solves its own values.  (Round 4 uploaded
// the 12 MB of snapshots to EVERY device and repeated the 0.40 ms Gram kernel there: 0.7 ms per device that bought nothing.)
extern "C" int kp_multi_fit(kp_multi* mg, const kp_basis_desc* desc, const double* alpha, const double* beta, const double* u, int64_t Ns,
                            const double* lasso, int n_lasso, double* K_out) {
  if (!mg || !desc || !alpha || !beta || Ns < 1 || n_lasso < 1 || !lasso || !K_out || (desc->m > 0 && !u))
    return mg ? mg->fail(KP_ERR_ARG, "kp_multi_fit: bad argument") : KP_ERR_ARG;
  std::lock_guard<std::mutex> call(mg->mu);
  mg->err.clear();
  const double t_call = now_ms();
  std::vector<uint8_t> key;
  if (desc_key(desc, &key)) return mg->fail(KP_ERR_ARG, "kp_multi_fit: bad dictionary descriptor");
  const int n = (int)mg->w.size();
  const int nw = std::min(n, n_lasso);                      // workers with at least one value
  Worker* w0 = mg->w[0].get();
  // phase 0: every worker's dictionary and [G | C] buffer; worker 0 also uploads the snapshots and forms the Grams
  std::vector<std::function<int()>> jobs(n);
  for (int r = 0; r < nw; ++r) {
    Worker* w = mg->w[r].get();
    jobs[r] = [=, &key]() -> int {
      w->ms[0] = w->ms[1] = w->ms[2] = 0.0;
      int rc = worker_basis(w, desc, key);
      if (rc) return rc;
      if (w->ctx->async_pending) { rc = kp_synchronize(w->ctx); if (rc) return rc; }
      rc = kp_ensure_gc(w->ctx, w->basis->dev.W);
      if (rc || r != 0) return rc;
      double t0 = now_ms();
      rc = worker_snapshots(w, alpha, beta, u, Ns, 0, Ns, desc->nzeta, desc->m);
      if (rc) return rc;
      w->ms[0] = now_ms() - t0;
      t0 = now_ms();
      rc = kp_fit_gram(w->ctx, w->basis, w->snaps, nullptr, nullptr);       // leaves [G | C] in the context's buffer
      w->ms[1] = now_ms() - t0;
      return rc;
    };
  }
  int rc = run_all(mg, jobs);
  if (rc) return rc;
  const int W = w0->basis->dev.W;
  const size_t cnt = (size_t)2 * W * W;
  // phase 1: [G | C] to the peers (issued by worker 0 on its stream; the same device listed twice: a device-to-device copy)
  if (nw > 1) {
    std::vector<std::function<int()>> j1(n);
    j1[0] = [&]() -> int {
      const double t0 = now_ms();
      for (int r = 1; r < nw; ++r) {
        Worker* w = mg->w[r].get();
        KP_HIP(w0->ctx, hipMemcpyPeerAsync(w->ctx->GC, w->device, w0->ctx->GC, w0->device, cnt * 8, w0->ctx->stream));
      }
      KP_HIP(w0->ctx, hipStreamSynchronize(w0->ctx->stream));
      w0->ms[2] = now_ms() - t0;
      return KP_OK;
    };
    rc = run_all(mg, j1);
    if (rc) return rc;
  }
  // phase 2: every worker factorises once and solves its values as one batch, its results straight into the caller's stack
  std::vector<int> ranks(nw, W);
  for (int r = 0; r < nw; ++r) {
    Worker* w = mg->w[r].get();
    jobs[r] = [=, &ranks]() -> int {
      std::vector<int> idx;                       // value i belongs to worker i mod n (as the one-process-per-GPU path deals them)
      std::vector<double> lv;
      for (int i = r; i < n_lasso; i += n) { idx.push_back(i); lv.push_back(lasso[i]); }
      double t0 = now_ms();
      std::vector<double> one;
      double* direct_out = nullptr;
      if (lv.size() == 1) {                       // kp_fit with K_out == NULL and ONE value is the asynchronous pipeline: keep this synchronous
        one.resize((size_t)W * W);
        direct_out = one.data();
      }
      w->ctx->gc_preloaded = true;                // the context's [G | C] is final: kp_fit skips its Gram launch (and needs no snapshots)
      int rc2 = kp_fit(w->ctx, w->basis, w0->snaps, lv.data(), (int)lv.size(), direct_out);
      w->ctx->gc_preloaded = false;
      if (rc2) return rc2;
      KP_HIP(w->ctx, hipStreamSynchronize(w->ctx->stream));
      ranks[r] = w->ctx->last_rank;
      w->ms[1] += now_ms() - t0;
      t0 = now_ms();
      rc2 = scatter_results(mg, w, W, idx, K_out);
      w->ms[2] += now_ms() - t0;
      return rc2;
    };
  }
  rc = run_all(mg, jobs);
  if (rc) return rc;
  for (int r = 0; r < nw; ++r) mg->w[r]->ms[3] = now_ms() - t_call;      // (the call had three rounds of jobs: the whole of it)
  // MATLAB's `\` warns on a rank-deficient Px and goes on (Ksysid.m:1069); the workers' warnings reach the caller here
  for (int r = 0; r < nw; ++r)
    if (ranks[r] >= 0 && ranks[r] < W) {
      mg->err = "warning: Gram matrix is rank deficient (rank " + std::to_string(ranks[r]) + " of " + std::to_string(W) + "); basic solution returned";
      break;
    }
  return KP_OK;
}

extern "C" int kp_multi_fit_sharded(kp_multi* mg, const kp_basis_desc* desc, const double* alpha, const double* beta, const double* u,
                                    int64_t Ns, const double* lasso, int n_lasso, double* K_out) {
  if (!mg || !desc || !alpha || !beta || Ns < 1 || n_lasso < 1 || !K_out || (desc->m > 0 && !u))
    return mg ? mg->fail(KP_ERR_ARG, "kp_multi_fit_sharded: bad argument") : KP_ERR_ARG;
  std::lock_guard<std::mutex> call(mg->mu);
  std::vector<uint8_t> key;
  if (desc_key(desc, &key)) return mg->fail(KP_ERR_ARG, "kp_multi_fit_sharded: bad dictionary descriptor");
  const int n = (int)std::min<int64_t>((int64_t)mg->w.size(), Ns);
  // phase 0: every worker's dictionary (device 0's gives W for the exchange buffer)
  std::vector<std::function<int()>> jobs(mg->w.size());
  for (int r = 0; r < n; ++r) {
    Worker* w = mg->w[r].get();
    jobs[r] = [=, &key]() -> int { return worker_basis(w, desc, key); };
  }
  int rc = run_all(mg, jobs);
  if (rc) return rc;
  Worker* w0 = mg->w[0].get();
  const int W = w0->basis->dev.W;
  const size_t cnt = (size_t)2 * W * W;
  // exchange buffer on device 0: one [G | C] slot per worker (workspace 8 is the collectives' staging slot)
  double* slots = nullptr;
  {
    std::vector<std::function<int()>> j0(mg->w.size());
    j0[0] = [&]() -> int {
      KP_HIP(w0->ctx, hipSetDevice(w0->device));
      if (w0->ctx->async_pending) { int r0 = kp_synchronize(w0->ctx); if (r0) return r0; }
      slots = (double*)w0->ctx->workspace(8, (size_t)n * cnt * 8);
      if (!slots) return w0->ctx->fail(KP_ERR_HIP, "kp_multi_fit_sharded: out of device memory");
      return kp_ensure_gc(w0->ctx, W);
    };
    rc = run_all(mg, j0);
    if (rc) return rc;
  }
  // phase 1: rows [r Ns / n, (r + 1) Ns / n) on worker r; its [G | C] goes to slot r of device 0 by a peer copy
  for (int r = 0; r < n; ++r) {
    Worker* w = mg->w[r].get();
    const int64_t r0 = Ns * r / n, r1 = Ns * (r + 1) / n;
    jobs[r] = [=]() -> int {
      double t0 = now_ms();
      int rc1 = worker_snapshots(w, alpha, beta, u, Ns, r0, r1 - r0, desc->nzeta, desc->m);
      if (rc1) return rc1;
      w->ms[0] = now_ms() - t0;
      t0 = now_ms();
      rc1 = kp_fit_gram(w->ctx, w->basis, w->snaps, nullptr, nullptr);       // leaves [G | C] in the context's buffer
      if (rc1) return rc1;
      w->ms[1] = now_ms() - t0;
      t0 = now_ms();
      KP_HIP(w->ctx, hipMemcpyPeerAsync(slots + (size_t)r * cnt, w0->device, w->ctx->GC, w->device, cnt * 8, w->ctx->stream));
      KP_HIP(w->ctx, hipStreamSynchronize(w->ctx->stream));
      w->ms[2] = now_ms() - t0;
      return KP_OK;
    };
  }
  rc = run_all(mg, jobs);
  if (rc) return rc;
  // phase 2: device 0 sums the slots in worker order and solves
  std::vector<std::function<int()>> j2(mg->w.size());
  j2[0] = [&]() -> int {
    kp_ctx* ctx = w0->ctx;
    hipLaunchKernelGGL(kp_multi_sum_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, slots, n, cnt, ctx->GC);
    KP_HIP(ctx, hipGetLastError());
    ctx->gc_preloaded = true;
    const int rc2 = kp_fit(ctx, w0->basis, w0->snaps, lasso, n_lasso, K_out);      // lasso == NULL: least squares
    ctx->gc_preloaded = false;
    return rc2;
  };
  return run_all(mg, j2);
}

// ---- the random-system sweep (evaluate_rand_models.m:45-144): systems dealt in contiguous chunks ----------------------
extern "C" int kp_multi_traj_upload(kp_multi* mg, const double* Y, const double* U, int nb, int ntrials, int T, int n, int m, const double* Yv,
                                    const double* Uv, int Tv, kp_multi_traj** out) {
  if (!mg || !Y || !U || !Yv || !Uv || !out || nb < 1) return mg ? mg->fail(KP_ERR_ARG, "kp_multi_traj_upload: bad argument") : KP_ERR_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> call(mg->mu);
  const int nw = (int)mg->w.size();
  std::unique_ptr<kp_multi_traj> mt(new kp_multi_traj());
  mt->mg = mg; mt->nb = nb; mt->n = n; mt->m = m;
  mt->shard.assign(nw, nullptr);
  mt->first.assign(nw, 0);
  mt->count.assign(nw, 0);
  const size_t rows = (size_t)ntrials * T;
  std::vector<std::function<int()>> jobs(nw);
  for (int r = 0; r < nw; ++r) {
    const int s0 = (int)((int64_t)nb * r / nw), s1 = (int)((int64_t)nb * (r + 1) / nw);
    mt->first[r] = s0;
    mt->count[r] = s1 - s0;
    if (s1 == s0) continue;
    Worker* w = mg->w[r].get();
    kp_multi_traj* mtp = mt.get();
    jobs[r] = [=]() -> int {
      const double t0 = now_ms();
      const int rc = kp_traj_upload(w->ctx, Y + (size_t)s0 * rows * n, U + (size_t)s0 * rows * m, s1 - s0, ntrials, T, n, m,
                                    Yv + (size_t)s0 * Tv * n, Uv + (size_t)s0 * Tv * m, Tv, &mtp->shard[r]);
      w->ms[0] = now_ms() - t0;
      return rc;
    };
  }
  int rc = run_all(mg, jobs);
  if (rc) {
    for (kp_traj* t : mt->shard) kp_traj_destroy(t);
    return rc;
  }
  *out = mt.release();
  return KP_OK;
}

extern "C" int kp_multi_traj_destroy(kp_multi_traj* mt) {
  if (!mt) return KP_OK;
  for (kp_traj* t : mt->shard) kp_traj_destroy(t);
  delete mt;
  return KP_OK;
}

extern "C" int kp_multi_sweep_eval_nested(kp_multi* mg, const kp_multi_traj* mt, const kp_basis_desc* desc, double lasso, int n_deg,
                                          double* err_out, int* status_out) {
  if (!mg || !mt || mt->mg != mg || !desc || n_deg < 1 || !err_out) return mg ? mg->fail(KP_ERR_ARG, "kp_multi_sweep_eval_nested: bad argument") : KP_ERR_ARG;
  std::lock_guard<std::mutex> call(mg->mu);
  std::vector<uint8_t> key;
  if (desc_key(desc, &key)) return mg->fail(KP_ERR_ARG, "kp_multi_sweep_eval_nested: bad dictionary descriptor");
  const int nw = (int)mg->w.size(), nb = mt->nb, n = mt->n;
  std::vector<std::function<int()>> jobs(nw);
  for (int r = 0; r < nw; ++r) {
    if (!mt->shard[r]) continue;
    Worker* w = mg->w[r].get();
    const int s0 = mt->first[r], c = mt->count[r];
    const kp_traj* tr = mt->shard[r];
    jobs[r] = [=, &key]() -> int {
      int rc = worker_basis(w, desc, key);
      if (rc) return rc;
      const double t0 = now_ms();
      std::vector<double> e((size_t)n_deg * c * n);
      std::vector<int> st((size_t)n_deg * c);
      rc = kp_sweep_eval_nested(w->ctx, tr, w->basis, lasso, n_deg, e.data(), st.data());
      if (rc) return rc;
      w->ms[1] = now_ms() - t0;
      for (int d = 0; d < n_deg; ++d) {           // degree-major tables: this shard's systems are a contiguous run of every degree's block
        std::memcpy(err_out + ((size_t)d * nb + s0) * n, e.data() + (size_t)d * c * n, (size_t)c * n * 8);
        if (status_out) std::memcpy(status_out + (size_t)d * nb + s0, st.data() + (size_t)d * c, (size_t)c * 4);
      }
      return KP_OK;
    };
  }
  return run_all(mg, jobs);
}

// ---- batched MPC (Monte-Carlo closed loops, random-state sweeps): problems dealt in contiguous chunks --------------------
extern "C" int kp_multi_mpc_create(kp_multi* mg, int model_type, const double* A, const double* B, int N, int m, int Np, const double* proj,
                                   int nproj, double q_run, double q_term, const double* r, const double* lo, const double* hi,
                                   double slope_lim, double smooth_lim, kp_multi_mpc** out) {
  if (!mg || !out) return mg ? mg->fail(KP_ERR_ARG, "kp_multi_mpc_create: bad argument") : KP_ERR_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> call(mg->mu);
  const int nw = (int)mg->w.size();
  std::unique_ptr<kp_multi_mpc> mm(new kp_multi_mpc());
  mm->mg = mg; mm->N = N; mm->m = m; mm->Np = Np; mm->nproj = nproj; mm->nvar = m * Np;
  mm->mpc.assign(nw, nullptr);
  kp_multi_mpc* mp = mm.get();
  std::vector<std::function<int()>> jobs(nw);
  for (int k = 0; k < nw; ++k) {
    Worker* w = mg->w[k].get();
    jobs[k] = [=]() -> int {
      return kp_mpc_create(w->ctx, model_type, A, B, N, m, Np, proj, nproj, q_run, q_term, r, lo, hi, slope_lim, smooth_lim, &mp->mpc[k]);
    };
  }
  int rc = run_all(mg, jobs);
  if (rc) {
    for (kp_mpc* p : mm->mpc) kp_mpc_destroy(p);
    return rc;
  }
  *out = mm.release();
  return KP_OK;
}

extern "C" int kp_multi_mpc_set_state_bounds(kp_multi_mpc* mm, int n, const double* lo, const double* hi) {
  if (!mm) return KP_ERR_ARG;
  kp_multi* mg = mm->mg;
  std::lock_guard<std::mutex> call(mg->mu);
  std::vector<std::function<int()>> jobs(mg->w.size());
  for (size_t k = 0; k < mg->w.size(); ++k) {
    kp_mpc* p = mm->mpc[k];
    jobs[k] = [=]() -> int { return kp_mpc_set_state_bounds(p, n, lo, hi); };
  }
  return run_all(mg, jobs);
}

extern "C" int kp_multi_mpc_destroy(kp_multi_mpc* mm) {
  if (!mm) return KP_OK;
  for (kp_mpc* p : mm->mpc) kp_mpc_destroy(p);
  delete mm;
  return KP_OK;
}

extern "C" int kp_multi_mpc_step_batch(kp_multi_mpc* mm, int nb, const double* z, const double* u_prev, const double* Yr, double* U_out,
                                       int* status) {
  if (!mm || nb < 1 || !z || !u_prev || !Yr || !U_out) return mm ? mm->mg->fail(KP_ERR_ARG, "kp_multi_mpc_step_batch: bad argument") : KP_ERR_ARG;
  kp_multi* mg = mm->mg;
  std::lock_guard<std::mutex> call(mg->mu);
  const int nw = (int)mg->w.size();
  const int N = mm->N, m = mm->m, ny = mm->nproj * (mm->Np + 1), nvar = mm->nvar;
  std::vector<std::function<int()>> jobs(nw);
  for (int k = 0; k < nw; ++k) {
    const int p0 = (int)((int64_t)nb * k / nw), p1 = (int)((int64_t)nb * (k + 1) / nw);
    if (p1 == p0) continue;
    Worker* w = mg->w[k].get();
    kp_mpc* p = mm->mpc[k];
    jobs[k] = [=]() -> int {
      const double t0 = now_ms();
      const int rc = kp_mpc_step_batch(p, p1 - p0, z + (size_t)p0 * N, u_prev + (size_t)p0 * m, Yr + (size_t)p0 * ny, U_out + (size_t)p0 * nvar,
                                       status ? status + p0 : nullptr);
      w->ms[1] = now_ms() - t0;
      return rc;
    };
  }
  return run_all(mg, jobs);
}

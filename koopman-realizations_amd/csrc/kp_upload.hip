// Host -> HBM path of the snapshot arrays (`snapshotPairs.alpha / .beta / .u`, Ksysid.m:910-984, handed to
// Ksysid.get_Koopman, Ksysid.m:987).  The caller's arrays are ordinary pageable memory (mxArray data, numpy arrays):
// a plain hipMemcpy of them is bound by ONE host thread copying into the runtime's staging buffer (measured 9 GB/s,
// 1.3 ms for the 12 MB of a 1e5-pair matrix - three times the Gram kernel).  Here the copy is cut into 2 MB chunks
// that are moved into a pinned ring owned by the context - by the calling thread, helped by a few copy threads when the
// matrix is large - and the DMA of every chunk is issued as soon as it is staged (copy stream), so the host copy, the
// PCIe transfer and - for kp_snapshots_update - the Gram kernel of the previous snapshot matrix overlap.
// Measured: 12 MB in 0.43 ms (28 GB/s), 120 MB at 40 GB/s; fits streamed from host memory through two alternating
// objects run at 0.46 ms each (resident data: 0.436).
//
// Ordering of kp_snapshots_update against the kernels that read the object is by events, never by a device
// synchronisation: the DMA waits for the last reader enqueued so far (`ev_read`), later readers wait for the DMA
// (`ev_ready`); two objects filled alternately keep PCIe and the matrix pipes busy at the same time.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "kp_internal.h"

namespace {

struct Chunk {
  char* dst_dev;
  char* stage;
  const char* src;
  size_t bytes;
};

// A handful of sleeping workers; the caller takes part in the copy, so zero workers is a valid configuration.
struct CopyPool {
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv, cv_idle;
  bool stop = false;
  uint64_t gen = 0;
  int active = 0;                       // workers that have not yet left the current job
  const Chunk* chunks = nullptr;
  int n = 0;
  std::atomic<int> next{0};
  std::unique_ptr<std::atomic<int>[]> done;
  int done_cap = 0;

  explicit CopyPool(int nworkers) {
    for (int i = 0; i < nworkers; ++i) workers.emplace_back([this] { run(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> l(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : workers) t.join();
  }
  bool take_one() {
    const int i = next.fetch_add(1, std::memory_order_relaxed);
    if (i >= n) return false;
    std::memcpy(chunks[i].stage, chunks[i].src, chunks[i].bytes);
    done[i].store(1, std::memory_order_release);
    return true;
  }
  void run() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
      }
      while (take_one()) {
      }
      {
        std::lock_guard<std::mutex> l(mu);
        --active;
      }
      cv_idle.notify_one();
    }
  }
  void begin(const Chunk* c, int count) {
    if (count > done_cap) {
      done.reset(new std::atomic<int>[count]);
      done_cap = count;
    }
    for (int i = 0; i < count; ++i) done[i].store(0, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> l(mu);
      chunks = c;
      n = count;
      next.store(0, std::memory_order_relaxed);
      active = (int)workers.size();
      ++gen;
    }
    cv.notify_all();
  }
  // chunk i staged?  The caller copies chunks itself while it waits.
  void wait_chunk(int i) {
    while (!done[i].load(std::memory_order_acquire))
      if (!take_one()) std::this_thread::yield();
  }
  void end() {
    std::unique_lock<std::mutex> l(mu);
    cv_idle.wait(l, [&] { return active == 0; });
  }
};

}  // namespace

struct kp_stage {
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_order = nullptr;
  void* pinned[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  hipEvent_t dma_done[2] = {nullptr, nullptr};
  bool busy[2] = {false, false};
  int next = 0;
  std::unique_ptr<CopyPool> pool;
  std::vector<Chunk> chunks;
};

static int copy_threads() {
  static const int v = [] {
    const char* e = getenv("KP_COPY_THREADS");     // total threads of a staged copy, the caller included
    if (e) return std::max(1, std::min(16, atoi(e)));
    const unsigned hw = std::thread::hardware_concurrency();
    return hw >= 16 ? 4 : hw >= 4 ? 2 : 1;
  }();
  return v;
}

static size_t chunk_bytes() {
  static const size_t v = [] {
    const char* e = getenv("KP_COPY_CHUNK_KB");
    return (size_t)(e ? std::max(64, atoi(e)) : 2048) * 1024;
  }();
  return v;
}

static size_t pool_min_bytes() {
  static const size_t v = [] {
    const char* e = getenv("KP_COPY_POOL_MIN_MB");
    return (size_t)(e ? std::max(0, atoi(e)) : 32) << 20;
  }();
  return v;
}

static kp_stage* stage_of(kp_ctx* ctx) {
  if (ctx->stage) return ctx->stage;
  std::unique_ptr<kp_stage> st(new kp_stage());
  if (hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  const unsigned f = hipEventDisableTiming;
  if (hipEventCreateWithFlags(&st->ev_order, f) != hipSuccess || hipEventCreateWithFlags(&st->dma_done[0], f) != hipSuccess ||
      hipEventCreateWithFlags(&st->dma_done[1], f) != hipSuccess)
    return nullptr;
  st->pool.reset(new CopyPool(copy_threads() - 1));
  ctx->stage = st.release();
  return ctx->stage;
}

void kp_stage_destroy(kp_ctx* ctx) {
  kp_stage* st = ctx->stage;
  if (!st) return;
  if (st->copy_stream) (void)hipStreamSynchronize(st->copy_stream);
  st->pool.reset();
  for (int i = 0; i < 2; ++i) {
    if (st->pinned[i]) (void)hipHostFree(st->pinned[i]);
    if (st->dma_done[i]) (void)hipEventDestroy(st->dma_done[i]);
  }
  if (st->ev_order) (void)hipEventDestroy(st->ev_order);
  if (st->copy_stream) (void)hipStreamDestroy(st->copy_stream);
  delete st;
  ctx->stage = nullptr;
}

static void free_arrays(kp_snapshots* s) {
  if (s->alpha) (void)hipFree(s->alpha);
  if (s->beta) (void)hipFree(s->beta);
  if (s->u) (void)hipFree(s->u);
  s->alpha = s->beta = s->u = nullptr;
  s->cap_rows = 0;
}

// every array carries 64 doubles of zero padding behind its Ns * columns values: the Gram kernels prefetch up to three
// snapshot tiles (24 + 7 values) past the end of a row without bounds checks (the values are masked, the addresses must be mapped)
static const size_t kPad = 64 * sizeof(double);

static hipError_t alloc_arrays(kp_snapshots* s, int64_t rows) {
  const size_t bz = (size_t)rows * s->nzeta * sizeof(double), bu = (size_t)rows * s->m * sizeof(double);
  hipError_t e;
  if ((e = hipMalloc((void**)&s->alpha, bz + kPad)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&s->beta, bz + kPad)) != hipSuccess) return e;
  if ((e = hipMalloc((void**)&s->u, bu + kPad)) != hipSuccess) return e;
  s->cap_rows = rows;
  return hipSuccess;
}

// The staged, chunked copy.  `fresh`: nothing on the device can be reading the object (it was just allocated).
// `ld`: rows of the CALLER's column-major arrays (>= Ns; ld > Ns: the object takes Ns consecutive rows of every column -
// one device's share of a snapshot matrix that kp_multi_fit_sharded deals over several GPUs).
static int fill(kp_ctx* ctx, kp_snapshots* s, const double* alpha, const double* beta, const double* u, int64_t Ns, bool fresh, int64_t ld = -1) {
  if (ld < Ns) ld = Ns;
  kp_stage* st = stage_of(ctx);
  if (!st) return ctx->fail(KP_ERR_HIP, "kp_snapshots: could not create the copy stream");
  const size_t bz = (size_t)Ns * s->nzeta * sizeof(double), bu = (size_t)Ns * s->m * sizeof(double);
  const size_t total = 2 * bz + bu;
  const int slot = st->next;
  st->next ^= 1;
  if (st->busy[slot]) {                              // the DMA that last used this half of the ring
    KP_HIP(ctx, hipEventSynchronize(st->dma_done[slot]));
    st->busy[slot] = false;
  }
  if (st->cap[slot] < total) {
    if (st->pinned[slot]) (void)hipHostFree(st->pinned[slot]);
    st->pinned[slot] = nullptr;
    st->cap[slot] = 0;
    const size_t want = total + total / 8 + 4096;
    KP_HIP(ctx, hipHostMalloc(&st->pinned[slot], want, hipHostMallocDefault));
    st->cap[slot] = want;
  }
  // the DMA may not overtake kernels that still read the previous contents
  if (!fresh) {
    if (s->read_pending) {
      KP_HIP(ctx, hipStreamWaitEvent(st->copy_stream, s->ev_read, 0));
    } else if (!s->streaming) {                      // readers so far did not record: order behind everything enqueued
      KP_HIP(ctx, hipEventRecord(st->ev_order, ctx->stream));
      KP_HIP(ctx, hipStreamWaitEvent(st->copy_stream, st->ev_order, 0));
    }
  }
  if (fresh || Ns != s->Ns) {
    KP_HIP(ctx, hipMemsetAsync((char*)s->alpha + bz, 0, kPad, st->copy_stream));
    KP_HIP(ctx, hipMemsetAsync((char*)s->beta + bz, 0, kPad, st->copy_stream));
    KP_HIP(ctx, hipMemsetAsync((char*)s->u + bu, 0, kPad, st->copy_stream));
  }
  s->Ns = Ns;
  st->chunks.clear();
  {
    const size_t cb = chunk_bytes();
    char* stage = (char*)st->pinned[slot];
    auto cut = [&](double* dev, const double* src, size_t bytes) {
      for (size_t o = 0; o < bytes; o += cb) {
        const size_t nb = std::min(cb, bytes - o);
        st->chunks.push_back(Chunk{(char*)dev + o, stage, (const char*)src + o, nb});
        stage += nb;
      }
    };
    if (ld == Ns) {
      cut(s->alpha, alpha, bz);
      cut(s->beta, beta, bz);
      cut(s->u, u, bu);
    } else {                                           // a row range: column by column
      const size_t bcol = (size_t)Ns * sizeof(double);
      for (int c = 0; c < s->nzeta; ++c) cut(s->alpha + (size_t)c * Ns, alpha + (size_t)c * ld, bcol);
      for (int c = 0; c < s->nzeta; ++c) cut(s->beta + (size_t)c * Ns, beta + (size_t)c * ld, bcol);
      for (int c = 0; c < s->m; ++c) cut(s->u + (size_t)c * Ns, u + (size_t)c * ld, bcol);
    }
  }
  const int nch = (int)st->chunks.size();
  hipError_t e = hipSuccess;
  // measured (tools/upload_probe.py): for the 12 MB of 1e5 pairs one thread stages at PCIe speed from warm or cold host
  // memory and waking workers only adds jitter; at 120 MB the copy threads lift the rate from 29 to 40 GB/s
  const bool pooled = !st->pool->workers.empty() && total >= pool_min_bytes();
  if (pooled) st->pool->begin(st->chunks.data(), nch);
  for (int i = 0; i < nch; ++i) {
    const Chunk& c = st->chunks[i];
    if (pooled) st->pool->wait_chunk(i);
    else std::memcpy(c.stage, c.src, c.bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(c.dst_dev, c.stage, c.bytes, hipMemcpyHostToDevice, st->copy_stream);
  }
  if (pooled) st->pool->end();
  if (e != hipSuccess) return ctx->fail(KP_ERR_HIP, std::string("kp_snapshots: hipMemcpyAsync: ") + hipGetErrorString(e));
  KP_HIP(ctx, hipEventRecord(s->ev_ready, st->copy_stream));
  KP_HIP(ctx, hipEventRecord(st->dma_done[slot], st->copy_stream));
  st->busy[slot] = true;
  s->dma_pending = true;
  s->read_pending = false;
  return KP_OK;
}

extern "C" int kp_snapshots_upload(kp_ctx* ctx, const double* alpha, const double* beta, const double* u, int64_t Ns, int nzeta, int m,
                                   kp_snapshots** out) {
  if (!ctx || !out || Ns < 0 || nzeta < 1 || m < 0 || (Ns > 0 && (!alpha || !beta || (m > 0 && !u))))
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_snapshots_upload: bad argument") : KP_ERR_ARG;
  *out = nullptr;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<kp_snapshots, int (*)(kp_snapshots*)> s(new kp_snapshots(), kp_snapshots_destroy);
  s->ctx = ctx;
  s->nzeta = nzeta;
  s->m = m;
  hipError_t e = alloc_arrays(s.get(), Ns);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_read, hipEventDisableTiming);
  if (e != hipSuccess) return ctx->fail(KP_ERR_HIP, std::string("kp_snapshots_upload: ") + hipGetErrorString(e));
  int rc = fill(ctx, s.get(), alpha, beta, u, Ns, true);
  if (rc) return rc;
  // a new object is complete when this call returns (as before); only kp_snapshots_update leaves the DMA in flight
  KP_HIP(ctx, hipEventSynchronize(s->ev_ready));
  s->dma_pending = false;
  *out = s.release();
  return KP_OK;
}

// kp_snapshots_update for a row range of the caller's arrays (internal: kp_multi.hip): rows of length `ld`, the pointers
// already advanced to the first row of the range
int kp_snapshots_update_rows(kp_ctx* ctx, kp_snapshots* s, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t ld);

extern "C" int kp_snapshots_update(kp_ctx* ctx, kp_snapshots* s, const double* alpha, const double* beta, const double* u, int64_t Ns) {
  return kp_snapshots_update_rows(ctx, s, alpha, beta, u, Ns, Ns);
}

int kp_snapshots_update_rows(kp_ctx* ctx, kp_snapshots* s, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t ld) {
  if (!ctx || !s || s->ctx != ctx || Ns < 0 || (Ns > 0 && (!alpha || !beta || (s->m > 0 && !u))))
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_snapshots_update: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  bool fresh = false;
  if (Ns > s->cap_rows) {                            // grow: the old arrays must be idle before they are freed
    // idle the device WITHOUT consuming the deferred status of earlier fits (a NOT_SPD of a queued fit belongs to the
    // caller's next kp_synchronize / kp_fit_get_K, not to this refill): flush the queued solves, then wait for the streams
    int rc = kp_flush_pending(ctx);
    if (rc) return rc;
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) KP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    if (ctx->stage) KP_HIP(ctx, hipStreamSynchronize(ctx->stage->copy_stream));
    free_arrays(s);
    hipError_t e = alloc_arrays(s, Ns);
    if (e != hipSuccess) {
      // alloc_arrays may have failed part-way: leave an EMPTY object (no rows, no arrays) - a later fit on it fails with
      // "no snapshots" instead of launching kernels on null pointers
      free_arrays(s);
      s->Ns = 0;
      s->cap_rows = 0;
      return ctx->fail(KP_ERR_HIP, std::string("kp_snapshots_update: ") + hipGetErrorString(e));
    }
    fresh = true;
  }
  int rc = fill(ctx, s, alpha, beta, u, Ns, fresh, ld);
  if (rc) return rc;
  s->streaming = true;                               // from now on every reader records `ev_read`
  return KP_OK;
}

extern "C" int kp_snapshots_destroy(kp_snapshots* s) {
  if (!s) return KP_OK;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    if (s->dma_pending && s->ev_ready) (void)hipEventSynchronize(s->ev_ready);
    if (s->read_pending && s->ev_read) (void)hipEventSynchronize(s->ev_read);
  }
  free_arrays(s);
  if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
  if (s->ev_read) (void)hipEventDestroy(s->ev_read);
  delete s;
  return KP_OK;
}

// ---- page-locked host buffers for callers that assemble their inputs (see koopman_hip.h) ------------------------------
extern "C" int kp_host_alloc(kp_ctx* ctx, int64_t bytes, void** ptr) {
  if (!ctx || !ptr || bytes < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_host_alloc: bad argument") : KP_ERR_ARG;
  *ptr = nullptr;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  void* p = nullptr;
  KP_HIP(ctx, hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault));
  {
    std::lock_guard<std::mutex> lk(ctx->host_mu);
    ctx->host_blocks.push_back(p);
  }
  *ptr = p;
  return KP_OK;
}

extern "C" int kp_host_free(kp_ctx* ctx, void* ptr) {
  if (!ctx) return KP_ERR_ARG;
  if (!ptr) return KP_OK;
  bool mine = false;
  {
    std::lock_guard<std::mutex> lk(ctx->host_mu);
    for (size_t i = 0; i < ctx->host_blocks.size() && !mine; ++i)
      if (ctx->host_blocks[i] == ptr) {
        ctx->host_blocks.erase(ctx->host_blocks.begin() + (long)i);
        mine = true;
      }
  }
  if (mine) {
    KP_HIP(ctx, hipSetDevice(ctx->device));
    KP_HIP(ctx, hipDeviceSynchronize());            // no transfer from the block may still be in flight
    KP_HIP(ctx, hipHostFree(ptr));
    return KP_OK;
  }
  return ctx->fail(KP_ERR_ARG, "kp_host_free: not a block of this context");
}

void kp_host_free_all(kp_ctx* ctx) {
  for (void* p : ctx->host_blocks) (void)hipHostFree(p);
  ctx->host_blocks.clear();
}

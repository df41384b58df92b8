// Multi-GPU entry points: one process per GPU, RCCL over xGMI, loaded at run time (single-GPU use never touches librccl).
// The path shards by independent units (lasso values, random systems, MPC problems: train_models loop Ksysid.m:1372-1387,
// evaluate_rand_models.m:45-144) with NO collective on the data path; the only exchanges are the final gather of the
// results and, when ONE fit is sharded over snapshots, a single all-reduce of the Gram pair [G | C].
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kp_internal.h"

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // point-to-point (the gather to ONE rank): optional - without them kp_comm_gather_fits runs the all-gather
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false, p2p = false;
};

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    // the ROCm installation's library first, by path: a process that has also loaded a Python wheel's private copy of
    // RCCL (built against another HIP runtime) must not get that one under this library's streams and pointers
    const char* names[] = {getenv("KP_RCCL_PATH"), "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (a.handle) break;
    }
    if (!a.handle) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.Send = (decltype(a.Send))dlsym(a.handle, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(a.handle, "ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
    a.p2p = a.Send && a.Recv && a.GroupStart && a.GroupEnd && !getenv("KP_COMM_NO_P2P");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
}  // namespace

struct kp_comm_state {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

#define KP_NCCL(ctx, expr)                                                                              \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) return (ctx)->fail(KP_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
  } while (0)

extern "C" int kp_comm_unique_id(void* id128) {
  if (!id128) return KP_ERR_ARG;
  if (!rccl().ok) {
    kp_set_global_error("kp_comm_unique_id: librccl.so could not be loaded");
    return KP_ERR_HIP;
  }
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) {
    kp_set_global_error("kp_comm_unique_id: ncclGetUniqueId failed");
    return KP_ERR_HIP;
  }
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  std::memcpy(id128, &id, 128);
  return KP_OK;
}

extern "C" int kp_comm_create(kp_ctx* ctx, const void* id128, int rank, int world) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_create: bad argument") : KP_ERR_ARG;
  if (ctx->comm) return ctx->fail(KP_ERR_ARG, "kp_comm_create: this context already has a communicator");
  if (!rccl().ok) return ctx->fail(KP_ERR_HIP, "kp_comm_create: librccl.so could not be loaded");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = kp_synchronize(ctx);
  if (rc) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  kp_comm_state* cs = new kp_comm_state;
  cs->rank = rank;
  cs->world = world;
  ncclResult_t r = rccl().CommInitRank(&cs->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete cs;
    return ctx->fail(KP_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
  }
  if (ctx->comm_abandoned.load()) {
    // the caller's watchdog gave up on this bootstrap and the launch went on without RCCL (kp_comm_abandon): a communicator
    // that appears now must not be published - the other ranks will never enter a collective on it
    (void)rccl().CommDestroy(cs->comm);
    delete cs;
    return ctx->fail(KP_ERR_HIP, "kp_comm_create: abandoned by the caller before ncclCommInitRank returned");
  }
  ctx->comm = cs;
  return KP_OK;
}

// The caller's watchdog timed out on kp_comm_create (still blocked in another thread): from now on this context runs
// without a communicator, whatever that call does later.
extern "C" int kp_comm_abandon(kp_ctx* ctx) {
  if (!ctx) return KP_ERR_ARG;
  ctx->comm_abandoned.store(true);
  return KP_OK;
}

extern "C" int kp_comm_destroy(kp_ctx* ctx) {
  if (!ctx) return KP_ERR_ARG;
  if (!ctx->comm) return KP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm->comm) (void)rccl().CommDestroy(ctx->comm->comm);
  delete ctx->comm;
  ctx->comm = nullptr;
  return KP_OK;
}

extern "C" int kp_comm_info(const kp_ctx* ctx, int* rank, int* world) {
  if (!ctx) return KP_ERR_ARG;
  if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
  if (world) *world = ctx->comm ? ctx->comm->world : 1;
  return KP_OK;
}

// recv (world x bytes) = concatenation of every rank's send (bytes), rank order.  Host buffers; staged through HBM.
extern "C" int kp_comm_allgather(kp_ctx* ctx, const void* send, int64_t bytes, void* recv) {
  if (!ctx || !send || !recv || bytes < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_allgather: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->comm) {                               // single process: the gather of one
    std::memcpy(recv, send, (size_t)bytes);
    return KP_OK;
  }
  const int world = ctx->comm->world;
  const size_t b8 = ((size_t)bytes + 7) / 8 * 8;
  char* ws = (char*)ctx->workspace(8, b8 * (size_t)(world + 1));
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_comm_allgather: out of device memory");
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(ws, send, (size_t)bytes, hipMemcpyHostToDevice, s));
  KP_NCCL(ctx, rccl().AllGather(ws, ws + b8, b8, ncclChar, ctx->comm->comm, s));
  for (int r = 0; r < world; ++r)
    KP_HIP(ctx, hipMemcpyAsync((char*)recv + (size_t)r * bytes, ws + b8 * (size_t)(r + 1), (size_t)bytes, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// in-place sum over ranks of a host vector (barriers, timing maxima via +/-, small tables)
extern "C" int kp_comm_allreduce_sum(kp_ctx* ctx, double* inout, int64_t count) {
  if (!ctx || !inout || count < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_allreduce_sum: bad argument") : KP_ERR_ARG;
  if (!ctx->comm) return KP_OK;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  double* ws = (double*)ctx->workspace(8, (size_t)count * 8);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_comm_allreduce_sum: out of device memory");
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(ws, inout, (size_t)count * 8, hipMemcpyHostToDevice, s));
  KP_NCCL(ctx, rccl().AllReduce(ws, ws, (size_t)count, ncclDouble, ncclSum, ctx->comm->comm, s));
  KP_HIP(ctx, hipMemcpyAsync(inout, ws, (size_t)count * 8, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// device-resident all-reduce of the Gram pair of the current fit (used by kp_fit_sharded): G | C, 2 W^2 doubles
int kp_comm_allreduce_dev(kp_ctx* ctx, double* buf_dev, size_t count, hipStream_t s) {
  if (!ctx->comm) return KP_OK;
  KP_NCCL(ctx, rccl().AllReduce(buf_dev, buf_dev, count, ncclDouble, ncclSum, ctx->comm->comm, s));
  return KP_OK;
}

// K of fit `index` of the last asynchronous batch (or of the last synchronous kp_fit) of EVERY rank, gathered device to
// device and copied out once: K_all = world matrices W x W back to back, rank order.
extern "C" int kp_comm_allgather_fit(kp_ctx* ctx, int index, int W, double* K_all) {
  if (!ctx || !K_all || index < 0 || W != ctx->Kres_W) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_allgather_fit: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = kp_synchronize(ctx);
  if (rc) return rc;
  size_t slot = (size_t)index;
  if (ctx->kres_is_ring) {
    if (index >= ctx->async_count || index < ctx->async_count - ctx->kring_cap)
      return ctx->fail(KP_ERR_ARG, "kp_comm_allgather_fit: that fit is not in the result ring");
    slot = (size_t)(index % ctx->kring_cap);
  } else if (index >= ctx->Kres_n) {
    return ctx->fail(KP_ERR_ARG, "kp_comm_allgather_fit: index out of range");
  }
  const size_t cnt = (size_t)W * W;
  const double* src = ctx->Kres + slot * cnt;
  if (!ctx->comm) {
    KP_HIP(ctx, hipMemcpy(K_all, src, cnt * 8, hipMemcpyDeviceToHost));
    return KP_OK;
  }
  const int world = ctx->comm->world;
  double* ws = (double*)ctx->workspace(8, cnt * 8 * (size_t)world);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_comm_allgather_fit: out of device memory");
  hipStream_t s = ctx->stream;
  KP_NCCL(ctx, rccl().AllGather(src, ws, cnt, ncclDouble, ctx->comm->comm, s));
  KP_HIP(ctx, hipMemcpyAsync(K_all, ws, cnt * 8 * (size_t)world, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// The K stack of this rank's shard of a sweep - fits first .. first + count - 1 (kp_fit_get_K numbering: the values of the
// last synchronous kp_fit, or the fits of the last asynchronous batch) - of EVERY rank: one ncclAllGather of count W^2
// doubles straight from the result buffer, one copy out.  K_all = world x count matrices, rank-major.  Ranks whose shard is
// shorter than `count` (ragged round-robin deal) contribute padding slots of unspecified content.
// contiguous device source of `count` result slots from `first` on (padding where this rank's shard is shorter); *ws_out: the
// staging workspace (slot 8) sized for (extra_slots + 1) x count matrices - the source itself when a staging copy was needed
static int fits_source(kp_ctx* ctx, const char* who, int first, int count, int W, int world, size_t extra_blocks, const double** src_out, double** ws_out) {
  const size_t cnt = (size_t)W * W;
  hipStream_t s = ctx->stream;
  const bool have = ctx->Kres && W == ctx->Kres_W;          // a rank with an EMPTY shard has no results at all: all padding
  if (!have && world == 1) return ctx->fail(KP_ERR_ARG, std::string(who) + ": no results of that width");
  int avail = 0;                                             // valid fits from `first` on
  if (have) {
    if (ctx->kres_is_ring) {
      if (first + count > ctx->async_count && world == 1) return ctx->fail(KP_ERR_ARG, std::string(who) + ": range not in the result ring");   // one rank: nobody else to pad for
      if (first < ctx->async_count - ctx->kring_cap) return ctx->fail(KP_ERR_ARG, std::string(who) + ": range no longer in the result ring");
      avail = std::max(0, std::min(count, ctx->async_count - first));
    } else {
      if (first + count > ctx->Kres_n && world == 1) return ctx->fail(KP_ERR_ARG, std::string(who) + ": index out of range");
      avail = std::max(0, std::min(count, ctx->Kres_n - first));
    }
  }
  const int slot0 = have && ctx->kres_is_ring ? first % ctx->kring_cap : first;
  const bool wraps = have && ctx->kres_is_ring && slot0 + avail > ctx->kring_cap;
  const bool inside = have && !wraps && avail == count;
  double* ws = nullptr;
  if (extra_blocks || !inside) {
    ws = (double*)ctx->workspace(8, cnt * 8 * (size_t)count * (extra_blocks + 1));
    if (!ws) return ctx->fail(KP_ERR_HIP, std::string(who) + ": out of device memory");
  }
  if (inside) {
    *src_out = ctx->Kres + (size_t)slot0 * cnt;
  } else {
    for (int i = 0; i < avail; ++i) {
      const size_t sl = ctx->kres_is_ring ? (size_t)((first + i) % ctx->kring_cap) : (size_t)(first + i);
      KP_HIP(ctx, hipMemcpyAsync(ws + (size_t)i * cnt, ctx->Kres + sl * cnt, cnt * 8, hipMemcpyDeviceToDevice, s));
    }
    if (avail < count) KP_HIP(ctx, hipMemsetAsync(ws + (size_t)avail * cnt, 0, (size_t)(count - avail) * cnt * 8, s));
    *src_out = ws;
  }
  *ws_out = ws;
  return KP_OK;
}

extern "C" int kp_comm_allgather_fits(kp_ctx* ctx, int first, int count, int W, double* K_all) {
  if (!ctx || !K_all || first < 0 || count < 1 || W < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_allgather_fits: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = kp_synchronize(ctx);
  if (rc) return rc;
  const size_t cnt = (size_t)W * W;
  const int world = ctx->comm ? ctx->comm->world : 1;
  hipStream_t s = ctx->stream;
  const double* src = nullptr;
  double* ws = nullptr;
  rc = fits_source(ctx, "kp_comm_allgather_fits", first, count, W, world, ctx->comm ? (size_t)world : 0, &src, &ws);
  if (rc) return rc;
  if (!ctx->comm) {                               // no communicator: the gather of one (with one, a one-rank ncclAllGather runs)
    KP_HIP(ctx, hipMemcpyAsync(K_all, src, cnt * 8 * (size_t)count, hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    return KP_OK;
  }
  double* all = ws + (size_t)count * cnt;
  KP_NCCL(ctx, rccl().AllGather(src, all, cnt * (size_t)count, ncclDouble, ctx->comm->comm, s));
  KP_HIP(ctx, hipMemcpyAsync(K_all, all, cnt * 8 * (size_t)count * (size_t)world, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// The same stack gathered to ONE rank: every other rank sends its `count` matrices to `root` (ncclSend / ncclRecv in one
// group - xGMI is point to point, the root's seven links receive in parallel) and returns without a host copy; only the root
// pays the device-to-host transfer of world x count matrices.  What the caller of train_models needs: the reference's host
// is one interpreter (Ksysid.m:1370-1387), so one host wants the candidates - an all-gather hands the 58 MB stack of the
// 64-value grid to all eight ranks and has each of them copy it out.  K_root: world x count matrices on the root, rank-major;
// ignored (may be NULL) elsewhere.
extern "C" int kp_comm_gather_fits(kp_ctx* ctx, int root, int first, int count, int W, double* K_root) {
  if (!ctx || first < 0 || count < 1 || W < 1 || root < 0) return ctx ? ctx->fail(KP_ERR_ARG, "kp_comm_gather_fits: bad argument") : KP_ERR_ARG;
  const int world = ctx->comm ? ctx->comm->world : 1, rank = ctx->comm ? ctx->comm->rank : 0;
  if (root >= world || (rank == root && !K_root)) return ctx->fail(KP_ERR_ARG, "kp_comm_gather_fits: bad root / NULL result on the root");
  if (!ctx->comm) return kp_comm_allgather_fits(ctx, first, count, W, K_root);
  if (!rccl().p2p) {                              // no point-to-point entry points in this RCCL: the all-gather, result dropped off the root
    std::vector<double> tmp;
    double* dst = K_root;
    if (rank != root) { tmp.resize((size_t)world * count * W * W); dst = tmp.data(); }
    return kp_comm_allgather_fits(ctx, first, count, W, dst);
  }
  KP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = kp_synchronize(ctx);
  if (rc) return rc;
  const size_t cnt = (size_t)W * W, blk = cnt * (size_t)count;
  hipStream_t s = ctx->stream;
  const double* src = nullptr;
  double* ws = nullptr;
  rc = fits_source(ctx, "kp_comm_gather_fits", first, count, W, world, rank == root ? (size_t)world : 0, &src, &ws);
  if (rc) return rc;
  if (rank != root) {
    KP_NCCL(ctx, rccl().GroupStart());
    KP_NCCL(ctx, rccl().Send(src, blk, ncclDouble, root, ctx->comm->comm, s));
    KP_NCCL(ctx, rccl().GroupEnd());
    KP_HIP(ctx, hipStreamSynchronize(s));
    return KP_OK;
  }
  double* all = ws + blk;
  KP_HIP(ctx, hipMemcpyAsync(all + (size_t)root * blk, src, blk * 8, hipMemcpyDeviceToDevice, s));
  KP_NCCL(ctx, rccl().GroupStart());
  for (int r = 0; r < world; ++r)
    if (r != root) KP_NCCL(ctx, rccl().Recv(all + (size_t)r * blk, blk, ncclDouble, r, ctx->comm->comm, s));
  KP_NCCL(ctx, rccl().GroupEnd());
  KP_HIP(ctx, hipMemcpyAsync(K_root, all, blk * 8 * (size_t)world, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

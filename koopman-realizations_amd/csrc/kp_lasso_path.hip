// kp_lasso_path.hip - the L1-constrained fit of solve_KoopmanQP (Ksysid.m:1095-1176) by the regularisation-path homotopy.
//
// The QP of Ksysid.m:1126-1137 is  min 1/2 |Px K - Py|_F^2  s.t. |vec K|_1 <= t.  With the multiplier theta >= 0 of the L1 row the
// columns separate:  k_j(theta) = argmin 1/2 k'G k - c_j'k + theta |k|_1,  and  sum_j |k_j(theta)|_1 = t  fixes theta.  Every
// k_j(theta) is piecewise linear in theta (LARS with drops): from theta = max|c_j| (k = 0) downwards the support S and the signs s
// stay fixed between breakpoints, dk_S / d(-theta) = G_SS^-1 s_S, and a breakpoint is the first theta at which an entry off the
// support reaches |c - G k|_i = theta (it enters) or an entry on it reaches zero (it leaves).  The path does not depend on t: ONE
// path per column serves every lasso value of a fit (the train_models loop over a lasso vector, Ksysid.m:1372-1387).
//
// Why: the projected-gradient iteration of kp_lasso.hip needs O(sqrt(cond G)) iterations and the monomial dictionaries of the
// reference's own data (arm markers: cond(G) = 1e10 after the 1e-6 PSD guard of :1117-1120) left it at its cap for budgets within
// a decade of |K_LS|_1, where the multiplier is tiny and the answer dense (tools/lasso_illcond_probe.py).  The
// homotopy is exact in a bounded number of steps (3-6 W per column on those Grams) whatever the conditioning.
//
// One workgroup per column.  The inverse M = G_SS^-1 is kept EXPLICITLY and updated by bordering / deletion (rank-1 updates: one
// parallel pass, no triangular solves - a dependent chain of |S| steps per solve would cost more than everything else), with
// one step of iterative refinement of the bordering vector against G itself, which is what keeps M accurate at cond 1e10
// (tools/lasso_homotopy_probe2.py: without it the adds break down, with it the KKT conditions hold to 1e-10 after 450 steps).
// M lives in LDS while the support fits beside the vectors (136 entries at W = 136, 128 at W = 384); when a support outgrows it the
// column states are copied into the layout of the global-memory form of the kernel (template flag) and the walk goes on there.
//
//   walk     all columns walk the path in rounds (theta targets theta_max / 16^r) and record (theta, |k|_1) at every breakpoint;
//            after each round the host reads sum_j |k_j|_1: the values whose budget it now covers are bracketed by the round
//   theta    sum_j |k_j(theta)|_1 is piecewise linear and decreasing: theta_v with sum = t_v by bisection on the recorded
//            breakpoints, one workgroup per bracketed value
//   answers  from a copy of the state at the round's START the walk is repeated with stops at the theta_v in decreasing order
//            and k_j(theta_v) written at each stop (only the bracketing round is walked twice); the walk then goes on
#include "kp_internal.h"
#include <algorithm>
#include <chrono>
#include <numeric>
#include <vector>

namespace {
constexpr int P_TPB_LDS = 256;      // threads per column, inverse in LDS
#ifndef KP_P_TPB_GLOBAL
#define KP_P_TPB_GLOBAL 512
#endif
// ... inverse in global memory (supports beyond 128: the products are long).  512, not 1024 (round 5): a column's walk is a serial
// chain, so what counts is that ALL columns walk at once - 336 workgroups of 1024 threads are 1.3 rounds of one per CU, of 512
// threads two fit a CU and every column is resident: arm Gram at W = 336, budgets 0.5 / 0.2 / 0.1 |K_LS|_1: 90 / 52 / 33 ->
// 72 / 40 / 27 ms (256 threads: 69 / 41 / 28; tools/lasso_path_time.py)
constexpr int P_TPB_GLOBAL = KP_P_TPB_GLOBAL;
constexpr int P_LDS_BYTES = 160 * 1024;   // LDS of a CU: the inverse takes what the vectors leave (128 entries at W = 384, 136 at W = 136)
constexpr int P_WMAX = 512;         // widest dictionary (the library's own limit; the W-length vectors live in LDS)
constexpr int P_RESYNC = 16;        // steps between re-synchronisations of r = c - G k and of r_S = theta s_S
enum { PATH_OK = 0, PATH_OVERFLOW = 1, PATH_STEPS = 2, PATH_SINGULAR = 3, PATH_BP = 4, PATH_ADJUST = 5 };

struct PathHdr {                    // 64 bytes at the head of a column's state
  double theta, l1, last_del_sgn, theta0;
  int steps, cnt, last_add, last_del, status, nbp;
  double slope;                     // d|k|_1 / d(-theta) at the current point: s_S' G_SS^-1 s_S
};
static_assert(sizeof(PathHdr) == 64, "header size");

struct PathLayout {
  int W, ldm, cap;                  // width, leading dimension / capacity of M and of the support vectors, breakpoints per column
  size_t off_k, off_r, off_sgn, off_idx, off_M, off_bpt, off_bpl, off_bpe, stride;
};

// dynamic LDS of kp_lasso_path_kernel: [M] + 4 W-vectors + 5 support vectors + the reduction scratch, then the index lists and scalars
static size_t path_lds_bytes(int W, int ldm, bool mglobal, int tpb) {
  return ((mglobal ? 0 : (size_t)ldm * ldm) + 4 * (size_t)W + 5 * (size_t)ldm + tpb) * 8 + ((size_t)ldm + 2 + W + 2 + 24) * 4 + 40 * 8;
}

static PathLayout make_layout(int W, bool mglobal, int cap) {
  PathLayout L;
  L.W = W;
  L.ldm = W;
  if (!mglobal) {                                    // largest support whose inverse fits LDS beside the vectors (path_lds_bytes)
    while (L.ldm > 8 && path_lds_bytes(W, L.ldm, false, P_TPB_LDS) > (size_t)P_LDS_BYTES) --L.ldm;
  }
  L.cap = cap;
  size_t o = sizeof(PathHdr);
  L.off_k = o; o += (size_t)W * 8;
  L.off_r = o; o += (size_t)W * 8;
  L.off_sgn = o; o += (size_t)W * 8;
  L.off_idx = o; o += (((size_t)W * 4 + 7) & ~(size_t)7);
  L.off_M = o; o += (size_t)L.ldm * L.ldm * 8;
  L.off_bpt = o; o += (size_t)cap * 8;
  L.off_bpl = o; o += (size_t)cap * 8;
  L.off_bpe = o; o += (((size_t)cap * 4 + 7) & ~(size_t)7);      // the event behind every breakpoint (debugging aid: KP_LASSO_PATH_DEBUG)
  L.stride = (o + 255) & ~(size_t)255;
  return L;
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// y[t] = sum_{q < n} M[q * ld + t] x[q], t < n  (M symmetric: column q read with consecutive t).  TP = 64 ... TPB threads of rows
// times P = TPB / TP column ranges, partial sums combined in a fixed order through `red` (TPB doubles).
template <int TPB>
__device__ __forceinline__ void mv_sym(const double* M, int ld, int n, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ red) {
  const int tid = threadIdx.x;
  if (n <= 0) { __syncthreads(); return; }
  int TP = 64;
  while (TP < n && TP < TPB) TP *= 2;
  const int P = TPB / TP;
  if (P == 1) {                                     // every thread owns rows tid, tid + TPB, ... over all columns
    for (int t = tid; t < n; t += TPB) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int q = 0;
      for (; q + 7 < n; q += 8) {
        const double m0 = M[(size_t)q * ld + t], m1 = M[(size_t)(q + 1) * ld + t], m2 = M[(size_t)(q + 2) * ld + t], m3 = M[(size_t)(q + 3) * ld + t];
        const double m4 = M[(size_t)(q + 4) * ld + t], m5 = M[(size_t)(q + 5) * ld + t], m6 = M[(size_t)(q + 6) * ld + t], m7 = M[(size_t)(q + 7) * ld + t];
        a0 += m0 * x[q]; a1 += m1 * x[q + 1]; a2 += m2 * x[q + 2]; a3 += m3 * x[q + 3];
        a0 += m4 * x[q + 4]; a1 += m5 * x[q + 5]; a2 += m6 * x[q + 6]; a3 += m7 * x[q + 7];
      }
      for (; q + 3 < n; q += 4) {
        a0 += M[(size_t)q * ld + t] * x[q];
        a1 += M[(size_t)(q + 1) * ld + t] * x[q + 1];
        a2 += M[(size_t)(q + 2) * ld + t] * x[q + 2];
        a3 += M[(size_t)(q + 3) * ld + t] * x[q + 3];
      }
      for (; q < n; ++q) a0 += M[(size_t)q * ld + t] * x[q];
      y[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    return;
  }
  const int t = tid & (TP - 1), p = tid / TP;
  const int q0 = (int)((long)n * p / P), q1 = (int)((long)n * (p + 1) / P);
  double a0 = 0.0, a1 = 0.0;
  if (t < n) {
    int q = q0;
    for (; q + 7 < q1; q += 8) {                      // eight loads of M in flight (the loop is bound by their latency, not by the FMAs)
      const double m0 = M[(size_t)q * ld + t], m1 = M[(size_t)(q + 1) * ld + t], m2 = M[(size_t)(q + 2) * ld + t], m3 = M[(size_t)(q + 3) * ld + t];
      const double m4 = M[(size_t)(q + 4) * ld + t], m5 = M[(size_t)(q + 5) * ld + t], m6 = M[(size_t)(q + 6) * ld + t], m7 = M[(size_t)(q + 7) * ld + t];
      a0 += m0 * x[q]; a1 += m1 * x[q + 1]; a0 += m2 * x[q + 2]; a1 += m3 * x[q + 3];
      a0 += m4 * x[q + 4]; a1 += m5 * x[q + 5]; a0 += m6 * x[q + 6]; a1 += m7 * x[q + 7];
    }
    for (; q < q1; ++q) a0 += M[(size_t)q * ld + t] * x[q];
  }
  red[p * TP + t] = a0 + a1;
  __syncthreads();
  if (p == 0 && t < n) {
    double s = red[t];
    for (int pp = 1; pp < P; ++pp) s += red[pp * TP + t];
    y[t] = s;
  }
  __syncthreads();
}

// M[q * ld + t] += sign * (u[t] u[q]) * inv for t, q < n: the bordering / deletion update (symmetric bit for bit)
template <int TPB>
__device__ __forceinline__ void rank1(double* M, int ld, int n, const double* __restrict__ u, double sinv) {
  const int tid = threadIdx.x;
  int TP = 64;
  while (TP < n && TP < TPB) TP *= 2;
  const int P = TPB / TP, t0 = tid & (TP - 1), pp = tid / TP;
  for (int t = t0; t < n; t += TP) {
    const double ut = u[t];
    int q = pp;
    for (; q + 3 * P < n; q += 4 * P) {               // four read-modify-writes in flight
      const double m0 = M[(size_t)q * ld + t], m1 = M[(size_t)(q + P) * ld + t], m2 = M[(size_t)(q + 2 * P) * ld + t], m3 = M[(size_t)(q + 3 * P) * ld + t];
      M[(size_t)q * ld + t] = m0 + (ut * u[q]) * sinv;
      M[(size_t)(q + P) * ld + t] = m1 + (ut * u[q + P]) * sinv;
      M[(size_t)(q + 2 * P) * ld + t] = m2 + (ut * u[q + 2 * P]) * sinv;
      M[(size_t)(q + 3 * P) * ld + t] = m3 + (ut * u[q + 3 * P]) * sinv;
    }
    for (; q < n; q += P) M[(size_t)q * ld + t] += (ut * u[q]) * sinv;
  }
}

// y[r] = sum_{q < n} G[rows[r] + idx[q] * W] x[q], r < nr  (rows == nullptr: rows[r] = r).  A row is shared by PW adjacent lanes
// (PW a power of two, TPB threads cover TPB / PW rows per pass) so that many independent L2 loads are in flight whatever nr is.
template <int TPB>
__device__ __forceinline__ void mv_gather(const double* __restrict__ G, int W, const int* __restrict__ rows, int nr, const int* __restrict__ idx, int n,
                                          const double* __restrict__ x, double* __restrict__ y) {
  const int tid = threadIdx.x;
  int PW = 1;
  while (PW < 64 && nr * PW * 2 <= TPB) PW *= 2;
  while (PW > 1 && n < PW) PW >>= 1;
  const int part = tid & (PW - 1), rslot = tid / PW, rpp = TPB / PW;
  for (int r0 = 0; r0 < nr; r0 += rpp) {
    const int r = r0 + rslot;
    double acc = 0.0;
    if (r < nr) {
      const int i = rows ? rows[r] : r;
      const double* gi = G + i;
      int q = part;
      for (; q + 3 * PW < n; q += 4 * PW) {
        const double g0 = gi[(size_t)idx[q] * W], g1 = gi[(size_t)idx[q + PW] * W], g2 = gi[(size_t)idx[q + 2 * PW] * W], g3 = gi[(size_t)idx[q + 3 * PW] * W];
        acc += g0 * x[q];
        acc += g1 * x[q + PW];
        acc += g2 * x[q + 2 * PW];
        acc += g3 * x[q + 3 * PW];
      }
      for (; q < n; q += PW) acc += gi[(size_t)idx[q] * W] * x[q];
    }
    for (int o = PW >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (r < nr && part == 0) y[r] = acc;
  }
  __syncthreads();
}

struct Cand { double dl; int key; };                 // key = 2 * index + (sign < 0) for an entry, 2 * index | 0x40000000 for a leave
__device__ __forceinline__ Cand cmin(Cand a, Cand b) { return (b.dl < a.dl || (b.dl == a.dl && b.key < a.key)) ? b : a; }

template <bool MG, int TPB, bool GL>
__global__ __launch_bounds__(TPB) void kp_lasso_path_kernel(const double* __restrict__ Gg, const double* __restrict__ C, PathLayout L, char* __restrict__ arena,
                                                           double theta_stop, double theta_from, int max_steps, int init, int record, int adjust, int polish, double* __restrict__ Kout) {
  extern __shared__ double sm[];
  const int W = L.W, ld = L.ldm, tid = threadIdx.x, col = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PT = TPB, NW = TPB / 64;
  char* base = arena + (size_t)col * L.stride;
  PathHdr* hdr = reinterpret_cast<PathHdr*>(base);
  double* gk = reinterpret_cast<double*>(base + L.off_k);
  double* gr = reinterpret_cast<double*>(base + L.off_r);
  double* gs = reinterpret_cast<double*>(base + L.off_sgn);
  int* gidx = reinterpret_cast<int*>(base + L.off_idx);
  double* gM = reinterpret_cast<double*>(base + L.off_M);
  double* bpt = reinterpret_cast<double*>(base + L.off_bpt);
  double* bpl = reinterpret_cast<double*>(base + L.off_bpl);
  int* bpe = reinterpret_cast<int*>(base + L.off_bpe);
  const double* c = C + (size_t)col * W;
  // LDS: [M (LDS mode)] [G (W <= 96)] k r sgn a (W each) | sS d u g e (ld each) | red (TPB) | idx (ld) offl (W) ints | scalars
  double* p = sm;
  double* Ml = p; if (!MG) p += (size_t)ld * ld;
  double* Gl = p; if (GL) p += (size_t)W * W;       // G itself, when it fits beside the inverse (W <= 96): its gathers stay out of L2
  double* k = p; p += W;
  double* r = p; p += W;
  double* sg = p; p += W;
  double* a = p; p += W;
  double* sS = p; p += ld;
  double* d = p; p += ld;
  double* u = p; p += ld;
  double* g = p; p += ld;
  double* e = p; p += ld;
  double* red = p; p += TPB;
  int* idx = reinterpret_cast<int*>(p);
  int* offl = idx + ld + (ld & 1);
  int* sc_i = offl + W + (W & 1);                    // [0] noff, [1..16] per-wave counts, [17] position of the leaving entry
  double* sc_d = reinterpret_cast<double*>(sc_i + 24);  // [0..15] wave minima dl, [16..23] their keys (ints), [24..39] wave sums
  int* sc_k = reinterpret_cast<int*>(sc_d + 16);
  double* sc_s = sc_d + 24;
  auto sum_waves = [&]() { double v = sc_s[0]; for (int w = 1; w < NW; ++w) v += sc_s[w]; return v; };
  double* const M = MG ? gM : Ml;                    // the inverse on the support: global memory or LDS
  const double* const G = GL ? Gl : Gg;
  if (GL) {
    for (int e2 = tid; e2 < W * W; e2 += PT) Gl[e2] = Gg[e2];
    __syncthreads();
  }

  double theta, l1, last_del_sgn;
  int steps, cnt, last_add, last_del, status, nbp;
  if (init) {
    for (int i = tid; i < W; i += PT) { k[i] = 0.0; r[i] = c[i]; sg[i] = 0.0; }
    __syncthreads();
    // first entry: the largest |c_i| (ties: the smallest index)
    Cand best{-1.0, 0};
    for (int i = tid; i < W; i += PT) { const double v = fabs(r[i]); if (v > best.dl || (v == best.dl && i < best.key)) best = Cand{v, i}; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      Cand b{__shfl_xor(best.dl, o, 64), __shfl_xor(best.key, o, 64)};
      if (b.dl > best.dl || (b.dl == best.dl && b.key < best.key)) best = b;
    }
    if (lane == 0) { sc_d[wave] = best.dl; sc_k[wave] = best.key; }
    __syncthreads();
    best = Cand{sc_d[0], sc_k[0]};
    for (int w = 1; w < NW; ++w) { const Cand b{sc_d[w], sc_k[w]}; if (b.dl > best.dl || (b.dl == best.dl && b.key < best.key)) best = b; }
    __syncthreads();
    theta = best.dl; l1 = 0.0; last_del_sgn = 0.0; steps = 0; cnt = 0; last_add = -1; last_del = -1; status = PATH_OK; nbp = 0;
    const int j0 = best.key;
    const double gjj = G[(size_t)j0 * W + j0];
    if (theta > 0.0 && gjj > 0.0) {
      if (tid == 0) { idx[0] = j0; sg[j0] = r[j0] > 0.0 ? 1.0 : -1.0; M[0] = 1.0 / gjj; }
      cnt = 1; last_add = j0;
    } else {
      theta = 0.0;                                    // c = 0: the column is zero for every theta
    }
    if (tid == 0) {
      hdr->theta0 = theta;
      if (record) { bpt[0] = theta; bpl[0] = 0.0; }
    }
    nbp = 1;
    __syncthreads();
  } else {
    theta = hdr->theta; l1 = hdr->l1; last_del_sgn = hdr->last_del_sgn;
    steps = hdr->steps; cnt = hdr->cnt; last_add = hdr->last_add; last_del = hdr->last_del; status = hdr->status; nbp = hdr->nbp;
    for (int i = tid; i < W; i += PT) { k[i] = gk[i]; r[i] = gr[i]; sg[i] = gs[i]; }
    for (int t = tid; t < cnt; t += PT) idx[t] = gidx[t];
    if (!MG)
      for (int q = 0; q < cnt; ++q)
        for (int t = tid; t < cnt; t += PT) Ml[(size_t)q * ld + t] = gM[(size_t)q * ld + t];
    __syncthreads();
  }

  // r = c - G k from scratch, k_S corrected so that r_S = theta s_S (one Newton step with the inverse at hand), r again
  auto resync = [&]() {
    for (int pass = 0; pass < 2; ++pass) {
      for (int t = tid; t < cnt; t += PT) u[t] = k[idx[t]];
      __syncthreads();
      mv_gather<TPB>(G, W, nullptr, W, idx, cnt, u, a);
      for (int i = tid; i < W; i += PT) r[i] = c[i] - a[i];
      __syncthreads();
      if (pass == 1) break;
      for (int t = tid; t < cnt; t += PT) e[t] = r[idx[t]] - theta * sg[idx[t]];
      __syncthreads();
      mv_sym<TPB>(M, ld, cnt, e, g, red);
      for (int t = tid; t < cnt; t += PT) k[idx[t]] += g[t];
      __syncthreads();
    }
  };
  // adjust: ONE move to theta_stop (above or below theta) along the current segment, no event - the caller knows it is tiny (an
  // entry pushed past zero by it, at the 1e-12 level, leaves in the next step of the walk)
  bool adjusted = false;
  bool d_fresh = true;
  while (status == PATH_OK && (adjust ? (!adjusted && cnt > 0 && theta == theta_from && theta_stop != theta) : theta > theta_stop)) {
    if (steps >= max_steps) { status = PATH_STEPS; break; }
    ++steps;
    adjusted = true;
    // ---- every P_RESYNC steps: r = c - G k from scratch, k_S corrected so that r_S = theta s_S, r again
    if (steps % P_RESYNC == 0 && cnt > 0 && !adjust) { resync(); d_fresh = true; }
    // ---- direction on the support (d = M s_S: formed afresh at the start of a launch and after a re-synchronisation, otherwise
    // carried through the bordering / deletion of M in O(|S|) - one pass over M less per step), its image off the support
    for (int t = tid; t < cnt; t += PT) sS[t] = sg[idx[t]];
    if (tid < 24) sc_i[tid] = 0;
    __syncthreads();
    if (d_fresh) {
      mv_sym<TPB>(M, ld, cnt, sS, d, red);
      d_fresh = false;
    }
    // compact list of the rows off the support (order: by index)
    {
      int noff = 0;
      for (int i0 = 0; i0 < W; i0 += PT) {
        const int i = i0 + tid;
        const bool off = i < W && sg[i] == 0.0;
        const unsigned long long mk = __ballot(off);
        if (lane == 0) sc_i[1 + wave] = __popcll(mk);
        __syncthreads();
        int before = noff;
        for (int w = 0; w < wave; ++w) before += sc_i[1 + w];
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += sc_i[1 + w];
        if (off) offl[before + __popcll(mk & ((1ull << lane) - 1ull))] = i;
        noff += tot;
        __syncthreads();
      }
      if (tid == 0) sc_i[0] = noff;
      __syncthreads();
    }
    const int noff = sc_i[0];
    mv_gather<TPB>(G, W, offl, noff, idx, cnt, d, a);     // a[rr] = (G d)_i for i = offl[rr]   (on the support (G d)_i = s_i)
    // ---- the first event
    Cand best{theta - theta_stop, 0x7fffffff};
    if (!adjust)
    for (int rr = tid; rr < noff; rr += PT) {
      const int i = offl[rr];
      const double ai = a[rr], ri = r[i];
#pragma unroll
      for (int sgi = 0; sgi < 2; ++sgi) {
        const double s = sgi ? -1.0 : 1.0;
        const double den = s * ai - 1.0, num = s * ri - theta;
        if (den != 0.0 && !(i == last_del && s == last_del_sgn)) {
          const double dl = num / den;
          if (dl > 1e-14 * theta) best = cmin(best, Cand{dl, 2 * i + sgi});
          else if (num >= 0.0 && den < 0.0) best = cmin(best, Cand{0.0, 2 * i + sgi});     // on the boundary and moving out (a tie with the
        }                                                                                  // event just taken: duplicate columns): enters now
      }
    }
    if (!adjust)
    for (int t = tid; t < cnt; t += PT) {
      const int i = idx[t];
      if (d[t] * sg[i] < 0.0 && i != last_add) best = cmin(best, Cand{fmax(-k[i] / d[t], 0.0), (2 * i) | 0x40000000});
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = cmin(best, Cand{__shfl_xor(best.dl, o, 64), __shfl_xor(best.key, o, 64)});
    if (lane == 0) { sc_d[wave] = best.dl; sc_k[wave] = best.key; }
    __syncthreads();
    best = Cand{sc_d[0], sc_k[0]};
    for (int w = 1; w < NW; ++w) best = cmin(best, Cand{sc_d[w], sc_k[w]});
    const bool capped = best.key == 0x7fffffff;
    const double dl = best.dl;
    const double theta_new = capped ? theta_stop : theta - dl;
    // ---- move
    double part = 0.0;
    for (int t = tid; t < cnt; t += PT) {
      const int i = idx[t];
      const double kn = k[i] + dl * d[t];
      k[i] = kn;
      r[i] = theta_new * sS[t];
      part += fabs(kn);
    }
    for (int rr = tid; rr < noff; rr += PT) { const int i = offl[rr]; r[i] -= dl * a[rr]; }
    part = wsum(part);
    if (lane == 0) sc_s[wave] = part;
    __syncthreads();
    l1 = sum_waves();
    theta = theta_new;
    if (!adjust) { last_add = -1; last_del = -1; last_del_sgn = 0.0; }
    __syncthreads();
    if (!capped) {
      const int ei = (best.key & 0x3fffffff) >> 1;
      if (best.key & 0x40000000) {
        // ---- entry ei leaves: M <- M - m m' / m_q on the rest, the last position moves into the hole
        int qd = 0;
        for (int t = tid; t < cnt; t += PT) if (idx[t] == ei) sc_i[17] = t;
        __syncthreads();
        qd = sc_i[17];
        for (int t = tid; t < cnt; t += PT) u[t] = M[(size_t)qd * ld + t];
        __syncthreads();
        const double inv = 1.0 / u[qd];
        l1 -= fabs(k[ei]);
        last_del = ei; last_del_sgn = sg[ei];
        rank1<TPB>(M, ld, cnt, u, -inv);
        {                                               // d of the remaining support: d - m d_q / m_q
          const double dq = d[qd] * inv;
          __syncthreads();
          for (int t = tid; t < cnt; t += PT) d[t] -= u[t] * dq;
        }
        __syncthreads();
        const int last = cnt - 1;
        if (qd != last) {
          for (int t = tid; t < last; t += PT)
            if (t != qd) { const double v = M[(size_t)last * ld + t]; M[(size_t)qd * ld + t] = v; M[(size_t)t * ld + qd] = v; }
          if (tid == 0) { M[(size_t)qd * ld + qd] = M[(size_t)last * ld + last]; }
        }
        __syncthreads();
        if (tid == 0) { if (qd != last) { idx[qd] = idx[last]; d[qd] = d[last]; } sg[ei] = 0.0; k[ei] = 0.0; }
        cnt = last;
        __syncthreads();
      } else {
        // ---- entry ei enters with sign s: u = M g refined once against G_SS, alpha = G_pp - g'u, M bordered
        const double s = (best.key & 1) ? -1.0 : 1.0;
        if (cnt >= ld) { status = MG ? PATH_SINGULAR : PATH_OVERFLOW; }
        else {
          for (int t = tid; t < cnt; t += PT) g[t] = G[(size_t)ei * W + idx[t]];
          __syncthreads();
          mv_sym<TPB>(M, ld, cnt, g, u, red);
          mv_gather<TPB>(G, W, idx, cnt, idx, cnt, u, e);                   // e = G_SS u
          for (int t = tid; t < cnt; t += PT) e[t] = g[t] - e[t];
          __syncthreads();
          mv_sym<TPB>(M, ld, cnt, e, a, red);                                // (a is free here: rebuilt every step)
          double dot = 0.0, us = 0.0;
          for (int t = tid; t < cnt; t += PT) { const double ut = u[t] + a[t]; u[t] = ut; dot += g[t] * ut; us += ut * sS[t]; }
          dot = wsum(dot);
          us = wsum(us);
          if (lane == 0) { sc_s[wave] = dot; sc_d[wave] = us; }
          __syncthreads();
          const double gpp = G[(size_t)ei * W + ei];
          const double alpha = gpp - (sum_waves());
          double us_all = sc_d[0];
          for (int w = 1; w < NW; ++w) us_all += sc_d[w];
          if (!(alpha > 1e-10 * gpp)) {
            // the entering column is (numerically) a combination of the support's columns - an exact twin in a Gram that got past
            // the factorisation without the guard.  It adds nothing to the fit: barred from this column's support for good
            // (sign 3: neither on the support nor in the list of rows off it), its weight stays with its twins
            if (tid == 0) sg[ei] = 3.0;
          } else {
            const double inv = 1.0 / alpha;
            rank1<TPB>(M, ld, cnt, u, inv);
            for (int t = tid; t < cnt; t += PT) { const double v = -u[t] * inv; M[(size_t)cnt * ld + t] = v; M[(size_t)t * ld + cnt] = v; }
            // d of the bordered support: [d + u (u's - s_p) / alpha ; (s_p - u's) / alpha]
            const double coef = (us_all - s) * inv;
            for (int t = tid; t < cnt; t += PT) d[t] += u[t] * coef;
            if (tid == 0) { M[(size_t)cnt * ld + cnt] = inv; idx[cnt] = ei; sg[ei] = s; d[cnt] = -coef; }
            cnt += 1;
            last_add = ei;
          }
          __syncthreads();
        }
      }
    }
    if (record) {
      if (nbp >= L.cap) { status = PATH_BP; }
      else {
        if (tid == 0) { bpt[nbp] = theta; bpl[nbp] = l1; bpe[nbp] = capped ? -1 : best.key; }
        ++nbp;
      }
    }
  }

  // ---- slope of |k|_1 along the current segment (the budget correction of the caller), state back to memory, the column
  __syncthreads();
  double slope = 0.0;
  if (status == PATH_OK && cnt > 0) {
    if (polish) {                                     // an answer leaves from here: at the rounding level of r_S = theta s_S
      resync();
      double part = 0.0;
      for (int t = tid; t < cnt; t += PT) part += fabs(k[idx[t]]);
      part = wsum(part);
      if (lane == 0) sc_s[wave] = part;
      __syncthreads();
      l1 = sum_waves();
      __syncthreads();
    }
    for (int t = tid; t < cnt; t += PT) sS[t] = sg[idx[t]];
    __syncthreads();
    mv_sym<TPB>(M, ld, cnt, sS, d, red);
    double part = 0.0;
    for (int t = tid; t < cnt; t += PT) part += sS[t] * d[t];
    part = wsum(part);
    if (lane == 0) sc_s[wave] = part;
    __syncthreads();
    slope = sum_waves();
  }
  for (int i = tid; i < W; i += PT) { gk[i] = k[i]; gr[i] = r[i]; gs[i] = sg[i]; }
  for (int t = tid; t < cnt; t += PT) gidx[t] = idx[t];
  if (!MG)
    for (int q = 0; q < cnt; ++q)
      for (int t = tid; t < cnt; t += PT) gM[(size_t)q * ld + t] = Ml[(size_t)q * ld + t];
  if (Kout)
    for (int i = tid; i < W; i += PT) Kout[(size_t)col * W + i] = k[i];
  if (tid == 0) {
    hdr->theta = theta; hdr->l1 = l1; hdr->last_del_sgn = last_del_sgn;
    hdr->steps = steps; hdr->cnt = cnt; hdr->last_add = last_add; hdr->last_del = last_del; hdr->status = status; hdr->nbp = nbp;
    hdr->slope = slope;
  }
}

// out[0] = sum_j |k_j|_1, out[1] = max_j theta0_j, out[2] = max_j status, out[3] = sum_j steps, out[4] = max_j theta_j, out[5] = max_j cnt,
// out[6] = sum_j slope_j
__global__ __launch_bounds__(256) void kp_lasso_path_sum_kernel(const char* __restrict__ arena, size_t stride, int ncols, double* __restrict__ out) {
  __shared__ double red[4][7];
  double s = 0.0, t0 = 0.0, st = 0.0, sp = 0.0, th = 0.0, mc = 0.0, sl = 0.0;
  for (int j = threadIdx.x; j < ncols; j += 256) {
    const PathHdr* h = reinterpret_cast<const PathHdr*>(arena + (size_t)j * stride);
    s += h->l1; t0 = fmax(t0, h->theta0); st = fmax(st, (double)h->status); sp += (double)h->steps; th = fmax(th, h->theta); mc = fmax(mc, (double)h->cnt); sl += h->slope;
  }
  // fixed-order reduction: lanes by xor tree, waves in order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64); sp += __shfl_xor(sp, o, 64); sl += __shfl_xor(sl, o, 64);
    t0 = fmax(t0, __shfl_xor(t0, o, 64)); st = fmax(st, __shfl_xor(st, o, 64)); th = fmax(th, __shfl_xor(th, o, 64)); mc = fmax(mc, __shfl_xor(mc, o, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = s; red[wave][1] = t0; red[wave][2] = st; red[wave][3] = sp; red[wave][4] = th; red[wave][5] = mc; red[wave][6] = sl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    out[1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    out[2] = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
    out[3] = red[0][3] + red[1][3] + red[2][3] + red[3][3];
    out[4] = fmax(fmax(red[0][4], red[1][4]), fmax(red[2][4], red[3][4]));
    out[5] = fmax(fmax(red[0][5], red[1][5]), fmax(red[2][5], red[3][5]));
    out[6] = red[0][6] + red[1][6] + red[2][6] + red[3][6];
  }
}

// |k_j(theta)|_1 from the recorded breakpoints of column j (theta decreasing along the list, linear in between)
__device__ __forceinline__ double path_l1_at(const double* __restrict__ bt, const double* __restrict__ bl, int n, double theta) {
  if (n <= 0 || theta >= bt[0]) return 0.0;
  if (theta <= bt[n - 1]) return bl[n - 1];
  int lo = 0, hi = n - 1;                            // bt[lo] > theta > bt[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (bt[mid] > theta) lo = mid; else hi = mid;
  }
  const double w = bt[lo] - bt[hi];
  return w > 0.0 ? bl[lo] + (bl[hi] - bl[lo]) * ((bt[lo] - theta) / w) : bl[hi];
}

// theta_v with sum_j |k_j(theta_v)|_1 = t_v: bisection (geometric while the lower end is positive) on the decreasing piecewise
// linear sum; one workgroup per value; theta_out[v] = 0 when even theta_lo does not reach the budget
__global__ __launch_bounds__(256) void kp_lasso_path_theta_kernel(const char* __restrict__ arena, PathLayout L, int ncols, const double* __restrict__ t, double theta_lo,
                                                                  double theta_hi, double* __restrict__ theta_out) {
  __shared__ double red[4];
  const double tv = t[blockIdx.x];
  auto total = [&](double th) -> double {
    double s = 0.0;
    for (int j = threadIdx.x; j < ncols; j += 256) {
      const char* base = arena + (size_t)j * L.stride;
      const PathHdr* h = reinterpret_cast<const PathHdr*>(base);
      s += path_l1_at(reinterpret_cast<const double*>(base + L.off_bpt), reinterpret_cast<const double*>(base + L.off_bpl), h->nbp, th);
    }
    s = wsum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  double lo = theta_lo, hi = theta_hi;               // total(lo) >= tv > total(hi) = 0
  if (total(lo) < tv) { if (threadIdx.x == 0) theta_out[blockIdx.x] = -1.0; return; }
  for (int it = 0; it < 220; ++it) {
    const double mid = lo > 0.0 ? sqrt(lo) * sqrt(hi) : hi * 0x1p-24;
    if (!(mid > lo && mid < hi)) break;
    if (total(mid) >= tv) lo = mid; else hi = mid;
  }
  // inside the last bracket the sum is linear unless a breakpoint lies in it: one secant step
  const double flo = total(lo), fhi = total(hi);
  double th = lo;
  if (flo > fhi) th = lo + (hi - lo) * ((flo - tv) / (flo - fhi));
  th = fmin(fmax(th, lo), hi);
  if (threadIdx.x == 0) theta_out[blockIdx.x] = th;
}

// The column states of an arena in the LDS layout (inverse with leading dimension Ls.ldm) copied into the layout of the
// global-memory kernel (leading dimension W): a support has outgrown the LDS-resident inverse and the walk goes on where it is.
// blockIdx.y: 0 = the walk's arena (with its breakpoint store), 1 = the state at the start of the round.
__global__ __launch_bounds__(256) void kp_lasso_path_relayout_kernel(const char* __restrict__ src0, const char* __restrict__ src1, PathLayout Ls, char* __restrict__ dst0,
                                                                    char* __restrict__ dst1, PathLayout Ld) {
  const int col = blockIdx.x, tid = threadIdx.x, W = Ls.W;
  const char* sb = (blockIdx.y ? src1 : src0) + (size_t)col * Ls.stride;
  char* db = (blockIdx.y ? dst1 : dst0) + (size_t)col * Ld.stride;
  const PathHdr h = *reinterpret_cast<const PathHdr*>(sb);
  if (tid == 0) {
    PathHdr o = h;
    if (o.status == PATH_OVERFLOW) o.status = PATH_OK;
    *reinterpret_cast<PathHdr*>(db) = o;
  }
  for (int i = tid; i < W; i += 256) {
    reinterpret_cast<double*>(db + Ld.off_k)[i] = reinterpret_cast<const double*>(sb + Ls.off_k)[i];
    reinterpret_cast<double*>(db + Ld.off_r)[i] = reinterpret_cast<const double*>(sb + Ls.off_r)[i];
    reinterpret_cast<double*>(db + Ld.off_sgn)[i] = reinterpret_cast<const double*>(sb + Ls.off_sgn)[i];
    reinterpret_cast<int*>(db + Ld.off_idx)[i] = reinterpret_cast<const int*>(sb + Ls.off_idx)[i];
  }
  const double* Ms = reinterpret_cast<const double*>(sb + Ls.off_M);
  double* Md = reinterpret_cast<double*>(db + Ld.off_M);
  for (int e = tid; e < h.cnt * h.cnt; e += 256) {
    const int q = e / h.cnt, t = e - q * h.cnt;
    Md[(size_t)q * Ld.ldm + t] = Ms[(size_t)q * Ls.ldm + t];
  }
  if (blockIdx.y == 0)
    for (int i = tid; i < h.nbp; i += 256) {
      reinterpret_cast<double*>(db + Ld.off_bpt)[i] = reinterpret_cast<const double*>(sb + Ls.off_bpt)[i];
      reinterpret_cast<double*>(db + Ld.off_bpl)[i] = reinterpret_cast<const double*>(sb + Ls.off_bpl)[i];
      reinterpret_cast<int*>(db + Ld.off_bpe)[i] = reinterpret_cast<const int*>(sb + Ls.off_bpe)[i];
    }
}
}  // namespace

// The lasso values t[0..nv) of one fit by the homotopy: K_dev[v] (W x ncols, device) <- K(theta_v).  G_dev: the Gram matrix as the
// QP uses it (PSD guard applied), C_dev: W x ncols.  stats (may be NULL): [0] steps of phase 1 over all columns, [1] largest support,
// [2] milliseconds, [3] 1 when the inverse lived in global memory.  known_active: the caller's least-squares solution exceeds every budget.
static int path_batch(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv, double* const* K_dev,
                      double* stats, bool known_active);

int kp_lasso_path_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv, double* const* K_dev,
                            double* stats, bool known_active) {
  const int rc = path_batch(ctx, G_dev, C_dev, W, ncols, t, nv, K_dev, stats, known_active);
  // the column states of the widest dictionaries with the inverse in memory are gigabytes (W = 336: 0.9 GB - kept; W = 512: 2.7 GB): above 1.5 GB not kept
  // between calls (the answers have left them: they were written to K_dev by the kernels)
  if (ctx->ws[12] && ctx->ws_bytes[12] > ((size_t)1536 << 20)) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->ws[12]);
    ctx->ws[12] = nullptr;
    ctx->ws_bytes[12] = 0;
  }
  return rc;
}

static int path_batch(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv, double* const* K_dev,
                      double* stats, bool known_active) {
  if (W > P_WMAX) return ctx->fail(KP_ERR_ARG, "kp_fit_lasso: the homotopy serves W <= 512");
  if (nv <= 0) return KP_OK;
  hipStream_t s = ctx->stream;
  const auto t_start = std::chrono::steady_clock::now();
  static const bool force_global = getenv("KP_LASSO_PATH_GLOBAL") != nullptr;
  bool mglobal = force_global;
  const int cap = 64 * W + 512;                       // steps (= breakpoints) allowed per column
  const double tmax = *std::max_element(t, t + nv);
  double* res = (double*)ctx->workspace(13, (size_t)(8 + nv * 2 + nv * 8) * 8);
  double* hres = (double*)kp_pinned_scratch(ctx, (size_t)(8 + nv * 2 + nv * 8) * 8);
  if (!res || !hres) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of memory");
  for (int attempt = 0; attempt < 2; ++attempt) {
    PathLayout L = make_layout(W, mglobal, cap + 2);
    // two copies of the column states: the walk itself (with the breakpoint store) and the state at the start of the current round,
    // from which the answers of the values that round brackets are walked.  Behind them, when a support can outgrow the LDS-resident
    // inverse (ldm < W), room for the same two in the layout of the global-memory kernel: the walk MOVES there when it happens
    const PathLayout Lg = make_layout(W, true, cap + 2);
    const bool may_move = !mglobal && L.ldm < W;
    char* arena = (char*)ctx->workspace(12, 2 * L.stride * (size_t)ncols + (may_move ? 2 * Lg.stride * (size_t)ncols : 0));
    if (!arena) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of device memory");
    char* snap = arena + L.stride * (size_t)ncols;
    char* const arena_g = snap + L.stride * (size_t)ncols;
    int tpb = mglobal ? P_TPB_GLOBAL : P_TPB_LDS;
    // G in LDS as well when it fits beside the whole inverse (W <= 96)
    static const bool no_gl = getenv("KP_LASSO_PATH_NO_GLDS") != nullptr;
    bool gl = !mglobal && !no_gl && L.ldm == W && path_lds_bytes(W, L.ldm, false, tpb) + (size_t)W * W * 8 <= (size_t)P_LDS_BYTES;
    size_t lds = path_lds_bytes(W, L.ldm, mglobal, tpb) + (gl ? (size_t)W * W * 8 : 0);
    auto launch = [&](char* ar, double stop, int init, int record, double* Kout, int adjust = 0, double from = 0.0, int polish = 0) -> int {
      if (mglobal) {
        hipLaunchKernelGGL((kp_lasso_path_kernel<true, P_TPB_GLOBAL, false>), dim3(ncols), dim3(P_TPB_GLOBAL), lds, s, G_dev, C_dev, L, ar, stop, from, cap, init, record, adjust, polish, Kout);
      } else if (gl) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kp_lasso_path_kernel<false, P_TPB_LDS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((kp_lasso_path_kernel<false, P_TPB_LDS, true>), dim3(ncols), dim3(P_TPB_LDS), lds, s, G_dev, C_dev, L, ar, stop, from, cap, init, record, adjust, polish, Kout);
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kp_lasso_path_kernel<false, P_TPB_LDS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((kp_lasso_path_kernel<false, P_TPB_LDS, false>), dim3(ncols), dim3(P_TPB_LDS), lds, s, G_dev, C_dev, L, ar, stop, from, cap, init, record, adjust, polish, Kout);
      }
      KP_HIP(ctx, hipGetLastError());
      return KP_OK;
    };
    auto summary = [&](const char* ar) -> int {      // -> hres[0..6]
      hipLaunchKernelGGL(kp_lasso_path_sum_kernel, dim3(1), dim3(256), 0, s, ar, L.stride, ncols, res);
      KP_HIP(ctx, hipMemcpyAsync(hres, res, 7 * 8, hipMemcpyDeviceToHost, s));
      KP_HIP(ctx, hipStreamSynchronize(s));
      return KP_OK;
    };
    int rc = launch(arena, INFINITY, 1, 1, nullptr);      // initialisation only: theta_stop above every start
    if (rc) return rc;
    rc = summary(arena);
    if (rc) return rc;
    const double theta_max = hres[1];
    if (!(theta_max > 0.0)) {                        // C = 0: K = 0 for every budget
      for (int v = 0; v < nv; ++v) KP_HIP(ctx, hipMemsetAsync(K_dev[v], 0, (size_t)W * ncols * 8, s));
      if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; stats[3] = mglobal ? 1.0 : 0.0; }
      return KP_OK;
    }
    double* t_dev = res + 8;
    double* th_dev = res + 8 + nv;
    double* h_t = hres + 8;
    double* h_th = hres + 8 + nv;
    std::vector<char> finished(nv, 0);
    int nfin = 0;
    double stop = theta_max, steps1 = 0, maxcnt = 0;
    bool overflow = false;
    int status = PATH_OK;
    while (nfin < nv) {
      // the state (not the breakpoint store) at the start of the round
      KP_HIP(ctx, hipMemcpy2DAsync(snap, L.stride, arena, L.stride, L.off_bpt, (size_t)ncols, hipMemcpyDeviceToDevice, s));
      stop = stop > theta_max * 1e-18 ? stop * (1.0 / 16.0) : 0.0;
      rc = launch(arena, stop, 0, 1, nullptr);
      if (rc) return rc;
      rc = summary(arena);
      if (rc) return rc;
      status = (int)hres[2]; steps1 = hres[3]; maxcnt = std::max(maxcnt, hres[5]);
      if (status == PATH_OVERFLOW && may_move && !mglobal) {
        // a support has outgrown the LDS-resident inverse: both state copies into the layout of the global-memory kernel, and on
        // from where every column stands (the entry that did not fit sits on the boundary: the next step takes it)
        hipLaunchKernelGGL(kp_lasso_path_relayout_kernel, dim3(ncols, 2), dim3(256), 0, s, arena, snap, L, arena_g, arena_g + Lg.stride * (size_t)ncols, Lg);
        KP_HIP(ctx, hipGetLastError());
        arena = arena_g;
        snap = arena_g + Lg.stride * (size_t)ncols;
        L = Lg;
        mglobal = true; gl = false; tpb = P_TPB_GLOBAL;
        lds = path_lds_bytes(W, L.ldm, true, tpb);
        rc = launch(arena, stop, 0, 1, nullptr);
        if (rc) return rc;
        rc = summary(arena);
        if (rc) return rc;
        status = (int)hres[2]; steps1 = hres[3]; maxcnt = std::max(maxcnt, hres[5]);
      }
      if (status == PATH_OVERFLOW) { overflow = true; break; }
      if (status != PATH_OK) break;
      const double l1_end = hres[0];
      // values whose budget this round reaches (all that are left once theta = 0 is reached: their constraint is inactive)
      std::vector<int> br;
      for (int v = 0; v < nv; ++v)
        if (!finished[v] && (t[v] <= l1_end || stop == 0.0)) br.push_back(v);
      if (br.empty()) continue;
      const int nb = (int)br.size();
      for (int q = 0; q < nb; ++q) h_t[q] = t[br[q]];
      KP_HIP(ctx, hipMemcpyAsync(t_dev, h_t, (size_t)nb * 8, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(kp_lasso_path_theta_kernel, dim3(nb), dim3(256), 0, s, arena, L, ncols, t_dev, stop, theta_max, th_dev);
      KP_HIP(ctx, hipMemcpyAsync(h_th, th_dev, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
      KP_HIP(ctx, hipStreamSynchronize(s));
      std::vector<double> theta(nb);
      std::vector<char> active(nb);
      for (int q = 0; q < nb; ++q) { active[q] = h_th[q] >= 0.0; theta[q] = active[q] ? h_th[q] : 0.0; }      // budget not reached at theta = 0: K(0)
      // the caller's least-squares solution exceeds these budgets, the walk's end does not reach them: at cond(G) ~ 1e13 the inverse
      // kept by rank-1 updates has lost the directions that carry the difference - no answer rather than a wrong one
      if (known_active && std::find(active.begin(), active.end(), (char)0) != active.end())
        return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy lost accuracy (cond(G) beyond ~1e12)");
      std::vector<int> order(nb);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return theta[x] > theta[y]; });
      // the answers: from the round's start to every theta_v in decreasing order.  At cond 1e10 two walks with different stops agree
      // on |K|_1 to ~1e-7 only, so the budget is met by Newton steps on the actual sums: along a segment |K(theta)|_1 is linear with
      // slope sum_j s_j'G_SS^-1 s_j - a small gap (or one backwards) by a move inside the segment, a larger one forwards by walking on
      for (int qq = 0; qq < nb; ++qq) {
        const int q = order[qq], v = br[q];
        double cur = theta[q];
        rc = launch(snap, cur, 0, 0, K_dev[v], 0, 0.0, 1);
        if (rc) return rc;
        for (int corr = 0; active[q] && corr < 16; ++corr) {
          rc = summary(snap);
          if (rc) return rc;
          if ((int)hres[2] != PATH_OK) return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy failed in its second pass");
          const double gap = t[v] - hres[0], slope = hres[6];
          if (fabs(gap) <= 1e-13 * t[v] || !(slope > 0.0)) break;
          if (corr == 15) {
            if (fabs(gap) <= 1e-9 * t[v]) break;
            return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy passes disagree on |K|_1");
          }
          const double next = cur - gap / slope;
          if (!(next > 0.0) || next == cur) break;
          // (|K(theta)|_1 is convex in theta: a Newton step forwards over breakpoints lands past the budget; backwards there is only
          // the move inside the segment - past the segment's start it leaves the path by O(gap), the accuracy of the walk itself)
          if (fabs(gap) > 1e-3 * t[v]) return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy passes disagree on |K|_1");
          if (fabs(gap) <= 1e-6 * t[v] || gap < 0.0) rc = launch(snap, next, 0, 0, K_dev[v], 1, cur, 0);
          else rc = launch(snap, next, 0, 0, K_dev[v], 0, 0.0, 1);
          if (rc) return rc;
          cur = next;
        }
        finished[v] = 1;
        ++nfin;
      }
      rc = summary(snap);
      if (rc) return rc;
      if ((int)hres[2] != PATH_OK) return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy failed in its second pass");
    }
    static const bool debug = getenv("KP_LASSO_PATH_DEBUG") != nullptr;
    if (debug) {                                     // the tail of the recorded breakpoints of the first columns
      KP_HIP(ctx, hipStreamSynchronize(s));
      for (int j = 0; j < std::min(ncols, 3); ++j) {
        PathHdr h;
        KP_HIP(ctx, hipMemcpy(&h, arena + (size_t)j * L.stride, sizeof(h), hipMemcpyDeviceToHost));
        const int nb_ = std::min(h.nbp, 64);
        std::vector<double> bt(nb_), bl(nb_);
        KP_HIP(ctx, hipMemcpy(bt.data(), arena + (size_t)j * L.stride + L.off_bpt + (size_t)(h.nbp - nb_) * 8, (size_t)nb_ * 8, hipMemcpyDeviceToHost));
        KP_HIP(ctx, hipMemcpy(bl.data(), arena + (size_t)j * L.stride + L.off_bpl + (size_t)(h.nbp - nb_) * 8, (size_t)nb_ * 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "kp_lasso_path: column %d: theta0 %.4e theta %.4e |k|_1 %.8f steps %d support %d status %d breakpoints %d; tail:", j, h.theta0, h.theta, h.l1,
                h.steps, h.cnt, h.status, h.nbp);
        std::vector<int> be(nb_);
        KP_HIP(ctx, hipMemcpy(be.data(), arena + (size_t)j * L.stride + L.off_bpe + (size_t)(h.nbp - nb_) * 4, (size_t)nb_ * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < nb_; ++i)
          fprintf(stderr, " (%.6e %.8f %s%d)", bt[i], bl[i], be[i] < 0 ? "stop" : (be[i] & 0x40000000) ? "del " : (be[i] & 1) ? "add- " : "add+ ", be[i] < 0 ? 0 : (be[i] & 0x3fffffff) >> 1);
        fprintf(stderr, "\n");
      }
    }
    if (overflow && !mglobal) { mglobal = true; continue; }      // supports beyond the LDS-resident inverse: again with M in memory
    if (status != PATH_OK)
      return ctx->fail(KP_ERR_NOT_CONVERGED, status == PATH_STEPS ? "kp_fit_lasso: homotopy step cap reached"
                                             : status == PATH_BP ? "kp_fit_lasso: homotopy breakpoint store full"
                                                                 : "kp_fit_lasso: homotopy met a singular support (dependent dictionary columns)");
    if (stats) {
      stats[0] = steps1; stats[1] = maxcnt;
      stats[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
      stats[3] = mglobal ? 1.0 : 0.0;
    }
    return KP_OK;
  }
  return ctx->fail(KP_ERR_NOT_CONVERGED, "kp_fit_lasso: homotopy support overflow");
}

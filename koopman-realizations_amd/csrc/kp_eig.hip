// Symmetric eigendecomposition on the device for the PCA step of get_econ_observables (Ksysid.m:1498: `pca` of the lifted
// snapshots - centred, principal axes = eigenvectors of the covariance).  The covariance comes from the fused Gram kernel
// (the dictionary's constant column carries the column sums), so the Ns x Nfull lifted matrix is never formed for it.
//
// Two-sided parallel cyclic Jacobi in ONE workgroup: n / 2 disjoint rotations per round (round-robin tournament
// ordering), n - 1 rounds per sweep.  S and V stay in global memory (n <= 256: at most 512 KB each, L2-resident;
// visibility between the threads of the workgroup is given by the barriers).  Latency-bound set-up work: a few ms.
#include "kp_internal.h"

#define EIG_MAXN 256

__global__ __launch_bounds__(256) void kp_jacobi_eig_kernel(double* __restrict__ S, double* __restrict__ V, int n, int max_sweeps, double tol,
                                                            int* __restrict__ sweeps_out) {
  __shared__ double cs[EIG_MAXN];          // (c, s) per pair
  __shared__ short pq[EIG_MAXN];           // (p, q) per pair
  __shared__ double offmax[256];
  __shared__ double scale_sh;
  const int tid = threadIdx.x;
  const int ne = n + (n & 1);               // players (a dummy when n is odd)
  const int np = ne / 2;
  for (int e = tid; e < n * n; e += 256) V[e] = (e % n == e / n) ? 1.0 : 0.0;
  {
    double d = 0.0;
    for (int i = tid; i < n; i += 256) d = fmax(d, fabs(S[i + (size_t)i * n]));
    offmax[tid] = d;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h) offmax[tid] = fmax(offmax[tid], offmax[tid + h]);
      __syncthreads();
    }
    if (tid == 0) scale_sh = offmax[0] > 0.0 ? offmax[0] : 1.0;
    __syncthreads();
  }
  const double small = tol * scale_sh;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    double my_off = 0.0;
    for (int r = 0; r < ne - 1; ++r) {
      // pairs of this round and their rotations
      if (tid < np) {
        int a, b;
        if (tid == 0) { a = ne - 1; b = r; }
        else { a = (r + tid) % (ne - 1); b = (r - tid + (ne - 1)) % (ne - 1); }
        int p = min(a, b), q = max(a, b);
        double c = 1.0, s = 0.0;
        if (q < n) {                                    // (a pair with the dummy player is skipped)
          const double spq = S[p + (size_t)q * n];
          my_off = fmax(my_off, fabs(spq));
          if (fabs(spq) > 1e-300) {
            const double th = (S[q + (size_t)q * n] - S[p + (size_t)p * n]) / (2.0 * spq);
            const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
          }
        } else {
          p = q = -1;
        }
        pq[2 * tid] = (short)p; pq[2 * tid + 1] = (short)q;
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
      __syncthreads();
      // S <- S J and V <- V J: columns p, q of every row
      for (int e = tid; e < n * np; e += 256) {
        const int i = e % n, k = e / n;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        if (p < 0) continue;
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double sp = S[i + (size_t)p * n], sq = S[i + (size_t)q * n];
        S[i + (size_t)p * n] = c * sp - s * sq;
        S[i + (size_t)q * n] = s * sp + c * sq;
        const double vp = V[i + (size_t)p * n], vq = V[i + (size_t)q * n];
        V[i + (size_t)p * n] = c * vp - s * vq;
        V[i + (size_t)q * n] = s * vp + c * vq;
      }
      __syncthreads();
      // S <- J' S: rows p, q of every column
      for (int e = tid; e < n * np; e += 256) {
        const int j = e % n, k = e / n;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        if (p < 0) continue;
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double sp = S[p + (size_t)j * n], sq = S[q + (size_t)j * n];
        S[p + (size_t)j * n] = c * sp - s * sq;
        S[q + (size_t)j * n] = s * sp + c * sq;
      }
      __syncthreads();
    }
    // largest off-diagonal entry met in this sweep (before its rotation): converged when negligible
    offmax[tid] = my_off;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h) offmax[tid] = fmax(offmax[tid], offmax[tid + h]);
      __syncthreads();
    }
    const double om = offmax[0];
    __syncthreads();
    if (om <= small) { ++sweep; break; }
  }
  if (tid == 0 && sweeps_out) *sweeps_out = sweep;
}

// ---- the same iteration over several workgroups (n > 40: the one-workgroup form moves 6 n doubles per rotation through
// one CU's load path and takes 88 ms at n = 220) ----------------------------------------------------------------------
// Every workgroup computes the n / 2 rotations of a round (cheap, and it makes the convergence decision identical everywhere
// without an exchange); the two-sided update S <- J' S J is applied per 2 x 2 block (pair k, pair k') in ONE pass - both
// rotations at once - so a round needs one grid barrier; V <- V J per (row, pair).  Blocks and (row, pair) items are dealt
// round-robin to all threads of the grid.  The grid barrier is a counter at agent scope with a release fence (L2
// write-back) before and an acquire fence (invalidate) after it: S and V are plain global arrays shared across XCDs.
#define EIGM_MAXN 1024
#define EIGM_NT 256

__device__ __forceinline__ bool eig_grid_barrier(unsigned* ctr, unsigned target) {
  __shared__ int ok_sh;
  __syncthreads();                                   // every thread's stores have reached L2
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    bool ok = true;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 19)) { ok = false; break; }   // ~1 s: a peer that never became resident - give up instead of hanging
    }
    __threadfence();
    ok_sh = ok ? 1 : 0;
  }
  __syncthreads();
  return ok_sh != 0;
}

__global__ __launch_bounds__(EIGM_NT) void kp_jacobi_eig_mw_kernel(double* __restrict__ S0, double* __restrict__ S1, double* __restrict__ V, int n,
                                                                  int max_sweeps, double tol, unsigned* __restrict__ ctr,
                                                                  int* __restrict__ out) {   // out[0] = sweeps, out[1] = buffer holding the result
  __shared__ double cs[EIGM_MAXN];          // (c, s) per pair
  __shared__ short pq[EIGM_MAXN];           // (p, q) per pair
  __shared__ double offmax[EIGM_NT];
  __shared__ double scale_sh;
  __shared__ int idle_sh;                   // odd n: the index that sits out this round
  const int tid = threadIdx.x;
  const int nwg = gridDim.x, wg = blockIdx.x;
  const int gtid = wg * EIGM_NT + tid, gsize = nwg * EIGM_NT;
  const int ne = n + (n & 1);               // players (a dummy when n is odd)
  const int np = ne / 2;
  for (int64_t e = gtid; e < (int64_t)n * n; e += gsize) V[e] = (e % n == e / n) ? 1.0 : 0.0;
  {
    double d = 0.0;
    for (int i = tid; i < n; i += EIGM_NT) d = fmax(d, fabs(S0[i + (size_t)i * n]));
    offmax[tid] = d;
    __syncthreads();
    for (int h = EIGM_NT / 2; h > 0; h >>= 1) {
      if (tid < h) offmax[tid] = fmax(offmax[tid], offmax[tid + h]);
      __syncthreads();
    }
    if (tid == 0) scale_sh = offmax[0] > 0.0 ? offmax[0] : 1.0;
    __syncthreads();
  }
  const double small = tol * scale_sh;
  unsigned bar = 0;
  if (!eig_grid_barrier(ctr, (bar += 1) * nwg)) return;      // V initialised everywhere
  // S ping-pongs between two buffers: a round reads `src` and writes every entry of `dst`, so that no workgroup can
  // overwrite an entry another one still has to read - one grid barrier per round
  double* src = S0;
  double* dst = S1;
  int cur = 0, sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    double my_off = 0.0;
    for (int r = 0; r < ne - 1; ++r) {
      if (tid == 0) idle_sh = -1;
      __syncthreads();
      for (int k = tid; k < np; k += EIGM_NT) {
        int a, b;
        if (k == 0) { a = ne - 1; b = r; }
        else { a = (r + k) % (ne - 1); b = (r - k + (ne - 1)) % (ne - 1); }
        int p = min(a, b), q = max(a, b);
        double c = 1.0, sn = 0.0;
        if (q < n) {
          const double spq = src[p + (size_t)q * n];
          my_off = fmax(my_off, fabs(spq));
          if (fabs(spq) > 1e-300) {
            const double th = (src[q + (size_t)q * n] - src[p + (size_t)p * n]) / (2.0 * spq);
            const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            sn = t * c;
          }
        } else {                                        // paired with the dummy player: p sits out
          idle_sh = p;
          p = q = -1;
        }
        pq[2 * k] = (short)p; pq[2 * k + 1] = (short)q;
        cs[2 * k] = c; cs[2 * k + 1] = sn;
      }
      __syncthreads();
      const int idle = idle_sh;
      // S <- J' S J, one 2 x 2 block (rows p, q; columns p', q') per item; items with the idle index: a 1 x 2 / 2 x 1 / 1 x 1 block
      for (int e = gtid; e < np * np; e += gsize) {
        const int k = e % np, k2 = e / np;
        const int p = pq[2 * k], q = pq[2 * k + 1], p2 = pq[2 * k2], q2 = pq[2 * k2 + 1];
        const double c = cs[2 * k], sn = cs[2 * k + 1], c2 = cs[2 * k2], s2 = cs[2 * k2 + 1];
        if (p >= 0 && p2 >= 0) {
          const double a = src[p + (size_t)p2 * n], b = src[p + (size_t)q2 * n], cc = src[q + (size_t)p2 * n], d = src[q + (size_t)q2 * n];
          const double a1 = c2 * a - s2 * b, b1 = s2 * a + c2 * b;            // columns: rotation of pair k2
          const double c1 = c2 * cc - s2 * d, d1 = s2 * cc + c2 * d;
          dst[p + (size_t)p2 * n] = c * a1 - sn * c1;                          // rows: rotation of pair k
          dst[q + (size_t)p2 * n] = sn * a1 + c * c1;
          dst[p + (size_t)q2 * n] = c * b1 - sn * d1;
          dst[q + (size_t)q2 * n] = sn * b1 + c * d1;
        } else if (p < 0 && p2 >= 0) {                                         // row `idle`: columns rotate only
          const double a = src[idle + (size_t)p2 * n], b = src[idle + (size_t)q2 * n];
          dst[idle + (size_t)p2 * n] = c2 * a - s2 * b;
          dst[idle + (size_t)q2 * n] = s2 * a + c2 * b;
        } else if (p >= 0 && p2 < 0) {                                         // column `idle`: rows rotate only
          const double a = src[p + (size_t)idle * n], b = src[q + (size_t)idle * n];
          dst[p + (size_t)idle * n] = c * a - sn * b;
          dst[q + (size_t)idle * n] = sn * a + c * b;
        } else {
          dst[idle + (size_t)idle * n] = src[idle + (size_t)idle * n];
        }
      }
      // V <- V J: columns p, q of every row (in place: every item owns its two entries)
      for (int e = gtid; e < n * np; e += gsize) {
        const int i = e % n, k = e / n;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        if (p < 0) continue;
        const double c = cs[2 * k], sn = cs[2 * k + 1];
        const double vp = V[i + (size_t)p * n], vq = V[i + (size_t)q * n];
        V[i + (size_t)p * n] = c * vp - sn * vq;
        V[i + (size_t)q * n] = sn * vp + c * vq;
      }
      if (!eig_grid_barrier(ctr, (bar += 1) * nwg)) return;
      double* t = src; src = dst; dst = t;
      cur ^= 1;
    }
    // largest off-diagonal entry met in this sweep (before its rotation): converged when negligible (the same in every workgroup)
    offmax[tid] = my_off;
    __syncthreads();
    for (int h = EIGM_NT / 2; h > 0; h >>= 1) {
      if (tid < h) offmax[tid] = fmax(offmax[tid], offmax[tid + h]);
      __syncthreads();
    }
    const double om = offmax[0];
    __syncthreads();
    if (om <= small) { ++sweep; break; }
  }
  if (wg == 0 && tid == 0 && out) { out[0] = sweep; out[1] = cur; }
}

extern "C" int kp_sym_eig(kp_ctx* ctx, const double* S, int n, double* V_out, double* lam_out, int* sweeps) {
  if (!ctx || !S || !V_out || !lam_out || n < 1 || n > EIGM_MAXN) return ctx ? ctx->fail(KP_ERR_ARG, "kp_sym_eig: bad argument (n <= 1024)") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const size_t bS = (size_t)n * n * 8;
  char* ws = (char*)ctx->workspace(6, 3 * bS + 256);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_sym_eig: out of device memory");
  double *dS = (double*)ws, *dS1 = (double*)(ws + bS), *dV = (double*)(ws + 2 * bS);
  int* dsw = (int*)(ws + 3 * bS);            // [0] sweeps, [1] result buffer, [2] barrier counter
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dS, S, bS, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemsetAsync(dsw, 0, 64, s));
  // off-diagonal entries below 1e-15 of the largest diagonal entry are rounding noise of the rotations themselves: a
  // threshold below that (1e-17 before) never triggers and every call ran all 30 sweeps
  static const bool one_wg = getenv("KP_EIG_ONE_WG") != nullptr;
  const bool multi = n > 40 && !one_wg && n <= EIGM_MAXN;
  if (!multi && n > EIG_MAXN) return ctx->fail(KP_ERR_ARG, "kp_sym_eig: n > 256 needs the multi-workgroup form");
  if (multi) {
    // one item per thread and round is plenty: (n / 2)^2 blocks over at most 64 workgroups (all resident: the stream is idle)
    const int np = (n + 1) / 2;
    int nwg = (np * np + EIGM_NT - 1) / EIGM_NT;
    nwg = std::max(1, std::min(nwg, 64));
    hipLaunchKernelGGL(kp_jacobi_eig_mw_kernel, dim3(nwg), dim3(EIGM_NT), 0, s, dS, dS1, dV, n, 30, 1e-15, (unsigned*)(dsw + 2), dsw);
  } else {
    hipLaunchKernelGGL(kp_jacobi_eig_kernel, dim3(1), dim3(256), 0, s, dS, dV, n, 30, 1e-15, dsw);
  }
  KP_HIP(ctx, hipGetLastError());
  int sw[2] = {0, 0};
  KP_HIP(ctx, hipMemcpyAsync(sw, dsw, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  if (multi && sw[0] == 0) {
    // a workgroup gave up waiting at the grid barrier (its peers were not all resident: a GPU shared with another process).
    // The one-workgroup form needs no peers.
    if (n > EIG_MAXN) return ctx->fail(KP_ERR_HIP, "kp_sym_eig: the workgroups of the eigensolver did not all become resident");
    KP_HIP(ctx, hipMemcpyAsync(dS, S, bS, hipMemcpyHostToDevice, s));
    KP_HIP(ctx, hipMemsetAsync(dsw, 0, 64, s));
    hipLaunchKernelGGL(kp_jacobi_eig_kernel, dim3(1), dim3(256), 0, s, dS, dV, n, 30, 1e-15, dsw);
    KP_HIP(ctx, hipGetLastError());
    KP_HIP(ctx, hipMemcpyAsync(sw, dsw, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    sw[1] = 0;
  }
  std::vector<double> Sd(bS / 8);
  KP_HIP(ctx, hipMemcpyAsync(Sd.data(), (multi && sw[1]) ? dS1 : dS, bS, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipMemcpyAsync(V_out, dV, bS, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) lam_out[i] = Sd[(size_t)i * n + i];
  if (sweeps) *sweeps = sw[0];
  return KP_OK;
}

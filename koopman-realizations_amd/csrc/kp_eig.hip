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

extern "C" int kp_sym_eig(kp_ctx* ctx, const double* S, int n, double* V_out, double* lam_out, int* sweeps) {
  if (!ctx || !S || !V_out || !lam_out || n < 1 || n > EIG_MAXN) return ctx ? ctx->fail(KP_ERR_ARG, "kp_sym_eig: bad argument (n <= 256)") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const size_t bS = (size_t)n * n * 8;
  char* ws = (char*)ctx->workspace(6, 2 * bS + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_sym_eig: out of device memory");
  double *dS = (double*)ws, *dV = (double*)(ws + bS);
  int* dsw = (int*)(ws + 2 * bS);
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dS, S, bS, hipMemcpyHostToDevice, s));
  // off-diagonal entries below 1e-15 of the largest diagonal entry are rounding noise of the rotations themselves: a
  // threshold below that (1e-17 before) never triggers and every call ran all 30 sweeps
  hipLaunchKernelGGL(kp_jacobi_eig_kernel, dim3(1), dim3(256), 0, s, dS, dV, n, 30, 1e-15, dsw);
  KP_HIP(ctx, hipGetLastError());
  std::vector<double> Sd(bS / 8);
  KP_HIP(ctx, hipMemcpyAsync(Sd.data(), dS, bS, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipMemcpyAsync(V_out, dV, bS, hipMemcpyDeviceToHost, s));
  int sw = 0;
  KP_HIP(ctx, hipMemcpyAsync(&sw, dsw, sizeof(int), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) lam_out[i] = Sd[(size_t)i * n + i];
  if (sweeps) *sweeps = sw;
  return KP_OK;
}

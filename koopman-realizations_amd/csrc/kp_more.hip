// Remaining rows of the fit path: L1-ball constrained least squares (solve_KoopmanQP),
// the M-projection of get_model computed from the Grams, and batched validation rollouts.
#include <algorithm>
#include <cmath>
#include <vector>

#include "kp_internal.h"

// ---- small dense helpers -------------------------------------------------------------------
// C(M x N) = alpha * op(A) * op(B) + beta * C, column-major; one thread per output element.
// Lanes run along the rows of C, so op(A) = A reads are coalesced and B reads broadcast.
__global__ __launch_bounds__(256) void kp_gemm_kernel(int tA, int tB, int M, int N, int K, double alpha,
                                                      const double* __restrict__ A, int lda, const double* __restrict__ B,
                                                      int ldb, double beta, double* __restrict__ C, int ldc) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  int i = (int)(e % M), j = (int)(e / M);
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    double av = tA ? A[k + (size_t)i * lda] : A[i + (size_t)k * lda];
    double bv = tB ? B[j + (size_t)k * ldb] : B[k + (size_t)j * ldb];
    s += av * bv;
  }
  double c0 = beta != 0.0 ? beta * C[i + (size_t)j * ldc] : 0.0;
  C[i + (size_t)j * ldc] = alpha * s + c0;
}

static hipError_t gemm(hipStream_t st, int tA, int tB, int M, int N, int K, double alpha, const double* A, int lda,
                       const double* B, int ldb, double beta, double* C, int ldc) {
  int64_t tot = (int64_t)M * N;
  hipLaunchKernelGGL(kp_gemm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tA, tB, M, N, K, alpha, A, lda, B, ldb,
                     beta, C, ldc);
  return hipGetLastError();
}

extern "C" int kp_fit_lasso_batch(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, const double* t, int nv,
                                  int max_iter, double tol, double* K, int* iters) {
  if (!ctx || !G || !C || !K || !t || W < 1 || ncols < 1 || nv < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_lasso_batch: bad argument") : KP_ERR_ARG;
  for (int v = 0; v < nv; ++v)
    if (!(t[v] >= 0.0)) return ctx->fail(KP_ERR_ARG, "kp_fit_lasso_batch: negative L1 budget");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  size_t bG = (size_t)W * W * 8, bC = (size_t)W * ncols * 8;
  char* ws = (char*)ctx->workspace(6, bG + (size_t)(1 + nv) * bC);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of device memory");
  double* Gd = (double*)ws;
  double* Cd = (double*)(ws + bG);
  double* Kd = (double*)(ws + bG + bC);
  std::vector<double*> dst(nv);
  for (int v = 0; v < nv; ++v) dst[v] = Kd + (size_t)v * W * ncols;
  KP_HIP(ctx, hipMemcpyAsync(Gd, G, bG, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(Cd, C, bC, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = kp_lasso_batch_dev(ctx, Gd, Cd, W, ncols, t, nv, max_iter > 0 ? max_iter : 20000, tol > 0 ? tol : 1e-10, dst.data(), iters);
  if (rc && rc != KP_ERR_NOT_CONVERGED) return rc;
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(K, Kd, (size_t)nv * bC, hipMemcpyDeviceToHost, ctx->stream));
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[3] = ms;
  return rc;
}

extern "C" int kp_fit_lasso(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, double t, int max_iter, double tol,
                            double* K, int* iters) {
  return kp_fit_lasso_batch(ctx, G, C, W, ncols, &t, 1, max_iter, tol, K, iters);
}

// ---- iterative refinement of the least-squares fit with the residual taken from the data ------------------------
// K <- K + G^-1 Px' (Py - Px K).  The reference solves Px \ Py by QR (Ksysid.m:1069); the normal equations lose
// cond(Px)^2 eps, which shows for the dictionaries WITHOUT dim_red margins (arm data, poly-2 econ: cond 2e5 -> 4e-6).
// One or two steps with the residual formed from the lifted rows themselves recover the accuracy of the QR solution.
// Px, Py are materialised here (Ns x W each; this is the accuracy path, not the throughput path) and the products are
// plain GEMMs.
__global__ void kp_axpy_kernel(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) y[e] += a * x[e];
}

// R[i][j] = sum_k Px[k][i] E[k][j] for tall Px, E (Ns x W, column-major): one workgroup per output element, the Ns-long
// dot product split over its 256 threads (coalesced column reads), fixed-order tree reduction
__global__ __launch_bounds__(256) void kp_tall_dot_kernel(const double* __restrict__ Px, const double* __restrict__ E, int64_t Ns, int W,
                                                          double* __restrict__ R) {
  const int i = blockIdx.x % W, j = blockIdx.x / W;
  const double* a = Px + (size_t)i * Ns;
  const double* b = E + (size_t)j * Ns;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < Ns; k += 256) s += a[k] * b[k];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) R[i + (size_t)j * W] = red[0];
}

extern "C" int kp_fit_refine(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, int steps, double* K) {
  if (!ctx || !basis || !snaps || !K || steps < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_refine: bad argument") : KP_ERR_ARG;
  const BasisDev& b = basis->dev;
  if (snaps->nzeta != b.nzeta || snaps->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_refine: snapshot/basis dimension mismatch");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int W = b.W;
  const int64_t Ns = snaps->Ns;
  const size_t bW = (size_t)W * W * 8, bP = (size_t)Ns * W * 8;
  char* ws = (char*)ctx->workspace(7, 3 * bP + 5 * bW);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_refine: out of device memory");
  double *Px = (double*)ws, *Py = (double*)(ws + bP), *E = (double*)(ws + 2 * bP);
  double *GC = (double*)(ws + 3 * bP), *Kd = GC + 2 * (size_t)W * W, *R = Kd + (size_t)W * W, *dK = R + (size_t)W * W;
  hipStream_t s = ctx->stream;
  ctx->reserve_cus = 0;
  int rc = kp_gram_dispatch(ctx, basis, snaps, GC);
  if (rc) return rc;
  rc = kp_lift_dev(ctx, basis, KP_LIFT_ROW, snaps->alpha, snaps->u, Ns, Px);
  if (!rc) rc = kp_lift_dev(ctx, basis, KP_LIFT_ROW, snaps->beta, snaps->u, Ns, Py);
  if (rc) return rc;
  KP_HIP(ctx, kp_snaps_release(snaps, s));
  KP_HIP(ctx, hipMemcpyAsync(Kd, K, bW, hipMemcpyHostToDevice, s));
  for (int it = 0; it < steps; ++it) {
    KP_HIP(ctx, hipMemcpyAsync(E, Py, bP, hipMemcpyDeviceToDevice, s));
    KP_HIP(ctx, gemm(s, 0, 0, (int)Ns, W, W, -1.0, Px, (int)Ns, Kd, W, 1.0, E, (int)Ns));       // E = Py - Px K
    hipLaunchKernelGGL(kp_tall_dot_kernel, dim3((unsigned)(W * W)), dim3(256), 0, s, Px, E, Ns, W, R);
    KP_HIP(ctx, hipGetLastError());
    rc = kp_chol_solve_dev(ctx, GC, R, W, W, dK);                                                 // G dK = Px' E
    if (rc) return rc;
    hipLaunchKernelGGL(kp_axpy_kernel, dim3((unsigned)(((int64_t)W * W + 255) / 256)), dim3(256), 0, s, (int64_t)W * W, 1.0, dK, Kd);
    KP_HIP(ctx, hipGetLastError());
  }
  KP_HIP(ctx, hipMemcpyAsync(K, Kd, bW, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// ---- get_model M-projection from the Grams (Ksysid.m:1206-1225) ---------------------------------
// L = [Px U] K1 with K1 = K(:,1:N)  =>  L'L = K1' G K1,  L'R = K1' C(:,1:N);  M' = (L'L) \ (L'R);
// A = K(1:N,1:N)', B = K(N+1:end,1:N)';  out.A = M A, out.B = M B.
__global__ void kp_transpose_kernel(const double* __restrict__ in, int rows, int cols, int ld, double* __restrict__ out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)rows * cols) {
    int i = (int)(e % rows), j = (int)(e / rows);
    out[j + (size_t)i * cols] = in[i + (size_t)j * ld];
  }
}

extern "C" int kp_model_project(kp_ctx* ctx, const double* K, const double* G, const double* C, int N, int m, double* A_out,
                                double* B_out, double* M_out) {
  if (!ctx || !K || !G || !C || !A_out || !B_out || !M_out || N < 1 || m < 0)
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_model_project: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int W = N + m;
  hipStream_t s = ctx->stream;
  size_t bW = (size_t)W * W * 8, bN = (size_t)N * N * 8;
  char* ws = (char*)ctx->workspace(6, 3 * bW + (size_t)W * N * 8 + 6 * bN + (size_t)N * m * 8 * 2 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_model_project: out of device memory");
  double* Kd = (double*)ws;
  double* Gd = Kd + (size_t)W * W;
  double* Cd = Gd + (size_t)W * W;
  double* T = Cd + (size_t)W * W;       // W x N : G K1
  double* LtL = T + (size_t)W * N;      // N x N
  double* LtR = LtL + (size_t)N * N;    // N x N
  double* Mt = LtR + (size_t)N * N;     // N x N : M'
  double* At = Mt + (size_t)N * N;      // N x N : A = K(1:N,1:N)'
  double* MA = At + (size_t)N * N;
  double* Md = MA + (size_t)N * N;
  double* Bt = Md + (size_t)N * N;      // N x m
  double* MB = Bt + (size_t)N * m;
  KP_HIP(ctx, hipMemcpyAsync(Kd, K, bW, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(Gd, G, bW, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(Cd, C, bW, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, gemm(s, 0, 0, W, N, W, 1.0, Gd, W, Kd, W, 0.0, T, W));       // T = G K1
  KP_HIP(ctx, gemm(s, 1, 0, N, N, W, 1.0, Kd, W, T, W, 0.0, LtL, N));      // K1' G K1
  KP_HIP(ctx, gemm(s, 1, 0, N, N, W, 1.0, Kd, W, Cd, W, 0.0, LtR, N));     // K1' C(:,1:N)
  int rc = kp_chol_solve_dev(ctx, LtL, LtR, N, N, Mt);                       // M' = (L'L) \ (L'R)
  if (rc) return rc;
  {
    int bad = 0;
    const size_t off = kp_chol_info_offset(N, N);
    KP_HIP(ctx, hipMemcpyAsync(&bad, (char*)ctx->ws[5] + off, sizeof(int), hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    if (bad) {   // L is rank deficient (a rank-deficient K): `M = L \ R` (Ksysid.m:1218) returns a basic solution; same here
      int r = 0;
      rc = kp_pivchol_solve_dev(ctx, LtL, LtR, N, N, Mt, &r);
      if (rc) return rc;
    }
  }
  int64_t nn = (int64_t)N * N;
  hipLaunchKernelGGL(kp_transpose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, Kd, N, N, W, At);   // A = K(1:N,1:N)'
  if (m > 0) {
    int64_t nm = (int64_t)m * N;
    hipLaunchKernelGGL(kp_transpose_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s, Kd + N, m, N, W, Bt);  // B = K(N+1:end,1:N)'
  }
  KP_HIP(ctx, gemm(s, 1, 0, N, N, N, 1.0, Mt, N, At, N, 0.0, MA, N));      // M A = (M')' A
  if (m > 0) KP_HIP(ctx, gemm(s, 1, 0, N, m, N, 1.0, Mt, N, Bt, N, 0.0, MB, N));
  hipLaunchKernelGGL(kp_transpose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, Mt, N, N, N, Md);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipMemcpyAsync(A_out, MA, bN, hipMemcpyDeviceToHost, s));
  if (m > 0) KP_HIP(ctx, hipMemcpyAsync(B_out, MB, (size_t)N * m * 8, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipMemcpyAsync(M_out, Md, bN, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}

// ---- batched rollouts (val_model / val_BLmodel, Ksysid.m:1678-1689, 1772-1787) -------------------
// One workgroup per model; z lives in LDS, thread r owns row r of z+.
// One workgroup per trajectory.  The recurrence is serial in t, so a step must be short: the model matrices are
// staged in LDS once (A always when N*N <= RO_STAGE doubles, B when it fits as well), the state is double buffered
// (one barrier per step; a wave-level fence when the workgroup is a single wave), and the inputs of step t+1 are
// fetched before the barrier of step t.
#define RO_STAGE 8192
#define RO_TC 256
template <bool ONE_WAVE>
__global__ __launch_bounds__(256) void kp_rollout_kernel(int bilinear, const double* __restrict__ A, const double* __restrict__ B,
                                                         int N, int m, const double* __restrict__ z0, const double* __restrict__ U,
                                                         int T, int n_out, double* __restrict__ Y, int stageA, int stageB, int tcmax) {
  extern __shared__ double sm[];
  double* zb = sm;                                   // [2][N]
  double* Ash = sm + 2 * N;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int bidx = blockIdx.x;
  const int mb = bilinear ? N * m : m;
  double* Bsh = Ash + (stageA ? N * N : 0);
  const double* Ab = A + (size_t)bidx * N * N;
  const double* Bb = B + (size_t)bidx * N * mb;
  const double* Ub = U + (size_t)bidx * T * m;
  double* Yb = Y + (size_t)bidx * T * n_out;
  if (stageA) {
    for (int e = tid; e < N * N; e += nth) Ash[e] = Ab[e];
    Ab = Ash;
  }
  if (stageB) {
    for (int e = tid; e < N * mb; e += nth) Bsh[e] = Bb[e];
    Bb = Bsh;
  }
  for (int r = tid; r < N; r += nth) zb[r] = z0[(size_t)bidx * N + r];
  auto sync = [&]() {
    if (ONE_WAVE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
  };
  sync();
  // time runs in chunks of tcmax <= RO_TC steps (as many as the LDS left by the model holds): the inputs of a chunk are staged in LDS and the outputs collected there, so no
  // global-memory latency sits inside the serial recurrence
  double* Uc = Bsh + (stageB ? N * mb : 0);          // [m][tcmax]
  double* Yc = Uc + m * tcmax;                       // [n_out][tcmax]
  for (int t0 = 0; t0 < T; t0 += tcmax) {
    const int tc = min(tcmax, T - t0);
    for (int e = tid; e < m * tc; e += nth) {
      const int i = e / tc, tt = e - i * tc;
      Uc[i * tcmax + tt] = Ub[(size_t)i * T + t0 + tt];
    }
    sync();
    for (int tt = 0; tt < tc; ++tt) {
      const int t = t0 + tt;
      const double* z = zb + (t & 1) * N;
      double* zn = zb + ((t + 1) & 1) * N;
      for (int r = tid; r < n_out; r += nth) Yc[r * tcmax + tt] = z[r];     // y = C z, C = [I 0] (Ksysid.m:1203)
      if (t == T - 1) break;
      for (int r = tid; r < N; r += nth) {
        double s = 0.0;
#pragma unroll 4
        for (int c = 0; c < N; ++c) s += Ab[r + (size_t)c * N] * z[c];
        if (bilinear) {
          for (int i = 0; i < m; ++i) {
            const double* Bi = Bb + (size_t)i * N * N;
            double q = 0.0;
#pragma unroll 4
            for (int c = 0; c < N; ++c) q += Bi[r + (size_t)c * N] * z[c];
            s += q * Uc[i * tcmax + tt];
          }
        } else {
          for (int i = 0; i < m; ++i) s += Bb[r + (size_t)i * N] * Uc[i * tcmax + tt];
        }
        zn[r] = s;
      }
      sync();
    }
    sync();   // the last step leaves the loop before its barrier: its Yc entries are read by other waves below
    for (int e = tid; e < n_out * tc; e += nth) {
      const int r = e / tc, tt = e - r * tc;
      Yb[(size_t)r * T + t0 + tt] = Yc[r * tcmax + tt];
    }
    sync();
  }
}

extern "C" int kp_rollout(kp_ctx* ctx, int model_type, int batch, const double* A, const double* B, int N, int m, const double* z0,
                          const double* U, int T, int n_out, double* Y) {
  if (!ctx || !A || !B || !z0 || !U || !Y || batch < 1 || N < 1 || m < 0 || T < 1 || n_out < 1 || n_out > N)
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_rollout: bad argument") : KP_ERR_ARG;
  if (model_type != KP_MODEL_LINEAR && model_type != KP_MODEL_BILINEAR)
    return ctx->fail(KP_ERR_ARG, "kp_rollout: linear or bilinear models only");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const int bil = model_type == KP_MODEL_BILINEAR;
  const size_t mb = bil ? (size_t)N * m : (size_t)m;
  size_t nA = (size_t)batch * N * N, nB = (size_t)batch * N * mb, nz = (size_t)batch * N, nU = (size_t)batch * T * m,
         nY = (size_t)batch * T * n_out;
  double* ws = (double*)ctx->workspace(6, (nA + nB + nz + nU + nY) * 8);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_rollout: out of device memory");
  double *dA = ws, *dB = dA + nA, *dz = dB + nB, *dU = dz + nz, *dY = dU + nU;
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dA, A, nA * 8, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dB, B, nB * 8, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dz, z0, nz * 8, hipMemcpyHostToDevice, s));
  if (nU) KP_HIP(ctx, hipMemcpyAsync(dU, U, nU * 8, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipEventRecord(ctx->ev0, s));
  {
    const int stageA = (size_t)N * N <= RO_STAGE;
    const int stageB = stageA && (size_t)N * N + (size_t)N * mb <= RO_STAGE + RO_STAGE / 2;
    const size_t lds_model = ((size_t)2 * N + (stageA ? (size_t)N * N : 0) + (stageB ? (size_t)N * mb : 0)) * 8;
    // input / output chunk: as many steps as fit beside the model in 128 KB (wide outputs get shorter chunks)
    const int tcmax = (int)std::min<size_t>(RO_TC, (128 * 1024 - lds_model) / ((size_t)(m + n_out) * 8));
    if (tcmax < 1) return ctx->fail(KP_ERR_ARG, "kp_rollout: model too large for the LDS staging");
    const size_t lds = lds_model + (size_t)(m + n_out) * tcmax * 8;
    static KpLdsCache lds_c0, lds_c1;
    KP_HIP(ctx, kp_ensure_lds(lds_c0, (const void*)kp_rollout_kernel<false>, 128 * 1024));
    KP_HIP(ctx, kp_ensure_lds(lds_c1, (const void*)kp_rollout_kernel<true>, 128 * 1024));
    if (N <= 64)
      hipLaunchKernelGGL(kp_rollout_kernel<true>, dim3(batch), dim3(64), lds, s, bil, dA, dB, N, m, dz, dU, T, n_out, dY, stageA, stageB, tcmax);
    else
      hipLaunchKernelGGL(kp_rollout_kernel<false>, dim3(batch), dim3(256), lds, s, bil, dA, dB, N, m, dz, dU, T, n_out, dY, stageA, stageB, tcmax);
  }
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, s));
  KP_HIP(ctx, hipMemcpyAsync(Y, dY, nY * 8, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[5] = ms;
  return KP_OK;
}

// ---- nonlinear rollouts (val_NLmodel, Ksysid.m:1848-1863):  zeta+ = Kf * lift.econ_full([zeta; u]) -------
// One workgroup per rollout; the lift of the current point and the nzeta x N matrix-vector product run
// inside the kernel, so a T-step validation is one launch instead of T lift calls.
__global__ __launch_bounds__(256) void kp_rollout_nl_kernel(BasisDev b, const double* __restrict__ Kf /* batch x [nzeta x N] col-major */,
                                                            const double* __restrict__ zeta0, const double* __restrict__ U, int T,
                                                            double* __restrict__ Z /* batch x [T x nzeta] col-major */) {
  extern __shared__ double sm[];
  double* v = sm;                 // nvars = nzeta + m : [zeta; u]
  double* full = v + b.nvars;     // nfull
  double* z = full + b.nfull;     // N
  const int tid = threadIdx.x, bi = blockIdx.x;
  const int nz = b.nzeta, m = b.m, N = b.N;
  const double* Kb = Kf + (size_t)bi * nz * N;
  const double* Ub = U + (size_t)bi * T * m;
  double* Zb = Z + (size_t)bi * T * nz;
  for (int i = tid; i < nz; i += 256) v[i] = zeta0[(size_t)bi * nz + i];
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    for (int i = tid; i < nz; i += 256) Zb[(size_t)i * T + t] = v[i];
    if (t == T - 1) break;
    for (int i = tid; i < m; i += 256) v[nz + i] = Ub[(size_t)i * T + t];
    __syncthreads();
    for (int c = tid; c < b.nfull; c += 256) full[c] = kp_eval_col(b, b.cols[c], v, 1);
    __syncthreads();
    for (int c = tid; c < N; c += 256) {
      double val;
      if (b.k_pcs == 0)
        val = full[c];
      else if (c < b.nvars)
        val = v[c];
      else if (c < b.nvars + b.k_pcs) {
        const double* pc = b.pcs + (size_t)(c - b.nvars) * b.nfull;
        val = 0.0;
        for (int i = 0; i < b.nfull; ++i) val += pc[i] * full[i];
      } else
        val = 1.0;
      z[c] = val;
    }
    __syncthreads();
    for (int r = tid; r < nz; r += 256) {
      double s = 0.0;
      for (int c = 0; c < N; ++c) s += Kb[r + (size_t)c * nz] * z[c];
      v[r] = s;
    }
    __syncthreads();
  }
}

extern "C" int kp_rollout_nl(kp_ctx* ctx, const kp_basis* basis, int batch, const double* Kf, const double* zeta0, const double* U, int T,
                             double* Z) {
  if (!ctx || !basis || !Kf || !zeta0 || !U || !Z || batch < 1 || T < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_rollout_nl: bad argument") : KP_ERR_ARG;
  const BasisDev& b = basis->dev;
  if (b.model_type != KP_MODEL_NONLINEAR) return ctx->fail(KP_ERR_ARG, "kp_rollout_nl: the dictionary must be of the nonlinear model type");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int nz = b.nzeta, m = b.m, N = b.N;
  size_t nK = (size_t)batch * nz * N, nz0 = (size_t)batch * nz, nU = (size_t)batch * T * m, nZ = (size_t)batch * T * nz;
  double* ws = (double*)ctx->workspace(6, (nK + nz0 + nU + nZ) * 8);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_rollout_nl: out of device memory");
  double *dK = ws, *dz = dK + nK, *dU = dz + nz0, *dZ = dU + nU;
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dK, Kf, nK * 8, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dz, zeta0, nz0 * 8, hipMemcpyHostToDevice, s));
  if (nU) KP_HIP(ctx, hipMemcpyAsync(dU, U, nU * 8, hipMemcpyHostToDevice, s));
  size_t lds = (size_t)(b.nvars + b.nfull + N) * 8;
  if (lds > 64 * 1024) return ctx->fail(KP_ERR_ARG, "kp_rollout_nl: dictionary too large");
  KP_HIP(ctx, hipEventRecord(ctx->ev0, s));
  hipLaunchKernelGGL(kp_rollout_nl_kernel, dim3(batch), dim3(256), lds, s, b, dK, dz, dU, T, dZ);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, s));
  KP_HIP(ctx, hipMemcpyAsync(Z, dZ, nZ * 8, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[5] = ms;
  return KP_OK;
}

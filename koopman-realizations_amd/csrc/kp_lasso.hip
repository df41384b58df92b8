// L1-ball constrained least squares of Ksysid.solve_KoopmanQP (Ksysid.m:1095-1176):
//     min_K 1/2 tr(K'GK) - tr(K'C)   s.t.  ||vec K||_1 <= t,      t = lasso * N (Ksysid.m:996)
// The reference hands the equivalent QP (split K = K+ - K-, one L1 row, Ksysid.m:1126-1137) to quadprog;
// here: accelerated projected gradient (FISTA with gradient restart), projection onto the L1 ball by
// Newton's method on the piecewise-linear threshold equation (Michelot's fixed point, warm-started with
// the previous iteration's threshold).  Everything runs as multi-workgroup kernels:
//   * all lasso values of a fit form one batch: G [K_1 ... K_nv] is a single wide MFMA product (v_mfma_f64_4x4x4_4b_f64;
//     G symmetric, so G K = G'K is a Gram-type product); values that are finished leave the batch (compaction)
//   * the momentum is applied algebraically: Y = K + mom (K - Kold) and G Y = GK + mom (GK - GKold), so one
//     product per iteration suffices and the restart decision of iteration i is a scalar read by iteration i+1
//   * an iteration is TWO launches: the product, and kp_lasso_project_kernel (V, threshold search, soft threshold,
//     momentum scalars; cross-workgroup sums through write-through slots, bounded spins); the separate v / newton / final
//     kernels (every reduction finished by the last workgroup to arrive) are the fallback
//   * every 10 iterations the host reads the states: a value ends when its active-set candidate (Cholesky per column on
//     the current support) satisfies every optimality condition of the QP, or when the iteration has converged
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kp_internal.h"
#include "kp_symm_gemm.h"

// C (W x nc) = G (W x W, symmetric) * X (W x nc): kp_symm_gemm.h (k-blocked, double-buffered 4x4x4-MFMA tiles; a
// stage-everything kernel for the few columns of the last running values)
static hipError_t symm_gemm(hipStream_t st, const double* G, const double* X, int W, int nc, double* C) {
  static const int variant = [] { const char* e = getenv("KP_SYMM_GEMM"); return e ? atoi(e) : 0; }();
  return kp_symm_gemm2(st, G, X, W, nc, C, variant);
}

// ------------------------------------------------------------------------------------------------
// largest eigenvalue of G (Lipschitz constant of the gradient) by power iteration: y = G v with one wave per
// row (G symmetric: row i is the contiguous column i), then one workgroup normalises.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) void kp_symv_kernel(const double* __restrict__ G, const double* __restrict__ v, int W, double* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= W) return;
  double s = 0.0;
  for (int k = lane; k < W; k += 64) s += G[k + (size_t)row * W] * v[k];
  s = wave_sum(s);
  if (lane == 0) y[row] = s;
}

__global__ __launch_bounds__(256) void kp_pw_norm_kernel(const double* __restrict__ y, int W, double* __restrict__ v, double* __restrict__ lam, int init) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  if (init) {
    for (int i = tid; i < W; i += 256) v[i] = 1.0 + 0.01 * (i % 7);
    return;
  }
  double p = 0.0;
  for (int i = tid; i < W; i += 256) p += y[i] * y[i];
  p = wave_sum(p);
  if ((tid & 63) == 0) red[tid >> 6] = p;
  __syncthreads();
  const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
  for (int i = tid; i < W; i += 256) v[i] = y[i] / nrm;
  if (tid == 0) lam[0] = nrm * 1.0001;   // slight over-estimate keeps the step safe
}

// ------------------------------------------------------------------------------------------------
// FISTA iteration kernels.  All lasso values of one fit (train_models loop, Ksysid.m:1372-1387) run as ONE batch:
// blockIdx.y = value; buffers hold nb matrices back to back, so G * [K_1 ... K_nb] is a single wide product.
// ------------------------------------------------------------------------------------------------
#define LS_NBLK 256
struct LassoState {
  double tk, mom, theta, t, invL;
  double change, kmax, tot;
  double pol_theta;                  // multiplier of the L1 row found by the active-set polish
  unsigned long long pol_res;        // bits of max |g + theta sign(k)| on the support (non-negative double)
  int done, notconv, passes, maxpasses, restarts;
  int pol_bad;                       // polish rejected: bit 0 sign flip / theta, bit 1 multiplier bound off the support, bit 2 column too dense / singular
  int pol_nchg;                      // entries that enter or leave the candidate's support in the next active-set round
  int pol_on;                        // this value takes part in the current active-set round
  unsigned counter;
  double part[LS_NBLK][3];
};

// block reduction of up to three sums; result valid in thread 0
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double (*red)[3]) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = a; red[w][1] = b; red[w][2] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    c = red[0][2] + red[1][2] + red[2][2] + red[3][2];
  }
}

// true in exactly one workgroup (of this value's row of the grid): the last one to have published its partial
__device__ __forceinline__ bool last_block(LassoState* st) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(&st->counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  // the partials of the other workgroups were released (agent scope) before their ticket; acquire before reading them
  // (the reads below are also volatile = cache-bypassing; per-XCD L2s and per-CU L1s are not coherent on their own)
  if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return is_last != 0;
}

// V = Y - (GY - C)/L with Y = K + mom (K - Kold), GY = GK + mom (GK - GKold); first Newton step from the previous threshold
__global__ __launch_bounds__(256) void kp_lasso_v_kernel(const double* __restrict__ Kc, const double* __restrict__ Ko,
                                                         const double* __restrict__ GKc, const double* __restrict__ GKo,
                                                         const double* __restrict__ C, int64_t n, double* __restrict__ V,
                                                         LassoState* __restrict__ stv) {
  __shared__ double red[4][3];
  LassoState* st = stv + blockIdx.y;
  const int64_t o = (int64_t)blockIdx.y * n;
  Kc += o; Ko += o; GKc += o; GKo += o; V += o;
  const double mom = st->mom, invL = st->invL, th0 = st->theta;
  double tot = 0.0, ss = 0.0, cc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double k = Kc[i], g = GKc[i];
    const double y = k + mom * (k - Ko[i]);
    const double gy = g + mom * (g - GKo[i]);
    const double v = y - (gy - C[i]) * invL;
    V[i] = v;
    const double a = fabs(v);
    tot += a;
    if (a > th0) { ss += a; cc += 1.0; }
  }
  block_sum3(tot, ss, cc, red);
  if (threadIdx.x == 0) { st->part[blockIdx.x][0] = tot; st->part[blockIdx.x][1] = ss; st->part[blockIdx.x][2] = cc; }
  if (last_block(st)) {
    const volatile double(*pp)[3] = st->part;
    const bool on = threadIdx.x < gridDim.x;          // LS_NBLK <= 256: one partial per thread, fixed reduction tree
    tot = on ? pp[threadIdx.x][0] : 0.0; ss = on ? pp[threadIdx.x][1] : 0.0; cc = on ? pp[threadIdx.x][2] : 0.0;
    __syncthreads();
    block_sum3(tot, ss, cc, red);
  if (threadIdx.x == 0) {
    st->tot = tot;
    st->passes = 1;
    if (tot <= st->t) { st->theta = 0.0; st->done = 1; }
    else { st->theta = cc > 0.0 ? fmax((ss - st->t) / cc, 0.0) : (tot - st->t) / (double)n; st->done = 0; }
    st->counter = 0u;
  }
  }
}

// one Newton step on f(theta) = sum max(|v| - theta, 0) - t  (Michelot: theta <- (sum_{|v|>theta} |v| - t) / #{|v|>theta});
// after the first step the sequence increases monotonically and stops exactly at the fixed point
__global__ __launch_bounds__(256) void kp_lasso_newton_kernel(const double* __restrict__ V, int64_t n, LassoState* __restrict__ stv) {
  __shared__ double red[4][3];
  LassoState* st = stv + blockIdx.y;
  if (st->done) return;
  V += (int64_t)blockIdx.y * n;
  const double th = st->theta;
  double ss = 0.0, cc = 0.0, z = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double a = fabs(V[i]);
    if (a > th) { ss += a; cc += 1.0; }
  }
  block_sum3(ss, cc, z, red);
  if (threadIdx.x == 0) { st->part[blockIdx.x][0] = ss; st->part[blockIdx.x][1] = cc; }
  if (last_block(st)) {
    const volatile double(*pp)[3] = st->part;
    const bool on = threadIdx.x < gridDim.x;
    ss = on ? pp[threadIdx.x][0] : 0.0; cc = on ? pp[threadIdx.x][1] : 0.0; z = 0.0;
    __syncthreads();
    block_sum3(ss, cc, z, red);
  if (threadIdx.x == 0) {
    const double nt = cc > 0.0 ? (ss - st->t) / cc : th;
    if (nt > th) st->theta = nt; else st->done = 1;
    st->passes += 1;
    st->counter = 0u;
  }
  }
}

// Kn = soft(V, theta); restart test <Y - Kn, Kn - K> > 0; momentum scalars for the next iteration
__global__ __launch_bounds__(256) void kp_lasso_final_kernel(const double* __restrict__ V, const double* __restrict__ Kc,
                                                             const double* __restrict__ Ko, int64_t n, double* __restrict__ Kn,
                                                             LassoState* __restrict__ stv) {
  __shared__ double red[4][3];
  LassoState* st = stv + blockIdx.y;
  const int64_t o = (int64_t)blockIdx.y * n;
  V += o; Kc += o; Ko += o; Kn += o;
  const double th = st->theta, mom = st->mom;
  double dot = 0.0, chg = 0.0, kmx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = V[i], k = Kc[i];
    const double a = fabs(v) - th;
    const double kn = a > 0.0 ? copysign(a, v) : 0.0;
    const double y = k + mom * (k - Ko[i]);
    Kn[i] = kn;
    dot += (y - kn) * (kn - k);
    chg = fmax(chg, fabs(kn - k));
    kmx = fmax(kmx, fabs(kn));
  }
  dot = wave_sum(dot); chg = wave_max(chg); kmx = wave_max(kmx);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = dot; red[w][1] = chg; red[w][2] = kmx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->part[blockIdx.x][0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    st->part[blockIdx.x][1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    st->part[blockIdx.x][2] = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
  }
  if (last_block(st)) {
    const volatile double(*pp)[3] = st->part;
    const bool on = threadIdx.x < gridDim.x;
    dot = on ? pp[threadIdx.x][0] : 0.0; chg = on ? pp[threadIdx.x][1] : 0.0; kmx = on ? pp[threadIdx.x][2] : 0.0;
    __syncthreads();
    dot = wave_sum(dot); chg = wave_max(chg); kmx = wave_max(kmx);
    if ((threadIdx.x & 63) == 0) { red[w][0] = dot; red[w][1] = chg; red[w][2] = kmx; }
    __syncthreads();
  if (threadIdx.x == 0) {
    dot = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    chg = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    kmx = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
    const double tk = st->tk;
    const bool restart = dot > 0.0;
    const double tn = restart ? 1.0 : 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
    st->mom = restart ? 0.0 : (tk - 1.0) / tn;
    st->tk = tn;
    st->change = chg;
    st->kmax = kmx;
    st->restarts += restart ? 1 : 0;
    if (!st->done) st->notconv += 1;          // the Newton passes of this iteration did not reach the fixed point
    st->maxpasses = max(st->maxpasses, st->passes);
    st->done = 0;
    st->counter = 0u;
  }
  }
}

// ------------------------------------------------------------------------------------------------
// One FISTA iteration's projection step as ONE launch: V, the threshold search, the soft threshold and the momentum
// scalars (kp_lasso_v / newton x P / final above are the multi-launch form, kept as the fallback).  A value is handled
// by wpv workgroups of 1024 threads (its slice of V is re-read from L2 every pass); they exchange their partial sums
// through per-workgroup slots in global memory: payload written with agent-scope (sc1, write-through) stores, the storing
// lane's s_waitcnt vmcnt(0), then the sequence number as the flag; readers poll the flags with agent-scope loads and read
// the payload with agent-scope loads (MI355X_MICROARCH.md, "Valid forms").  Every workgroup sums the slots in the same
// order, so all of them take the same branch at every pass.  All wpv x nba workgroups must be resident at once (the
// host sizes wpv for one workgroup per CU); every spin is bounded and a time-out is reported, never a hang.
// ------------------------------------------------------------------------------------------------
#define LP_NT 1024
#define LP_MAXW 64
#define LP_MAXPASS 40
struct LassoXchg {                   // per value
  double slot[2][LP_MAXW][4];        // [parity][workgroup][3 values + pad]
  unsigned flag[LP_MAXW];            // sequence number of the workgroup's latest publication
  unsigned timeout;
};

// sums (op 0) or maxima (op 1) of three values over the wpv workgroups of this value; result valid in every thread
__device__ __forceinline__ bool lp_exchange(double& a, double& b, double& c, int op_b_c_max, LassoXchg* xc, int w, int wpv, unsigned seq,
                                            double (*red)[3], double* res) {
  const int tid = threadIdx.x, wv = tid >> 6;
  // workgroup-level reduction first
  if (op_b_c_max) { a = wave_sum(a); b = wave_max(b); c = wave_max(c); }
  else { a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); }
  if ((tid & 63) == 0) { red[wv][0] = a; red[wv][1] = b; red[wv][2] = c; }
  __syncthreads();
  if (tid == 0) {
    double sa = 0.0, sb = op_b_c_max ? 0.0 : 0.0, sc = 0.0;
    for (int q = 0; q < LP_NT / 64; ++q) {
      sa += red[q][0];
      if (op_b_c_max) { sb = fmax(sb, red[q][1]); sc = fmax(sc, red[q][2]); }
      else { sb += red[q][1]; sc += red[q][2]; }
    }
    double* sl = xc->slot[seq & 1][w];
    __hip_atomic_store(&sl[0], sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sl[1], sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sl[2], sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&xc->flag[w], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // one poller per peer workgroup
  __shared__ int ok_sh;
  if (tid == 0) ok_sh = 1;
  __syncthreads();
  if (tid < wpv) {
    int spins = 0;
    bool ok = true;
    while ((int)(__hip_atomic_load(&xc->flag[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { ok = false; break; }
    }
    const double* sl = xc->slot[seq & 1][tid];
    red[tid][0] = __hip_atomic_load(&sl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid][1] = __hip_atomic_load(&sl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid][2] = __hip_atomic_load(&sl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ok) { ok_sh = 0; atomicAdd(&xc->timeout, 1u); }
  }
  __syncthreads();
  if (tid == 0) {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    for (int q = 0; q < wpv; ++q) {                  // fixed order: identical in every workgroup
      sa += red[q][0];
      if (op_b_c_max) { sb = fmax(sb, red[q][1]); sc = fmax(sc, red[q][2]); }
      else { sb += red[q][1]; sc += red[q][2]; }
    }
    res[0] = sa; res[1] = sb; res[2] = sc;
  }
  __syncthreads();
  a = res[0]; b = res[1]; c = res[2];
  const bool ok = ok_sh != 0;
  __syncthreads();
  return ok;
}

// EPT > 0: the workgroup's slice of V stays in REGISTERS (EPT elements per thread) between its passes - V is neither written
// to nor re-read from memory (at 42 running values a slice is 18 816 elements = 19 per thread; the Michelot passes then cost
// their exchange only).  EPT = 0: the slice goes through the V buffer (any size).
template <int EPT>
__global__ __launch_bounds__(LP_NT) void kp_lasso_project_kernel(const double* __restrict__ Kc, const double* __restrict__ Ko,
                                                                 const double* __restrict__ GKc, const double* __restrict__ GKo,
                                                                 const double* __restrict__ C, int64_t n, double* __restrict__ V,
                                                                 double* __restrict__ Kn, LassoState* __restrict__ stv,
                                                                 LassoXchg* __restrict__ xcv, unsigned seq0) {
  __shared__ double red[LP_MAXW > LP_NT / 64 ? LP_MAXW : LP_NT / 64][3];
  __shared__ double res[3];
  const int w = blockIdx.x, wpv = gridDim.x, val = blockIdx.y, tid = threadIdx.x;
  LassoState* st = stv + val;
  LassoXchg* xc = xcv + val;
  const int64_t o = (int64_t)val * n;
  Kc += o; Ko += o; GKc += o; GKo += o; V += o; Kn += o;
  const int64_t chunk = (n + wpv - 1) / wpv, i0 = (int64_t)w * chunk, i1 = min(n, i0 + chunk);
  const double mom = st->mom, invL = st->invL, tball = st->t;
  double theta = st->theta;
  unsigned seq = seq0;
  double vr[EPT > 0 ? EPT : 1];
  // V = Y - (G Y - C) / L and the first Newton step from the previous threshold
  double tot = 0.0, ss = 0.0, cc = 0.0;
  if (EPT > 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = i0 + tid + (int64_t)e * LP_NT;
      double v = 0.0;
      if (i < i1) {
        const double k = Kc[i], g = GKc[i];
        const double y = k + mom * (k - Ko[i]);
        const double gy = g + mom * (g - GKo[i]);
        v = y - (gy - C[i]) * invL;
      }
      vr[e] = v;
      const double a = fabs(v);
      tot += a;
      if (a > theta) { ss += a; cc += 1.0; }
    }
  } else {
    for (int64_t i = i0 + tid; i < i1; i += LP_NT) {
      const double k = Kc[i], g = GKc[i];
      const double y = k + mom * (k - Ko[i]);
      const double gy = g + mom * (g - GKo[i]);
      const double v = y - (gy - C[i]) * invL;
      V[i] = v;
      const double a = fabs(v);
      tot += a;
      if (a > theta) { ss += a; cc += 1.0; }
    }
  }
  bool ok = lp_exchange(tot, ss, cc, 0, xc, w, wpv, ++seq, red, res);
  bool done = false;
  int passes = 1;
  if (tot <= tball) { theta = 0.0; done = true; }
  else theta = cc > 0.0 ? fmax((ss - tball) / cc, 0.0) : (tot - tball) / (double)n;
  // Michelot passes: monotone from the first step on, stop exactly at the fixed point
  while (!done && ok && passes < LP_MAXPASS) {
    ss = 0.0; cc = 0.0;
    double z = 0.0;
    if (EPT > 0) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double a = fabs(vr[e]);                   // elements past the slice are 0: never above a threshold >= 0
        if (a > theta) { ss += a; cc += 1.0; }
      }
    } else {
      for (int64_t i = i0 + tid; i < i1; i += LP_NT) {
        const double a = fabs(V[i]);
        if (a > theta) { ss += a; cc += 1.0; }
      }
    }
    ok = lp_exchange(ss, cc, z, 0, xc, w, wpv, ++seq, red, res);
    const double nt = cc > 0.0 ? (ss - tball) / cc : theta;
    if (nt > theta) theta = nt; else done = true;
    ++passes;
  }
  // Kn = soft(V, theta); restart test <Y - Kn, Kn - K> > 0; statistics
  double dot = 0.0, chg = 0.0, kmx = 0.0;
  if (EPT > 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = i0 + tid + (int64_t)e * LP_NT;
      if (i < i1) {
        const double v = vr[e], k = Kc[i];
        const double a = fabs(v) - theta;
        const double kn = a > 0.0 ? copysign(a, v) : 0.0;
        const double y = k + mom * (k - Ko[i]);
        Kn[i] = kn;
        dot += (y - kn) * (kn - k);
        chg = fmax(chg, fabs(kn - k));
        kmx = fmax(kmx, fabs(kn));
      }
    }
  } else {
    for (int64_t i = i0 + tid; i < i1; i += LP_NT) {
      const double v = V[i], k = Kc[i];
      const double a = fabs(v) - theta;
      const double kn = a > 0.0 ? copysign(a, v) : 0.0;
      const double y = k + mom * (k - Ko[i]);
      Kn[i] = kn;
      dot += (y - kn) * (kn - k);
      chg = fmax(chg, fabs(kn - k));
      kmx = fmax(kmx, fabs(kn));
    }
  }
  ok = lp_exchange(dot, chg, kmx, 1, xc, w, wpv, ++seq, red, res) && ok;
  if (w == 0 && tid == 0) {
    const double tk = st->tk;
    const bool restart = dot > 0.0;
    const double tn = restart ? 1.0 : 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
    st->mom = restart ? 0.0 : (tk - 1.0) / tn;
    st->tk = tn;
    st->theta = theta;
    st->tot = tot;
    st->change = chg;
    st->kmax = kmx;
    st->restarts += restart ? 1 : 0;
    if (!done || !ok) st->notconv += 1;
    st->maxpasses = max(st->maxpasses, passes);
    st->passes = passes;
  }
}

// ------------------------------------------------------------------------------------------------
// Active-set polish.  FISTA identifies the support and the signs of the optimum long before it has converged; on a
// fixed support S_j / sign pattern s_j per column the KKT system of the QP (Ksysid.m:1126-1137: stationarity
// G k_j - c_j + theta s_j = 0 on S_j, one multiplier theta >= 0 for the L1 row, sum |k| = t) is LINEAR:
//     k_j = a_j - theta b_j,   G_SS a_j = c_S,   G_SS b_j = s_j,   theta = (sum_j s_j'a_j - t) / (sum_j s_j'b_j).
// The candidate is accepted only if it satisfies every optimality condition of the full problem (signs kept on the
// support, |g| <= theta off the support, theta >= 0): then it IS the optimum, to rounding - what quadprog returns.
// One wave per (value, column): nonzeros compacted in order, G_SS gathered into LDS, Cholesky + two substitutions.
// ------------------------------------------------------------------------------------------------
#define PL_CAP 64
__global__ __launch_bounds__(64) void kp_lasso_polish_cols_kernel(const double* __restrict__ G, const double* __restrict__ C,
                                                                  const double* __restrict__ Kc, int W, int ncols, int64_t n,
                                                                  LassoState* __restrict__ stv, double* __restrict__ Ah,
                                                                  double* __restrict__ Bh, double* __restrict__ pab) {
  __shared__ double A[PL_CAP][PL_CAP + 1];
  __shared__ int idx[PL_CAP];
  LassoState* st = stv + blockIdx.y;
  if (!st->pol_on) return;
  const int j = blockIdx.x, lane = threadIdx.x;
  const int64_t o = (int64_t)blockIdx.y * n + (int64_t)j * W;
  const double* k = Kc + o;
  double* ah = Ah + o;
  double* bh = Bh + o;
  double* pj = pab + ((int64_t)blockIdx.y * ncols + j) * 2;
  int cnt = 0;
  double sg = 0.0;
  for (int base = 0; base < W; base += 64) {
    const int i = base + lane;
    const double kv = i < W ? k[i] : 0.0;
    const bool nz = kv != 0.0;
    const unsigned long long mk = __ballot(nz);
    const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
    if (nz && pos < PL_CAP) idx[pos] = i;
    cnt += __popcll(mk);
    if (i < W) { ah[i] = 0.0; bh[i] = 0.0; }
  }
  if (cnt > PL_CAP) {
    if (lane == 0) { atomicOr(&st->pol_bad, 4); pj[0] = 0.0; pj[1] = 0.0; }
    return;
  }
  if (cnt == 0) {
    if (lane == 0) { pj[0] = 0.0; pj[1] = 0.0; }
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const bool act = lane < cnt;
  const int ir = act ? idx[lane] : 0;
  if (act) sg = k[ir] > 0.0 ? 1.0 : -1.0;
  double ra = act ? C[ir + (size_t)j * W] : 0.0, rb = sg;
  for (int c = 0; c < cnt; ++c) {
    const int ic = idx[c];
    if (act) A[lane][c] = G[ir + (size_t)ic * W];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  bool bad = false;
  for (int kk = 0; kk < cnt; ++kk) {                  // right-looking Cholesky, lane r owns row r
    const double d = A[kk][kk];
    if (!(d > 0.0)) { bad = true; break; }
    const double rinv = 1.0 / sqrt(d);
    double lrk = 0.0;
    if (lane > kk && act) { lrk = A[lane][kk] * rinv; A[lane][kk] = lrk; }
    if (lane == kk) A[kk][kk] = sqrt(d);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane > kk && act)
      for (int c = kk + 1; c <= lane; ++c) A[lane][c] -= lrk * A[c][kk];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  if (bad) {
    if (lane == 0) { atomicOr(&st->pol_bad, 4); pj[0] = 0.0; pj[1] = 0.0; }
    return;
  }
  for (int kk = 0; kk < cnt; ++kk) {                  // L y = rhs, both right-hand sides
    const double dk = A[kk][kk];
    const double ya = __shfl(ra, kk, 64) / dk, yb = __shfl(rb, kk, 64) / dk;
    if (lane == kk) { ra = ya; rb = yb; }
    if (lane > kk && act) { const double l = A[lane][kk]; ra -= l * ya; rb -= l * yb; }
  }
  for (int kk = cnt - 1; kk >= 0; --kk) {             // L'x = y
    const double dk = A[kk][kk];
    const double xa = __shfl(ra, kk, 64) / dk, xb = __shfl(rb, kk, 64) / dk;
    if (lane == kk) { ra = xa; rb = xb; }
    if (lane < kk) { const double l = A[kk][lane]; ra -= l * xa; rb -= l * xb; }
  }
  if (act) { ah[ir] = ra; bh[ir] = rb; }
  const double pa = wave_sum(act ? sg * ra : 0.0), pb = wave_sum(act ? sg * rb : 0.0);
  if (lane == 0) { pj[0] = pa; pj[1] = pb; }
}

// theta = (sum s'a - t) / (sum s'b) (every workgroup reduces the per-column sums in the same order), Kh = A - theta B,
// sign test against the support pattern of Kc
__global__ __launch_bounds__(256) void kp_lasso_polish_combine_kernel(const double* __restrict__ Ah, const double* __restrict__ Bh,
                                                                      const double* __restrict__ Kc, const double* __restrict__ pab,
                                                                      int ncols, int64_t n, double* __restrict__ Kh,
                                                                      LassoState* __restrict__ stv) {
  __shared__ double red[4][3];
  __shared__ double th_sh;
  LassoState* st = stv + blockIdx.y;
  if (!st->pol_on) return;
  const int64_t o = (int64_t)blockIdx.y * n;
  const double* pv = pab + (int64_t)blockIdx.y * ncols * 2;
  double sa = 0.0, sb = 0.0, z = 0.0;
  for (int j = threadIdx.x; j < ncols; j += 256) { sa += pv[2 * j]; sb += pv[2 * j + 1]; }
  block_sum3(sa, sb, z, red);
  if (threadIdx.x == 0) {
    const double th = sb > 0.0 ? (sa - st->t) / sb : -1.0;
    th_sh = th;
    if (blockIdx.x == 0) {
      st->pol_theta = th;
      if (!(th >= 0.0)) atomicOr(&st->pol_bad, 1);
    }
  }
  __syncthreads();
  const double th = th_sh;
  int flip = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double kh = Ah[o + i] - th * Bh[o + i], kc = Kc[o + i];
    Kh[o + i] = kh;
    if (kc != 0.0 && !(kh * kc > 0.0)) flip = 1;
  }
  const int any_flip = __any(flip);
  if (any_flip && (threadIdx.x & 63) == 0) atomicOr(&st->pol_bad, 1);
}

// optimality of the candidate on the FULL problem: g = G Kh - C; off the support |g| <= theta, on it g + theta s = 0.
// Also writes the pattern of the NEXT active-set round (a primal-dual active-set step of the QP): entries of the support
// whose candidate value lost the sign of the pattern leave it, entries off the support whose multiplier bound is violated
// enter it with the sign that lowers the objective; pol_nchg counts both.  `pat` and `next` may be the same array.
__global__ __launch_bounds__(256) void kp_lasso_polish_check_kernel(const double* __restrict__ GKh, const double* __restrict__ C,
                                                                    const double* pat, const double* __restrict__ Kh, int64_t n,
                                                                    LassoState* __restrict__ stv, double* next) {
  LassoState* st = stv + blockIdx.y;
  if (!st->pol_on) return;
  const int64_t o = (int64_t)blockIdx.y * n;
  const double th = st->pol_theta;
  const double lim = th * (1.0 + 1e-9);
  double ron = 0.0;
  int off_bad = 0, nchg = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double g = GKh[o + i] - C[i], kc = pat[o + i];
    double nx = kc;
    if (kc != 0.0) {
      ron = fmax(ron, fabs(g + (kc > 0.0 ? th : -th)));
      if (!(Kh[o + i] * kc > 0.0)) { nx = 0.0; ++nchg; }
    } else if (fabs(g) > lim) {
      off_bad = 1;
      nx = g > 0.0 ? -1.0 : 1.0;
      ++nchg;
    }
    next[o + i] = nx;
  }
  ron = wave_max(ron);
  const int any_off = __any(off_bad);
#pragma unroll
  for (int q = 32; q > 0; q >>= 1) nchg += __shfl_xor(nchg, q, 64);
  if ((threadIdx.x & 63) == 0) {
    if (ron > 0.0) atomicMax(&st->pol_res, (unsigned long long)__double_as_longlong(ron));
    if (any_off) atomicOr(&st->pol_bad, 2);
    if (nchg) atomicAdd(&st->pol_nchg, nchg);
  }
}

// between check blocks: clears the per-block statistics of the running values
__global__ void kp_lasso_ctl_kernel(LassoState* __restrict__ stv, int nb) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nb) return;
  LassoState* st = stv + v;
  st->notconv = 0; st->maxpasses = 0; st->pol_bad = 0; st->pol_res = 0ull; st->pol_nchg = 0; st->pol_on = 1;
}

// Device-to-device copies of a retirement pass (answers to their destinations, the last running slots into the holes) as ONE
// launch per 48 copies instead of one runtime copy command each (about 200 per 64-value grid, 3 us apiece).  The copies of a
// pass never overlap each other's sources (a slot moved into a hole has been examined and is a survivor) except through the
// ORDER of the old implementation: answer out first, then the move into that slot - kept by running answers and moves as
// separate launches.
#define LS_COPIES 48
struct LassoCopyList { const double* src[LS_COPIES]; double* dst[LS_COPIES]; long long n[LS_COPIES]; };
__global__ __launch_bounds__(256) void kp_lasso_copy_kernel(LassoCopyList l) {
  const double* __restrict__ s = l.src[blockIdx.y];
  double* __restrict__ d = l.dst[blockIdx.y];
  const long long n = l.n[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}

// what the host reads after a block / round: the head of every running value's state (everything in front of the partial-sum
// slots) and the time-out word of its exchange area, packed into 128-byte records - ONE small direct DMA into page-locked
// memory instead of two strided copies into pageable buffers
#define LS_REC 128
__global__ void kp_lasso_pack_kernel(const LassoState* __restrict__ stv, const LassoXchg* __restrict__ xcv, int nb, int head, char* __restrict__ out) {
  const int v = blockIdx.x, t = threadIdx.x;              // 32 threads x 4 bytes
  if (v >= nb) return;
  unsigned w = 0u;
  if (4 * t < head) w = ((const unsigned*)(stv + v))[t];
  else if (4 * t == LS_REC - 8) w = xcv ? xcv[v].timeout : 0u;
  ((unsigned*)(out + (size_t)v * LS_REC))[t] = w;
}

// between the active-set rounds of one check: the values in `on` go on, their round statistics are cleared
__global__ void kp_lasso_round_kernel(LassoState* __restrict__ stv, int nb, const int* __restrict__ on) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nb) return;
  LassoState* st = stv + v;
  st->pol_on = on[v];
  if (on[v]) { st->pol_bad = 0; st->pol_res = 0ull; st->pol_nchg = 0; }
}

// Deterministic two-stage sums: KP_RED_PARTS workgroups write one partial each (SQ: squares, else absolute values), one wave
// adds them in a fixed order.  (Round 2 summed the 1e5-element arrays of a lasso fit in ONE workgroup: 0.10 / 0.03 ms each.)
#define KP_RED_PARTS 64
template <bool SQ>
__global__ __launch_bounds__(256) void kp_reduce_partial_kernel(const double* __restrict__ A, int64_t n, double* __restrict__ part) {
  __shared__ double r[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)KP_RED_PARTS * 256) {
    const double a = A[i];
    s += SQ ? a * a : fabs(a);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = r[0] + r[1] + r[2] + r[3];
}
__global__ __launch_bounds__(64) void kp_reduce_final_kernel(const double* __restrict__ part, double* __restrict__ out) {
  const double s = wave_sum(part[threadIdx.x]);
  if (threadIdx.x == 0) out[0] = s;
}

__global__ void kp_lasso_prep_pack_kernel(const int* __restrict__ info, const double* __restrict__ l1, double* __restrict__ out) {
  out[0] = l1[0];
  out[1] = info[0] != 0 ? 1.0 : 0.0;
}
__global__ void kp_add_diag_if_kernel(double* G, int W, double v, const int* __restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W && flag[0] != 0) G[i + (size_t)i * W] += v;
}

// A *= 1 / sqrt(sumsq[0])
__global__ __launch_bounds__(256) void kp_scale_inv_kernel(double* __restrict__ A, int64_t n2, const double* __restrict__ sumsq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double inv = 1.0 / sqrt(sumsq[0]);
  if (i < n2) A[i] *= inv;
}
// out = |G|_F |A_K|_F^(1 / 2^K) (1 + 1e-12) from the two sums of squares; an underflowed |A_K|_F leaves the Frobenius bound
__global__ void kp_exp_kernel(const double* __restrict__ ss0, const double* __restrict__ ssK, double inv_pow, double* __restrict__ out) {
  const double f = sqrt(ss0[0]);
  out[0] = (ssK[0] > 0.0 ? f * exp(0.5 * inv_pow * log(ssK[0])) : f) * (1.0 + 1e-12);
}

// (source, destination) pairs of n doubles each, LS_COPIES per launch
static int lasso_copies(kp_ctx* ctx, hipStream_t s, std::vector<std::pair<const double*, double*>>& v, int64_t n) {
  for (size_t i0 = 0; i0 < v.size(); i0 += LS_COPIES) {
    LassoCopyList l;
    const int cnt = (int)std::min<size_t>(LS_COPIES, v.size() - i0);
    for (int k = 0; k < cnt; ++k) { l.src[k] = v[i0 + k].first; l.dst[k] = v[i0 + k].second; l.n[k] = (long long)n; }
    hipLaunchKernelGGL(kp_lasso_copy_kernel, dim3((unsigned)std::min<int64_t>(64, (n + 255) / 256), cnt), dim3(256), 0, s, l);
    KP_HIP(ctx, hipGetLastError());
  }
  v.clear();
  return KP_OK;
}

// (kp_fit.hip) smallest pivot of the factorisation at the head of workspace 5 relative to its diagonal entry: ~1 / cond(G)
__global__ void kp_pivot_ratio_kernel(const double* __restrict__ Lp, int n, const double* __restrict__ G, int W, double* __restrict__ out,
                                      const int* __restrict__ info, double* host_out);

// Least-squares solution + its L1 norm, PSD guard and Lipschitz constant: shared by all lasso values of one fit.
int kp_lasso_prepare(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, kp_lasso_prep* p) {
  const int64_t n = (int64_t)W * ncols;
  const size_t bK = (size_t)n * 8, bG = (size_t)W * W * 8;
  char* ws = (char*)ctx->workspace(3, bK + 3 * bG + (size_t)2 * W * 8 + 1024);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of device memory");
  p->Kls = (double*)ws;
  p->Gw = (double*)(ws + bK);
  double* Pa = (double*)(ws + bK + bG);          // two W x W buffers of the squaring sequence below
  double* Pb = (double*)(ws + bK + 2 * bG);
  double* vec = (double*)(ws + bK + 3 * bG);
  double* yv = vec + W;
  double* scal = yv + W;
  hipStream_t s = ctx->stream;
  // least-squares solution; if it satisfies the constraint it is the answer (the QP of Ksysid.m:1126-1137
  // then has an inactive L1 row)
  KP_HIP(ctx, hipMemcpyAsync(p->Gw, G_dev, bG, hipMemcpyDeviceToDevice, s));
  int rc = kp_chol_solve_dev(ctx, p->Gw, const_cast<double*>(C_dev), W, ncols, p->Kls);
  if (rc) return rc;
  double* part = scal + 8;                        // KP_RED_PARTS partial sums
  hipLaunchKernelGGL(kp_reduce_partial_kernel<false>, dim3(KP_RED_PARTS), dim3(256), 0, s, p->Kls, n, part);
  hipLaunchKernelGGL(kp_reduce_final_kernel, dim3(1), dim3(64), 0, s, part, scal);
  const int* info_dev = (const int*)((char*)ctx->ws[5] + kp_chol_info_offset(W, ncols));
  hipLaunchKernelGGL(kp_lasso_prep_pack_kernel, dim3(1), dim3(1), 0, s, info_dev, scal, scal + 3);   // scal[3] = |K_LS|_1, scal[4] = info
  // scal[5] = min_i L_ii^2 / G_ii ~ 1 / cond(G): tells the batch whether the projected-gradient iteration (O(sqrt(cond)) steps) has a chance
  hipLaunchKernelGGL(kp_pivot_ratio_kernel, dim3(1), dim3(256), 0, s, (const double*)ctx->ws[5], (W + 15) / 16 * 16, G_dev, W, scal + 5, info_dev, (double*)nullptr);
  // PSD guard of Ksysid.m:1117-1120: a non-PD Gram gets 1e-6 on the diagonal (decided on the device: no host round trip here)
  hipLaunchKernelGGL(kp_add_diag_if_kernel, dim3((W + 255) / 256), dim3(256), 0, s, p->Gw, W, 1e-6, info_dev);
  // Lipschitz constant of the gradient = lambda_max(G).  Round 2: 60 power iterations (120 launches, 0.7 ms) whose estimate
  // approaches lambda_max from BELOW.  Now an UPPER bound by repeated squaring on the matrix pipe: A_0 = G / |G|_F,
  // A_k = A_(k-1)^2 (powers of a symmetric matrix stay symmetric: the product kernel's premise);
  // lambda_max(A_0)^(2^K) = lambda_max(A_K) <= |A_K|_F, so lambda_max(G) <= |G|_F |A_K|_F^(1 / 2^K), which exceeds lambda_max
  // by at most the factor rank^(1 / 2^(K+1)) (K = 6: <= 1.05 at W = 336, a few 1e-3 for the spectra met here).  No
  // normalisation between the squarings: lambda_max(A_0) >= W^(-1/2), so |A_6|_F >= W^(-32) - far inside the f64 range.
  static const bool power_it = getenv("KP_LASSO_POWER_IT") != nullptr;
  if (power_it) {
    hipLaunchKernelGGL(kp_pw_norm_kernel, dim3(1), dim3(256), 0, s, yv, W, vec, scal, 1);
    for (int it = 0; it < 60; ++it) {
      hipLaunchKernelGGL(kp_symv_kernel, dim3((W + 3) / 4), dim3(256), 0, s, p->Gw, vec, W, yv);
      hipLaunchKernelGGL(kp_pw_norm_kernel, dim3(1), dim3(256), 0, s, yv, W, vec, scal, 0);
    }
  } else {
    const int64_t n2 = (int64_t)W * W;
    const unsigned nblk = (unsigned)((n2 + 255) / 256);
    KP_HIP(ctx, hipMemcpyAsync(Pa, p->Gw, bG, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(kp_reduce_partial_kernel<true>, dim3(KP_RED_PARTS), dim3(256), 0, s, Pa, n2, part);
    hipLaunchKernelGGL(kp_reduce_final_kernel, dim3(1), dim3(64), 0, s, part, scal + 1);
    hipLaunchKernelGGL(kp_scale_inv_kernel, dim3(nblk), dim3(256), 0, s, Pa, n2, scal + 1);
    const int K2 = 6;
    for (int k = 0; k < K2; ++k) {
      KP_HIP(ctx, symm_gemm(s, Pa, Pa, W, W, Pb));
      std::swap(Pa, Pb);
    }
    hipLaunchKernelGGL(kp_reduce_partial_kernel<true>, dim3(KP_RED_PARTS), dim3(256), 0, s, Pa, n2, part);
    hipLaunchKernelGGL(kp_reduce_final_kernel, dim3(1), dim3(64), 0, s, part, scal + 2);
    hipLaunchKernelGGL(kp_exp_kernel, dim3(1), dim3(1), 0, s, scal + 1, scal + 2, 1.0 / (double)(1 << K2), scal);
  }
  KP_HIP(ctx, hipGetLastError());
  // L, -, -, |K_LS|_1, info: one DMA (into the context's page-locked words when it has them), one synchronisation
  double back[6];
  double* dst = ctx->pin_small ? ctx->pin_small : back;
  KP_HIP(ctx, hipMemcpyAsync(dst, scal, sizeof(back), hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  p->L = dst[0]; p->l1_ls = dst[3]; p->bad = dst[4] != 0.0; p->pivot_ratio = dst[5];
  if (!(p->L > 0.0)) return ctx->fail(KP_ERR_ARG, "kp_fit_lasso: Gram matrix is zero");
  if (p->bad) {
    p->guarded = true;
    // The factorisation broke down and Gw now carries the 1e-6 guard: the QP the reference solves from here on (Ksysid.m:1117-1137) is
    // the one of the GUARDED matrix, which is positive definite - its least-squares solution decides which budgets are inactive
    // (they get exactly that solution) and starts the others.  (L was computed on the guarded matrix already.)
    rc = kp_chol_solve_dev(ctx, p->Gw, const_cast<double*>(C_dev), W, ncols, p->Kls);
    if (rc) return rc;
    hipLaunchKernelGGL(kp_reduce_partial_kernel<false>, dim3(KP_RED_PARTS), dim3(256), 0, s, p->Kls, n, part);
    hipLaunchKernelGGL(kp_reduce_final_kernel, dim3(1), dim3(64), 0, s, part, scal);
    hipLaunchKernelGGL(kp_lasso_prep_pack_kernel, dim3(1), dim3(1), 0, s, info_dev, scal, scal + 3);
    KP_HIP(ctx, hipGetLastError());
    KP_HIP(ctx, hipMemcpyAsync(dst, scal, sizeof(back), hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    if (dst[4] == 0.0) { p->l1_ls = dst[3]; p->bad = 0; }
  }
  p->ready = true;
  return KP_OK;
}

// All lasso values of one fit as a batch.  t[v]: L1 budgets; K_dev[v]: device destinations (W x ncols each);
// iters[v] (may be NULL): FISTA iterations spent on value v (0: the least-squares solution satisfies the constraint).
int kp_lasso_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, const double* t, int nv,
                       int max_iter, double tol, double* const* K_dev, int* iters, kp_lasso_prep* prep) {
  kp_lasso_prep local;
  if (!prep) prep = &local;
  if (!prep->ready) {
    int rc = kp_lasso_prepare(ctx, G_dev, C_dev, W, ncols, prep);
    if (rc) return rc;
  }
  const int64_t n = (int64_t)W * ncols;
  const size_t bK = (size_t)n * 8;
  hipStream_t s = ctx->stream;
  std::vector<int> act;                              // values whose L1 constraint is active
  {
    std::vector<std::pair<const double*, double*>> ls;      // the others get the least-squares solution
    for (int v = 0; v < nv; ++v) {
      if (iters) iters[v] = 0;
      if (!prep->bad && prep->l1_ls <= t[v]) ls.push_back({prep->Kls, K_dev[v]});
      else act.push_back(v);
    }
    const int rc = lasso_copies(ctx, s, ls, n);
    if (rc) return rc;
  }
  const int nb = (int)act.size();
  if (nb == 0) return KP_OK;
  // buffers, nb matrices each: K x3 (old, current, new), GK x2, V, polish A / B / G Kh; per-column sums; states
  const size_t bB = (size_t)nb * bK;
  const size_t b_pab = (size_t)nb * ncols * 2 * 8, b_st = (size_t)nb * sizeof(LassoState), b_xc = (size_t)nb * sizeof(LassoXchg);
  char* ws = (char*)ctx->workspace(7, 9 * bB + b_pab + b_st + b_xc + (size_t)nb * 4 + (size_t)nb * LS_REC + 1024 + 256);
  char* hrec = (char*)kp_pinned_scratch(ctx, (size_t)nb * LS_REC + (size_t)nb * 4);   // page-locked: records, then the round's `on` flags
  if (!hrec) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of page-locked host memory");
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of device memory");
  double* Kb[3] = {(double*)ws, (double*)(ws + bB), (double*)(ws + 2 * bB)};
  double* GKb[2] = {(double*)(ws + 3 * bB), (double*)(ws + 4 * bB)};
  double* V = (double*)(ws + 5 * bB);
  double* Ah = (double*)(ws + 6 * bB);
  double* Bh = (double*)(ws + 7 * bB);
  double* GKh = (double*)(ws + 8 * bB);
  double* pab = (double*)(ws + 9 * bB);
  LassoState* st = (LassoState*)(ws + 9 * bB + b_pab);
  LassoXchg* xchg = (LassoXchg*)(ws + 9 * bB + b_pab + ((b_st + 255) & ~(size_t)255));
  KP_HIP(ctx, hipMemsetAsync(xchg, 0, b_xc, s));
  // one launch per projection (kp_lasso_project_kernel) unless disabled; the multi-launch kernels are the fallback after
  // a reported time-out of its cross-workgroup exchange
  static const bool fused_env = getenv("KP_LASSO_NO_FUSED") == nullptr;
  bool fused = fused_env;
  unsigned seq = 0;
  KP_HIP(ctx, hipMemsetAsync(ws, 0, 5 * bB, s));                  // FISTA from K = 0 ...
  // ... unless the least-squares solution is usable: with old = current = K_LS and both products = G K_LS = C the first
  // step has zero gradient and zero momentum, so its projection IS proj_{L1 <= t}(K_LS) - the soft-thresholded LS solution,
  // whose support is close to the answer's for budgets near |K_LS|_1 and no worse than zero for small ones
  static const bool cold_start = getenv("KP_LASSO_COLD_START") != nullptr;
  if (!cold_start && !prep->bad) {
    std::vector<std::pair<const double*, double*>> init;
    for (int v = 0; v < nb; ++v) {
      init.push_back({prep->Kls, (double*)ws + (size_t)v * n});                       // K old
      init.push_back({prep->Kls, (double*)(ws + bB) + (size_t)v * n});                // K current
      init.push_back({C_dev, (double*)(ws + 3 * bB) + (size_t)v * n});                // G K old
      init.push_back({C_dev, (double*)(ws + 4 * bB) + (size_t)v * n});                // G K current
    }
    const int rc = lasso_copies(ctx, s, init, n);
    if (rc) return rc;
  }
  const size_t head = offsetof(LassoState, part);
  std::vector<char> hbuf(b_st, 0);
  auto hs = [&](int v) -> LassoState& { return *reinterpret_cast<LassoState*>(hbuf.data() + (size_t)v * sizeof(LassoState)); };
  for (int v = 0; v < nb; ++v) {
    LassoState& h = hs(v);
    h.tk = 1.0; h.mom = 0.0; h.theta = 0.0; h.t = t[act[v]]; h.invL = 1.0 / prep->L; h.change = 1e300; h.pol_on = 1;
  }
  KP_HIP(ctx, hipMemcpyAsync(st, hbuf.data(), b_st, hipMemcpyHostToDevice, s));
  const int nblk = (int)std::min<int64_t>(LS_NBLK, (n + 255) / 256);
  int ko = 0, kc = 1, kn = 2, go = 0, gc = 1;     // roles of the buffers
  int P = 24;                                      // Newton passes launched per iteration (adapted every block)
  int it = 0;
  static const int check_every = [] { const char* e = getenv("KP_LASSO_CHECK"); return e ? std::max(1, atoi(e)) : 10; }();
  static const bool polish = getenv("KP_LASSO_NO_POLISH") == nullptr;
  static const int first_check = [] { const char* e = getenv("KP_LASSO_FIRST_CHECK"); return e ? std::max(1, atoi(e)) : (getenv("KP_LASSO_COLD_START") ? check_every : 4); }();
  static const int as_rounds = [] { const char* e = getenv("KP_LASSO_ROUNDS"); return e ? std::max(0, atoi(e)) : 8; }();
  const int as_cap = (int)std::max<int64_t>(64, n / 8);     // more exchanges than this: the iterate is not near the optimum yet
  int* on_dev = (int*)((char*)xchg + b_xc);
  char* rec_dev = (char*)(((uintptr_t)(on_dev + nb) + 255) & ~(uintptr_t)255);
  int* on_host = (int*)(hrec + (size_t)nb * LS_REC);
  // Values that have their answer leave the batch: the running values occupy slots [0, nba) of every buffer (the last
  // running slot is moved into the hole), so the wide product and every grid shrink with the work that is left.
  int nba = nb;
  std::vector<int> slot_val(nb);
  for (int v = 0; v < nb; ++v) slot_val[v] = act[v];
  static const bool trace = getenv("KP_LASSO_TRACE") != nullptr;
  int gemm_timed_cols = 0;
  const auto t_start = std::chrono::steady_clock::now();
  // Values still running after `path_after` iterations go to the regularisation-path homotopy (kp_lasso_path.hip): exact in
  // a bounded number of steps whatever cond(G) is, where this iteration needs O(sqrt(cond)) of them.  KP_LASSO_PATH_AFTER=-1
  // disables it, 0 sends every active value there at once
  static const int path_env = [] { const char* e = getenv("KP_LASSO_PATH_AFTER"); return e ? atoi(e) : -2; }();
  // (default: 24 iterations - the first three checks - while every support fits the LDS-resident inverse and the path costs 3 - 13 ms, 100 beyond)
  // wider dictionaries: at once when the Gram is guarded or its factorisation's pivot ratio says cond(G) > 1e7 (no point in waiting:
  // O(sqrt(cond)) iterations), otherwise only as a late safety net - on a WELL-conditioned W = 336 problem a dense-support value
  // (budget 0.99 |K_LS|_1) converges in 284 iterations = 13 ms, and its path takes 57 ms (tools/lasso_dense_wellcond_probe.py)
  const bool ill = prep->guarded || prep->pivot_ratio < 1e-7;
  const int path_after = path_env != -2 ? path_env : ((W <= 136 || ill) ? 24 : 1000);
  bool path_tried = false;
  std::string path_err;
  ctx->timers[11] = 0.0;
  while (it < max_iter && nba > 0) {
    if (path_after >= 0 && it >= path_after && !path_tried && W <= 512) {
      path_tried = true;
      std::vector<double> tv(nba);
      std::vector<double*> dst(nba);
      for (int v = 0; v < nba; ++v) { tv[v] = t[slot_val[v]]; dst[v] = K_dev[slot_val[v]]; }
      double pst[4] = {0, 0, 0, 0};
      const int prc = kp_lasso_path_batch_dev(ctx, prep->Gw, C_dev, W, ncols, tv.data(), nba, dst.data(), pst, !prep->bad);
      if (trace) fprintf(stderr, "kp_lasso: homotopy for %d values after %d iterations: rc %d, %.0f steps, largest support %.0f, %.3f ms%s\n", nba, it, prc, pst[0],
                         pst[1], pst[2], pst[3] != 0.0 ? " (inverse in global memory)" : "");
      if (prc == KP_OK) {
        for (int v = 0; v < nba; ++v)
          if (iters) iters[slot_val[v]] = it;
        ctx->timers[11] = pst[2];
        nba = 0;
        break;
      }
      path_err = ctx->err;
      hrec = (char*)kp_pinned_scratch(ctx, (size_t)nb * LS_REC + (size_t)nb * 4);      // (the homotopy used the same scratch)
      if (!hrec) return ctx->fail(KP_ERR_HIP, "kp_fit_lasso: out of page-locked host memory");
      on_host = (int*)(hrec + (size_t)nb * LS_REC);
    }
    const dim3 grid(nblk, nba);
    // workgroups per value of the fused projection: all of them resident at once, one per CU
    const int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
    static const int wcap = [] { const char* e = getenv("KP_LASSO_WPV"); return e ? std::max(1, std::min(LP_MAXW, atoi(e))) : 32; }();
    const int wpv = std::max(1, std::min(wcap, ncu / nba));
    const int block_len = it == 0 ? first_check : check_every;
    for (int c = 0; c < block_len && it < max_iter; ++c, ++it) {
      if (fused && nba <= ncu) {
        // elements of a workgroup's slice per thread: in registers up to 24
        const int64_t ept = ((n + wpv - 1) / wpv + LP_NT - 1) / LP_NT;
        static const bool no_regs = getenv("KP_LASSO_V_IN_MEMORY") != nullptr;
#define KP_LP_LAUNCH(E) hipLaunchKernelGGL(kp_lasso_project_kernel<E>, dim3(wpv, nba), dim3(LP_NT), 0, s, Kb[kc], Kb[ko], GKb[gc], GKb[go], C_dev, n, V, \
                                           Kb[kn], st, xchg, seq)
        if (no_regs || ept > 24) KP_LP_LAUNCH(0);
        else if (ept <= 4) KP_LP_LAUNCH(4);
        else if (ept <= 8) KP_LP_LAUNCH(8);
        else if (ept <= 16) KP_LP_LAUNCH(16);
        else KP_LP_LAUNCH(24);
#undef KP_LP_LAUNCH
        seq += LP_MAXPASS + 4;
      } else {
        hipLaunchKernelGGL(kp_lasso_v_kernel, grid, dim3(256), 0, s, Kb[kc], Kb[ko], GKb[gc], GKb[go], C_dev, n, V, st);
        for (int p = 0; p < P; ++p) hipLaunchKernelGGL(kp_lasso_newton_kernel, grid, dim3(256), 0, s, V, n, st);
        hipLaunchKernelGGL(kp_lasso_final_kernel, grid, dim3(256), 0, s, V, Kb[kc], Kb[ko], n, Kb[kn], st);
      }
      // rotate: old <- current, current <- new; then the product of the new current (all values: one wide product)
      const int tmp = ko; ko = kc; kc = kn; kn = tmp;
      std::swap(go, gc);
      const bool time_it = it == 0 && ctx->evp[4] && ctx->evp[5];    // the batch's first and widest product, for the bench line
      if (time_it) KP_HIP(ctx, hipEventRecord(ctx->evp[4], s));
      KP_HIP(ctx, symm_gemm(s, prep->Gw, Kb[kc], W, nba * ncols, GKb[gc]));
      if (time_it) {
        KP_HIP(ctx, hipEventRecord(ctx->evp[5], s));
        gemm_timed_cols = nba * ncols;
      }
    }
    // candidate from a support / sign pattern (round 0: the current iterate's): Kh lands in the free "new" buffer, the
    // pattern of the next round in V (free between the iterations)
    auto polish_round = [&](const double* pat) -> int {
      const dim3 g2(nblk, nba);
      hipLaunchKernelGGL(kp_lasso_polish_cols_kernel, dim3(ncols, nba), dim3(64), 0, s, prep->Gw, C_dev, pat, W, ncols, n, st, Ah, Bh, pab);
      hipLaunchKernelGGL(kp_lasso_polish_combine_kernel, g2, dim3(256), 0, s, Ah, Bh, pat, pab, ncols, n, Kb[kn], st);
      KP_HIP(ctx, symm_gemm(s, prep->Gw, Kb[kn], W, nba * ncols, GKh));
      hipLaunchKernelGGL(kp_lasso_polish_check_kernel, g2, dim3(256), 0, s, GKh, C_dev, pat, Kb[kn], n, st, V);
      return KP_OK;
    };
    auto read_states = [&]() -> int {
      static_assert(offsetof(LassoState, part) <= LS_REC - 8, "state head does not fit the packed record");
      hipLaunchKernelGGL(kp_lasso_pack_kernel, dim3(nba), dim3(32), 0, s, st, fused ? xchg : (const LassoXchg*)nullptr, nba, (int)head, rec_dev);
      KP_HIP(ctx, hipGetLastError());
      KP_HIP(ctx, hipMemcpyAsync(hrec, rec_dev, (size_t)nba * LS_REC, hipMemcpyDeviceToHost, s));
      return KP_OK;
    };
    // after the synchronisation: records -> the host copies of the states
    auto unpack_states = [&](std::vector<unsigned>* touts) {
      for (int v = 0; v < nba; ++v) {
        memcpy(&hs(v), hrec + (size_t)v * LS_REC, head);
        if (touts) memcpy(&(*touts)[v], hrec + (size_t)v * LS_REC + LS_REC - 8, 4);
      }
    };
    auto accepted = [&](const LassoState& h) {
      const double pres = __builtin_bit_cast(double, h.pol_res);
      return polish && h.pol_bad == 0 && h.pol_theta >= 0.0 && pres <= 1e-9 * std::max(h.pol_theta, 1e-300);
    };
    std::vector<int> last_chg(nba, 1 << 29);
    // values that have their answer leave the batch (the last running slot moves into the hole: iterates, products, the
    // pattern of the active-set rounds and the state)
    std::vector<std::pair<const double*, double*>> cp_out, cp_mv;      // (source, destination), bK bytes each
    auto flush_copies = [&](std::vector<std::pair<const double*, double*>>& v) -> int { return lasso_copies(ctx, s, v, n); };
    auto retire = [&]() -> int {
      // Moves are chained through slots (the slot moved into hole v may itself be moved again when a lower slot retires), so
      // the SOURCE of every copy is resolved to where the data lies before this pass: `where[slot]` = original slot index.
      std::vector<int> where(nba);
      for (int v = 0; v < nba; ++v) where[v] = v;
      const int nba0 = nba;
      for (int v = nba - 1; v >= 0; --v) {             // downwards: the slot moved into a hole has been examined already
        const LassoState& h = hs(v);
        const double pres = __builtin_bit_cast(double, h.pol_res);
        const bool pol_ok = accepted(h);
        const bool conv = h.notconv == 0 && h.change <= tol * std::max(1.0, h.kmax);
        if (trace && nba <= 4)
          fprintf(stderr, "   value %d: polish flags %d theta %.6e residual %.3e exchanges %d | fista theta %.6e change %.3e kmax %.3e passes %d restarts %d%s\n",
                  slot_val[v], h.pol_bad, h.pol_theta, pres, h.pol_nchg, h.theta, h.change, h.kmax, h.maxpasses, h.restarts,
                  pol_ok ? " -> polished" : conv ? " -> converged" : "");
        if (!(pol_ok || conv)) continue;
        // the answer lies where slot v's data lay at the start of the pass (where[v]); Kh of a slot is never moved
        cp_out.push_back({(pol_ok ? Kb[kn] + (size_t)v * n : Kb[kc] + (size_t)where[v] * n), K_dev[slot_val[v]]});
        if (iters) iters[slot_val[v]] = it;
        const int last = nba - 1;
        if (v != last) {
          where[v] = where[last];
          memcpy(&hs(v), &hs(last), head);                 // (the device copy of the state moves with the slot's arrays below)
          slot_val[v] = slot_val[last];
          last_chg[v] = last_chg[last];
        }
        --nba;
      }
      // answers first (they read slots that the moves below overwrite), then every surviving slot that changed place - straight
      // from its original position
      int rc = flush_copies(cp_out);
      if (rc) return rc;
      for (int v = 0; v < nba; ++v)
        if (where[v] != v) {
          double* mv[5] = {Kb[ko], Kb[kc], GKb[go], GKb[gc], V};
          for (double* bsrc : mv) cp_mv.push_back({bsrc + (size_t)where[v] * n, bsrc + (size_t)v * n});
        }
      (void)nba0;
      rc = flush_copies(cp_mv);
      if (rc) return rc;
      // the states of the moved slots: sources are survivors' original slots, destinations retired ones - disjoint sets, one
      // launch (42 single hipMemcpyAsync of 100 bytes were 0.13 ms of a grid)
      static_assert(offsetof(LassoState, part) % 8 == 0, "state head is copied as doubles");
      std::vector<std::pair<const double*, double*>> cp_st;
      for (int v = 0; v < nba; ++v)
        if (where[v] != v) cp_st.push_back({(const double*)(st + where[v]), (double*)(st + v)});
      return lasso_copies(ctx, s, cp_st, (int64_t)(head / 8));
    };
    if (polish) {
      const int rc = polish_round(Kb[kc]);
      if (rc) return rc;
    }
    {
      const int rc = read_states();
      if (rc) return rc;
    }
    std::vector<unsigned> touts(nba, 0u);
    KP_HIP(ctx, hipStreamSynchronize(s));
    unpack_states(&touts);
    if (gemm_timed_cols) {
      float gms = 0;
      if (hipEventElapsedTime(&gms, ctx->evp[4], ctx->evp[5]) == hipSuccess) { ctx->timers[8] = gms; ctx->timers[9] = gemm_timed_cols; }
      gemm_timed_cols = 0;
    }
    if (trace)
      fprintf(stderr, "kp_lasso: iteration %d, %d values running, %.3f ms since start\n", it, nba,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    {
      const int rc = retire();
      if (rc) return rc;
    }
    // Active-set rounds: a candidate that fails only because a few entries have the wrong membership (sign lost on the
    // support, multiplier bound violated off it) is a few exchanges away from the optimum long before the iteration gets
    // there; the check kernel has left the exchanged pattern in V.  Rounds go on while the exchanges shrink.
    if (polish) {
      std::vector<int> on(nba);
      for (int r = 1; r <= as_rounds && nba > 0; ++r) {
        int n_on = 0;
        for (int v = 0; v < nba; ++v) {
          const LassoState& h = hs(v);
          on[v] = !(h.pol_bad & 4) && h.pol_nchg > 0 && h.pol_nchg <= as_cap && h.pol_nchg < 2 * last_chg[v] && h.pol_theta > 0.0;
          last_chg[v] = h.pol_nchg;
          n_on += on[v];
        }
        if (trace) {
          fprintf(stderr, "   round %d: %d of %d values go on; exchanges", r, n_on, nba);
          for (int v = 0; v < nba && v < 48; ++v) fprintf(stderr, " %d", hs(v).pol_nchg);
          fprintf(stderr, "\n");
        }
        if (!n_on) break;
        memcpy(on_host, on.data(), (size_t)nba * 4);
        KP_HIP(ctx, hipMemcpyAsync(on_dev, on_host, (size_t)nba * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kp_lasso_round_kernel, dim3((nba + 63) / 64), dim3(64), 0, s, st, nba, on_dev);
        int rc = polish_round(V);
        if (rc) return rc;
        rc = read_states();
        if (rc) return rc;
        KP_HIP(ctx, hipStreamSynchronize(s));
        unpack_states(nullptr);
        rc = retire();
        if (rc) return rc;
      }
    }
    for (unsigned tq : touts)
      if (tq) {                                      // an exchange of the one-launch projection timed out (workgroups not co-resident):
        fused = false;                               // continue with the multi-launch kernels, which need no residency
        KP_HIP(ctx, hipMemsetAsync(xchg, 0, b_xc, s));
        break;
      }
    bool all_exact = true;
    int maxp = 0;
    for (int v = 0; v < nba; ++v) {
      all_exact &= hs(v).notconv == 0;
      maxp = std::max(maxp, hs(v).maxpasses);
    }
    if (nba == 0) break;
    // adapt the number of Newton passes to what the values still running needed in the last block
    P = all_exact ? std::max(2, maxp + 1) : std::min(48, P + 6);
    hipLaunchKernelGGL(kp_lasso_ctl_kernel, dim3((nba + 63) / 64), dim3(64), 0, s, st, nba);
  }
  if (nba > 0) {
    for (int v = 0; v < nba; ++v) {
      KP_HIP(ctx, hipMemcpyAsync(K_dev[slot_val[v]], Kb[kc] + (size_t)v * n, bK, hipMemcpyDeviceToDevice, s));
      if (iters) iters[slot_val[v]] = it;
    }
    return ctx->fail(KP_ERR_NOT_CONVERGED, path_err.empty() ? std::string("kp_fit_lasso: iteration cap reached") : "kp_fit_lasso: iteration cap reached (" + path_err + ")");
  }
  return KP_OK;
}

int kp_lasso_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double t, int max_iter, double tol,
                 double* K_dev, int* iters, kp_lasso_prep* prep) {
  double* dst[1] = {K_dev};
  return kp_lasso_batch_dev(ctx, G_dev, C_dev, W, ncols, &t, 1, max_iter, tol, dst, iters, prep);
}

// Batched small fits: many independent systems with the SAME dictionary and W <= 16, one workgroup per system.
// This is the shape of evaluate_rand_models.m (one Ksysid fit per random system, model type and degree;
// 1-D state, W <= 16): the per-system work (lift of ~1e4 snapshot pairs, Px'Px, Px'Py, the least-squares
// solve) is far too small for a launch sequence of its own, so one launch does all systems.
//   lift            generic column evaluation (any dictionary without dimension reduction), 64 snapshots per tile in LDS
//   Px'Px, Px'Py    register-blocked: a lane owns one 4 x 4 block of G and of C for 4 of the 64 snapshots of a tile
//                   (16 blocks x 16 snapshot groups = 256 threads): three 4-wide LDS reads per 32 FMAs.  (One output
//                   element per thread needed 3 LDS reads per 2 FMAs and was LDS-bound: 0.78 ms per 1024 systems.)
//                   The 16 groups are summed in a fixed order at the end (deterministic).
//   G K = C         sb_spd_solve16: Cholesky of the 16 x 16 block in registers (one wave, pivots by v_readlane, as
//                   kp_chol_kernel's diagonal block), then forward / backward substitution, one thread per right-hand side
// Replaces, for each system: Ksysid.get_Koopman (Ksysid.m:987-1092) with lasso = Inf.
#include "kp_internal.h"
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#define SB_TS 64      // snapshots per tile
#define SB_W 16       // maximum Px width
#define SB_LD 18      // row stride of the tiles (even: 16-byte aligned rows for the 4-wide operand reads)

__device__ __forceinline__ double sb_bcast(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Solves G X = C for the 16 x 16 SPD matrix Gs (identity on the padding) and 16 right-hand sides Cs, all in LDS with
// row stride SB_LD; X -> Xs.  Cholesky in registers: lane r (< 16) of wave 0 owns row r, pivots by v_readlane (as
// kp_chol_kernel's diagonal block), then forward / backward substitution, one thread per right-hand side.
// All 256 threads call; *bad is set when a pivot is not positive.
__device__ __forceinline__ void sb_spd_solve16(const double* Gs, const double* Cs, double* Xs, double* Ls, double* Dd, int* bad) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int r = tid & 15;
    double row[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) row[cc] = Gs[r * SB_LD + cc];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
      double d = sb_bcast(row[cc], cc);
      if (!(d > 0.0)) {
        if (tid == 0) *bad = 1;
        d = 1.0;
      }
      double id = __builtin_amdgcn_rsq(d);
      id = id * (1.5 - 0.5 * d * id * id);
      id = id * (1.5 - 0.5 * d * id * id);
      if (tid == cc) Dd[cc] = id;
      const double l = row[cc] * id;
      row[cc] = l;
#pragma unroll
      for (int c2 = cc + 1; c2 < 16; ++c2) row[c2] -= l * sb_bcast(l, c2);
    }
    if (tid < 16) {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) Ls[r * SB_LD + cc] = cc <= r ? row[cc] : 0.0;
    }
  }
  __syncthreads();
  // L Y = C, L' X = Y: one thread per right-hand side
  if (tid < 16) {
    const int j = tid;
    double y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s = Cs[i * SB_LD + j];
#pragma unroll
      for (int q = 0; q < i; ++q) s -= Ls[i * SB_LD + q] * y[q];
      y[i] = s * Dd[i];
    }
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int q = i + 1; q < 16; ++q) s -= Ls[q * SB_LD + i] * y[q];
      y[i] = s * Dd[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Xs[i * SB_LD + j] = y[i];
  }
  __syncthreads();
}

// recipes != nullptr (monomial dictionaries): column c is the product of <= nfmax entries of a per-snapshot power table
// x_v^e (recipe byte = v * D + e - 1, 0xff = unused; the table kp_basis_create builds for the fused Gram kernels), so a
// lift entry costs <= nfmax LDS reads and multiplies instead of the generic evaluation (exponent rows from global
// memory, up to 13 multiplies per column for the degree-13 dictionaries of the sweep).
//
// refine > 0: iterative refinement of the normal-equations solution with the residual formed FROM THE DATA,
//   R = Px' (Py - Px K),  G dK = R,  K += dK,
// in further sweeps over the snapshots (the tile E = Py - Px K replaces Py in the same accumulation).  The reference
// solves Px \ Py by QR (Ksysid.m:1069); the degree-13 dictionaries of the sweep have cond(Px) ~ 1e5, where the plain
// normal equations are only good to cond^2 eps ~ 1e-6 - one refinement step brings K to the accuracy of the QR solution.
// Device-resident trajectories of nb systems with one layout (kp_traj): per system the merged training trials
// (rows = ntrials * T, column-major rows x n / rows x m, RAW values) and the per-system scaling sc = [y offset (n) |
// y factor (n) | u offset (m) | u factor (m)] (get_scale, Ksysid.m:180-229).  With Y != nullptr the fit kernel forms
// the scaled snapshot pairs itself (get_snapshotPairs, Ksysid.m:941-978, delays = 0, all pairs in time order): pair p of
// a system is row (p / (T-1)) T + p % (T-1) and its successor - no pair across a trial seam - and the last good pair
// is dropped (:960), so a system has ntrials (T - 1) - 1 pairs.
struct TrajView {
  const double* Y;      // SCALED in place at upload (kp_traj_apply_scale_kernel): the fit passes only index
  const double* U;
  const double* sc;
  int ntrials, T, rows, n, m;
  unsigned long long div_magic;   // ceil(2^40 / (T - 1)): pair / (T - 1) = (pair * magic) >> 40 for pair < 2^24
};

__global__ __launch_bounds__(256) void kp_small_fit_kernel(BasisDev b, const double* __restrict__ alpha, const double* __restrict__ beta,
                                                           const double* __restrict__ u, int64_t Ns_total, int Ns, double* __restrict__ Kout,
                                                           double* __restrict__ Gout, double* __restrict__ Cout, int* __restrict__ status,
                                                           const uint32_t* __restrict__ recipes, int D, int nfmax, int refine, TrajView tv, int cheb) {
  extern __shared__ __align__(16) double sm[];
  // LDS: vx[nvars][TS] | vy[nvars][TS] | um[m][TS] | Px[TS][LD] | Py[TS][LD] | Gs[16][LD] | Cs[16][LD] | Ls[16][LD] | Dd[16] | Ks[16][LD] | Xs[16][LD] | pw
  const int nv = b.nvars, m = b.m, N = b.N, W = b.W;
  double* vx = sm;
  double* vy = vx + nv * SB_TS;
  double* um = vy + nv * SB_TS;
  double* Px = um + (m > 0 ? m : 1) * SB_TS;
  double* Py = Px + SB_TS * SB_LD;
  double* Gs = Py + SB_TS * SB_LD;
  double* Cs = Gs + 16 * SB_LD;
  double* Ls = Cs + 16 * SB_LD;
  double* Dd = Ls + 16 * SB_LD;
  double* Ks = Dd + 16;                             // current K, [row][col]
  double* Xs = Ks + 16 * SB_LD;                     // solution of the last solve
  double* pw = Xs + 16 * SB_LD;                     // [side][v * D + e - 1][snapshot]  (only with recipes)
  __shared__ int bad;
  __shared__ uint32_t recs[SB_W];
  if (recipes && threadIdx.x < N) recs[threadIdx.x] = recipes[threadIdx.x];
  const int tid = threadIdx.x;
  const int sys = blockIdx.x;
  const int64_t base = (int64_t)sys * Ns;           // first row of this system inside the merged arrays
  if (tid == 0) bad = 0;
  for (int e = tid; e < 2 * SB_TS * SB_LD; e += 256) Px[e] = 0.0;      // Px and Py (contiguous): unused columns stay zero
  const int gi = tid >> 4, gj = tid & 15;
  const int blk = tid & 15, grp = tid >> 4;          // 4 x 4 output block, snapshot group (4 consecutive snapshots of a tile)
  const int bi = (blk >> 2) * 4, bj = (blk & 3) * 4;
  Ks[gi * SB_LD + gj] = 0.0;
  for (int pass = 0; pass <= refine; ++pass) {
    double g[4][4], c[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) g[x][y] = c[x][y] = 0.0;
    __syncthreads();
    for (int r0 = 0; r0 < Ns; r0 += SB_TS) {
      const int nl = min(SB_TS, Ns - r0);
      // raw variables of the tile: x side (alpha [, u]), y side (beta [, u]), inputs
      for (int e = tid; e < (2 * nv + m) * SB_TS; e += 256) {
        const int v = e / SB_TS, p = e % SB_TS;
        double x = 0.0;
        if (p < nl && tv.Y) {                          // pair straight from the (already scaled) trajectories
          const int pair = r0 + p, Tm1 = tv.T - 1;
          const int tr = (int)(((unsigned long long)pair * tv.div_magic) >> 40);
          const int row = tr * tv.T + (pair - tr * Tm1);
          int var = v < nv ? v : v < 2 * nv ? v - nv : b.nzeta + (v - 2 * nv);        // index into [y (n) ; u (m)]
          const int shift = (v >= nv && v < 2 * nv && var < b.nzeta) ? 1 : 0;         // beta = the next row of y
          if (var < b.nzeta) x = tv.Y[((size_t)sys * tv.n + var) * tv.rows + row + shift];
          else x = tv.U[((size_t)sys * tv.m + (var - b.nzeta)) * tv.rows + row];
        } else if (p < nl) {
          const int64_t row = base + r0 + p;
          if (v < nv) x = v < b.nzeta ? alpha[(int64_t)v * Ns_total + row] : u[(int64_t)(v - b.nzeta) * Ns_total + row];
          else if (v < 2 * nv) x = (v - nv) < b.nzeta ? beta[(int64_t)(v - nv) * Ns_total + row] : u[(int64_t)(v - nv - b.nzeta) * Ns_total + row];
          else x = u[(int64_t)(v - 2 * nv) * Ns_total + row];
        }
        sm[e] = x;                                     // vx | vy | um are contiguous in this order
      }
      __syncthreads();
      if (recipes) {                                   // power table of both sides
        for (int e = tid; e < 2 * nv * SB_TS; e += 256) {
          const int p = e % SB_TS, sv = e / SB_TS;     // sv = side * nv + v; vx | vy are contiguous
          const double x = sm[sv * SB_TS + p];
          double q = x, qm = 1.0;
          for (int k = 0; k < D; ++k) {
            pw[(sv * D + k) * SB_TS + p] = q;
            const double qn = cheb ? 2.0 * x * q - qm : q * x;      // cheb: T_(k+2) = 2 x T_(k+1) - T_k instead of x^(k+2)
            qm = q;
            q = qn;
          }
        }
        __syncthreads();
      }
      // rows of Px / Py (Ksysid.m:1034-1064): [psi, u] / psi (x) [1; u] / psi([zeta; u])
      for (int e = tid; e < 2 * N * SB_TS; e += 256) {
        const int p = e % SB_TS, sc = e / SB_TS, side = sc / N, col = sc - side * N;
        double val;
        if (recipes) {
          const uint32_t rc = recs[col];
          val = p < nl ? 1.0 : 0.0;
          for (int f = 0; f < nfmax; ++f) {
            const uint32_t id = (rc >> (8 * f)) & 255u;
            const double t = pw[((side * nv) * D + (id == 255u ? 0u : id)) * SB_TS + p];       // unconditional read, then select
            val *= id == 255u ? 1.0 : t;
          }
        } else {
          val = p < nl ? kp_eval_col(b, b.cols[col], (side ? vy : vx) + p, SB_TS) : 0.0;
        }
        double* P = (side ? Py : Px) + p * SB_LD;
        P[col] = val;
        if (b.model_type == KP_MODEL_BILINEAR)
          for (int i = 0; i < m; ++i) P[(i + 1) * N + col] = val * um[i * SB_TS + p];
      }
      if (b.model_type == KP_MODEL_LINEAR)
        for (int e = tid; e < 2 * m * SB_TS; e += 256) {
          const int p = e % SB_TS, si = e / SB_TS, side = si / m, i = si - side * m;
          ((side ? Py : Px) + p * SB_LD)[N + i] = um[i * SB_TS + p];       // zero past the tail (um is)
        }
      __syncthreads();
      if (pass > 0) {                                  // E = Py - Px K in place of Py: thread = (snapshot, 4 columns)
        const int p = tid >> 2, j4 = (tid & 3) * 4;
        double2 e0 = *reinterpret_cast<const double2*>(Py + p * SB_LD + j4), e1 = *reinterpret_cast<const double2*>(Py + p * SB_LD + j4 + 2);
        for (int i = 0; i < W; ++i) {
          const double xi = Px[p * SB_LD + i];
          const double2 k0 = *reinterpret_cast<const double2*>(Ks + i * SB_LD + j4), k1 = *reinterpret_cast<const double2*>(Ks + i * SB_LD + j4 + 2);
          e0.x -= xi * k0.x; e0.y -= xi * k0.y; e1.x -= xi * k1.x; e1.y -= xi * k1.y;
        }
        *reinterpret_cast<double2*>(Py + p * SB_LD + j4) = e0;
        *reinterpret_cast<double2*>(Py + p * SB_LD + j4 + 2) = e1;
        __syncthreads();
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const double* xr = Px + (grp * 4 + s_) * SB_LD;
        const double* yr = Py + (grp * 4 + s_) * SB_LD;
        const double2 xa = *reinterpret_cast<const double2*>(xr + bi), xb = *reinterpret_cast<const double2*>(xr + bi + 2);
        const double2 ja = *reinterpret_cast<const double2*>(xr + bj), jb = *reinterpret_cast<const double2*>(xr + bj + 2);
        const double2 ya = *reinterpret_cast<const double2*>(yr + bj), yb = *reinterpret_cast<const double2*>(yr + bj + 2);
        const double xi[4] = {xa.x, xa.y, xb.x, xb.y}, xj[4] = {ja.x, ja.y, jb.x, jb.y}, yj[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (pass == 0) g[x][y] += xi[x] * xj[y];
            c[x][y] += xi[x] * yj[y];
          }
      }
      __syncthreads();
    }
    // sum of the 16 snapshot groups in a fixed order (pass 0: G and C; later passes: the residual R in Cs)
    if (pass == 0) Gs[gi * SB_LD + gj] = 0.0;
    Cs[gi * SB_LD + gj] = 0.0;
    for (int t = 0; t < 16; ++t) {
      __syncthreads();
      if (grp == t) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (pass == 0) Gs[(bi + x) * SB_LD + bj + y] += g[x][y];
            Cs[(bi + x) * SB_LD + bj + y] += c[x][y];
          }
      }
    }
    __syncthreads();
    if (pass == 0) {
      // G (identity on the padding so that the 16 x 16 factorisation is well defined) and C
      const double gv = Gs[gi * SB_LD + gj], cv = Cs[gi * SB_LD + gj];
      if (gi < W && gj < W) {
        if (Gout) Gout[(size_t)sys * W * W + (size_t)gj * W + gi] = gv;
        if (Cout) Cout[(size_t)sys * W * W + (size_t)gj * W + gi] = cv;
      } else {
        Gs[gi * SB_LD + gj] = gi == gj ? 1.0 : 0.0;
        Cs[gi * SB_LD + gj] = 0.0;
      }
      __syncthreads();
    }
    // G X = C (pass 0) or G dK = R: 16 x 16 Cholesky in registers + substitution
    sb_spd_solve16(Gs, Cs, Xs, Ls, Dd, &bad);
    Ks[gi * SB_LD + gj] += Xs[gi * SB_LD + gj];
    if (bad) break;                                  // (uniform: `bad` is shared and the solve ends with a barrier)
  }
  __syncthreads();
  if (gi < W && gj < W) Kout[(size_t)sys * W * W + (size_t)gj * W + gi] = bad ? __builtin_nan("") : Ks[gi * SB_LD + gj];
  if (tid == 0 && status) status[sys] = bad;
}

extern "C" int kp_fit_batch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, int nb, int64_t Ns_each, double* K_out,
                            double* G_out, double* C_out, int* status_out) {
  if (!ctx || !basis || !snaps || !K_out || nb < 1 || Ns_each < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_batch: bad argument") : KP_ERR_ARG;
  const BasisDev& b = basis->dev;
  if (snaps->nzeta != b.nzeta || snaps->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_batch: snapshot/basis dimension mismatch");
  if (snaps->Ns != (int64_t)nb * Ns_each) return ctx->fail(KP_ERR_ARG, "kp_fit_batch: snapshots must hold nb x Ns_each rows");
  if (b.W > SB_W || b.k_pcs != 0 || b.N != b.nfull) return ctx->fail(KP_ERR_ARG, "kp_fit_batch: needs W <= 16 and no dimension reduction");
  if (Ns_each > 0x7fffffff) return ctx->fail(KP_ERR_ARG, "kp_fit_batch: too many snapshots per system");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int W = b.W;
  const size_t bW = (size_t)nb * W * W * 8;
  char* ws = (char*)ctx->workspace(6, 3 * bW + (size_t)nb * 4 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_batch: out of device memory");
  double* dK = (double*)ws;
  double* dG = (double*)(ws + bW);
  double* dC = (double*)(ws + 2 * bW);
  int* dS = (int*)(ws + 3 * bW);
  hipStream_t s = ctx->stream;
  size_t lds = ((size_t)(2 * b.nvars + (b.m > 0 ? b.m : 1)) * SB_TS + 2 * SB_TS * SB_LD + 5 * 16 * SB_LD + 16) * sizeof(double);
  // power-table lift for monomial dictionaries whose table fits (2 sides x nvars x depth x 64 snapshots)
  // refinement steps with the residual taken from the data (default 1; KP_BATCH_REFINE=0 gives the plain normal equations)
  static const int refine = [] { const char* e = getenv("KP_BATCH_REFINE"); return e ? std::max(0, std::min(4, atoi(e))) : 1; }();
  const size_t pw_bytes = (size_t)2 * b.nvars * basis->pow_depth * SB_TS * sizeof(double);
  const bool use_rec = basis->fast && basis->d_recipes && basis->pow_depth >= 1 && lds + pw_bytes <= 64 * 1024;
  if (use_rec) lds += pw_bytes;
  KP_HIP(ctx, kp_snaps_acquire(snaps, s));
  KP_HIP(ctx, hipEventRecord(ctx->ev0, s));
  hipLaunchKernelGGL(kp_small_fit_kernel, dim3(nb), dim3(256), lds, s, b, snaps->alpha, snaps->beta, snaps->u, snaps->Ns, (int)Ns_each, dK, dG, dC,
                     dS, use_rec ? (const uint32_t*)basis->d_recipes : nullptr, basis->pow_depth, basis->max_factors > 0 ? basis->max_factors : 1,
                     refine, TrajView{}, 0);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, kp_snaps_release(snaps, s));
  KP_HIP(ctx, hipEventRecord(ctx->ev1, s));
  KP_HIP(ctx, hipMemcpyAsync(K_out, dK, bW, hipMemcpyDeviceToHost, s));
  if (G_out) KP_HIP(ctx, hipMemcpyAsync(G_out, dG, bW, hipMemcpyDeviceToHost, s));
  if (C_out) KP_HIP(ctx, hipMemcpyAsync(C_out, dC, bW, hipMemcpyDeviceToHost, s));
  if (status_out) KP_HIP(ctx, hipMemcpyAsync(status_out, dS, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[0] = ms;
  return KP_OK;
}

// ---- batched M-projection of get_model (Ksysid.m:1206-1225) for the small linear models of the sweep ---------------
// Per model, with K1 = K(:, 1:N) (W x N), A0 = K1(1:N, :)', B0 = K1(N+1:W, :)':
//   L'L = K1' G K1,   L'R = K1' C(:, 1:N),   M' = (L'L) \ (L'R),   A = M A0,  B = M B0.
// Thread (i, j) of the 16 x 16 grid owns one element of every product; matrices live in LDS with row stride SB_LD.
__global__ __launch_bounds__(256) void kp_small_project_kernel(const double* __restrict__ K, const double* __restrict__ G,
                                                               const double* __restrict__ C, int N, int m, double* __restrict__ Aout,
                                                               double* __restrict__ Bout, double* __restrict__ Mout, int* __restrict__ status) {
  __shared__ double Ks[16 * SB_LD], Gs[16 * SB_LD], Cs[16 * SB_LD], Ts[16 * SB_LD], LL[16 * SB_LD], LR[16 * SB_LD], Ms[16 * SB_LD], Ls[16 * SB_LD], Dd[16];
  __shared__ int bad;
  const int tid = threadIdx.x, i = tid >> 4, j = tid & 15, sys = blockIdx.x, W = N + m;
  const size_t off = (size_t)sys * W * W;
  const bool in = i < W && j < W;
  Ks[i * SB_LD + j] = in ? K[off + (size_t)j * W + i] : 0.0;            // [row][col]
  Gs[i * SB_LD + j] = in ? G[off + (size_t)j * W + i] : 0.0;
  Cs[i * SB_LD + j] = in ? C[off + (size_t)j * W + i] : 0.0;
  if (tid == 0) bad = 0;
  __syncthreads();
  double t = 0.0;                                                        // T = G K1  (W x N)
  if (j < N)
    for (int w = 0; w < W; ++w) t += Gs[i * SB_LD + w] * Ks[w * SB_LD + j];
  Ts[i * SB_LD + j] = t;
  __syncthreads();
  double ll = 0.0, lr = 0.0;                                             // L'L = K1' T, L'R = K1' C(:, 1:N)  (N x N)
  if (i < N && j < N)
    for (int w = 0; w < W; ++w) {
      ll += Ks[w * SB_LD + i] * Ts[w * SB_LD + j];
      lr += Ks[w * SB_LD + i] * Cs[w * SB_LD + j];
    }
  LL[i * SB_LD + j] = (i < N && j < N) ? ll : (i == j ? 1.0 : 0.0);      // identity on the padding
  LR[i * SB_LD + j] = (i < N && j < N) ? lr : 0.0;
  __syncthreads();
  sb_spd_solve16(LL, LR, Ms, Ls, Dd, &bad);                               // Ms = M' (N x N)
  const double nanv = __builtin_nan("");
  double a = 0.0, b = 0.0;                                               // A = M A0: A[i][j] = sum_k M[i][k] K[j][k]; B[i][j] = sum_k M[i][k] K[N + j][k]
  if (i < N)
    for (int k = 0; k < N; ++k) {
      const double mik = Ms[k * SB_LD + i];                              // M[i][k] = M'[k][i]
      if (j < N) a += mik * Ks[j * SB_LD + k];
      if (j < m) b += mik * Ks[(N + j) * SB_LD + k];
    }
  if (i < N && j < N) {
    Aout[(size_t)sys * N * N + (size_t)j * N + i] = bad ? nanv : a;
    if (Mout) Mout[(size_t)sys * N * N + (size_t)j * N + i] = bad ? nanv : Ms[j * SB_LD + i];
  }
  if (i < N && j < m) Bout[(size_t)sys * N * m + (size_t)j * N + i] = bad ? nanv : b;
  if (tid == 0 && status) status[sys] = bad;
}

extern "C" int kp_model_project_batch(kp_ctx* ctx, const double* K, const double* G, const double* C, int nb, int N, int m, double* A_out,
                                      double* B_out, double* M_out, int* status_out) {
  if (!ctx || !K || !G || !C || !A_out || !B_out || nb < 1 || N < 1 || m < 0 || N + m > SB_W)
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_model_project_batch: bad argument (needs N + m <= 16)") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int W = N + m;
  const size_t bW = (size_t)nb * W * W * 8, bA = (size_t)nb * N * N * 8, bB = (size_t)nb * N * (m > 0 ? m : 1) * 8;
  char* ws = (char*)ctx->workspace(6, 3 * bW + 2 * bA + bB + (size_t)nb * 4 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_model_project_batch: out of device memory");
  double *dK = (double*)ws, *dG = (double*)(ws + bW), *dC = (double*)(ws + 2 * bW);
  double *dA = (double*)(ws + 3 * bW), *dM = (double*)(ws + 3 * bW + bA), *dB = (double*)(ws + 3 * bW + 2 * bA);
  int* dS = (int*)(ws + 3 * bW + 2 * bA + bB);
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dK, K, bW, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dG, G, bW, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dC, C, bW, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(kp_small_project_kernel, dim3(nb), dim3(256), 0, s, dK, dG, dC, N, m, dA, dB, M_out ? dM : nullptr, dS);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipMemcpyAsync(A_out, dA, bA, hipMemcpyDeviceToHost, s));
  if (m > 0) KP_HIP(ctx, hipMemcpyAsync(B_out, dB, (size_t)nb * N * m * 8, hipMemcpyDeviceToHost, s));
  if (M_out) KP_HIP(ctx, hipMemcpyAsync(M_out, dM, bA, hipMemcpyDeviceToHost, s));
  if (status_out) KP_HIP(ctx, hipMemcpyAsync(status_out, dS, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  return KP_OK;
}


// =====================================================================================================================
// evaluate_rand_models.m:45-144 with the data resident on the device: trajectories uploaded once (kp_traj), then per
// (model type, degree) ONE call does scaling + snapshot pairs + fit (kp_small_fit_kernel) + model extraction (M-projection
// of get_model for linear models) + the validation rollout + the normalised mean error (:69-72), and only the error
// table entries come back.
// =====================================================================================================================
struct kp_traj {
  kp_ctx* ctx = nullptr;
  int nb = 0, ntrials = 0, T = 0, n = 0, m = 0, Tv = 0;
  double *Y = nullptr, *U = nullptr, *Yv = nullptr, *Uv = nullptr, *sc = nullptr;
  size_t cap[5] = {0, 0, 0, 0, 0};      // real sizes of the five blocks (they may come from the context's pool)
  int have = 0;                         // bit w: block w has been put; 31: scaled (ready)
};

// min / max per column over the training rows of one system -> offset (max+min)/2, factor (max-min)/2 (1 when the range
// is zero), Ksysid.m:187-210
__global__ __launch_bounds__(256) void kp_traj_scale_kernel(const double* __restrict__ Y, const double* __restrict__ U, int rows, int n, int m,
                                                            double* __restrict__ sc) {
  __shared__ double rmin[4], rmax[4];
  const int sys = blockIdx.x, tid = threadIdx.x;
  double* out = sc + (size_t)sys * 2 * (n + m);
  for (int v = 0; v < n + m; ++v) {
    const double* col = v < n ? Y + ((size_t)sys * n + v) * rows : U + ((size_t)sys * m + (v - n)) * rows;
    double lo = 1e300, hi = -1e300;
    for (int r = tid; r < rows; r += 256) { const double x = col[r]; lo = fmin(lo, x); hi = fmax(hi, x); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    __syncthreads();
    if ((tid & 63) == 0) { rmin[tid >> 6] = lo; rmax[tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
      lo = fmin(fmin(rmin[0], rmin[1]), fmin(rmin[2], rmin[3]));
      hi = fmax(fmax(rmax[0], rmax[1]), fmax(rmax[2], rmax[3]));
      const double off = (hi + lo) / 2.0, fac = (hi - lo) / 2.0;
      if (v < n) { out[v] = off; out[n + v] = fac == 0.0 ? 1.0 : fac; }
      else { out[2 * n + (v - n)] = off; out[2 * n + m + (v - n)] = fac == 0.0 ? 1.0 : fac; }
    }
  }
}

// scale_data (Ksysid.m:308-343) once, in place: training and validation trials with the training factors
__global__ void kp_traj_apply_scale_kernel(double* __restrict__ X, int rows, int width, int n, int m, int is_u, const double* __restrict__ sc) {
  const int sys = blockIdx.y;
  const double* s = sc + (size_t)sys * 2 * (n + m);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * width; e += gridDim.x * blockDim.x) {
    const int v = e / rows;
    const double off = is_u ? s[2 * n + v] : s[v], fac = is_u ? s[2 * n + m + v] : s[n + v];
    double* x = X + (size_t)sys * rows * width + e;
    *x = (*x - off) / fac;
  }
}

static TrajView traj_view(const kp_traj* t) {
  const unsigned long long d = (unsigned long long)(t->T - 1);
  return TrajView{t->Y, t->U, t->sc, t->ntrials, t->T, t->ntrials * t->T, t->n, t->m, ((1ull << 40) + d - 1) / d};
}

#define KP_TRAJ_POOL_MAX 10
// a block of at least `bytes` (and at most twice that) from the context's pool, else a new one; cap = its real size
static hipError_t traj_alloc(kp_ctx* ctx, double** p, size_t bytes, size_t* cap) {
  static const bool no_pool = getenv("KP_NO_TRAJ_POOL") != nullptr;
  if (!no_pool) {
    std::lock_guard<std::mutex> lk(ctx->host_mu);     // the pool is shared with a host thread that builds the next object
    int best = -1;
    for (size_t i = 0; i < ctx->traj_pool.size(); ++i) {
      const size_t c = ctx->traj_pool[i].second;
      if (c >= bytes && c <= 2 * bytes + 4096 && (best < 0 || c < ctx->traj_pool[best].second)) best = (int)i;
    }
    if (best >= 0) {
      *p = (double*)ctx->traj_pool[best].first;
      *cap = ctx->traj_pool[best].second;
      ctx->traj_pool.erase(ctx->traj_pool.begin() + best);
      return hipSuccess;
    }
  }
  *cap = bytes;
  return hipMalloc((void**)p, bytes);
}
static void traj_release(kp_ctx* ctx, void* p, size_t cap) {
  if (!p) return;
  static const bool no_pool = getenv("KP_NO_TRAJ_POOL") != nullptr;
  if (no_pool || !ctx) { (void)hipFree(p); return; }
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  ctx->traj_pool.push_back({p, cap});
  while (ctx->traj_pool.size() > KP_TRAJ_POOL_MAX) {       // the oldest goes
    (void)hipFree(ctx->traj_pool.front().first);
    ctx->traj_pool.erase(ctx->traj_pool.begin());
  }
}
void kp_traj_pool_free(kp_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  for (auto& b : ctx->traj_pool) (void)hipFree(b.first);
  ctx->traj_pool.clear();
}

// The object in three steps, so that a caller that assembles the four blocks one after the other (the host mirror gathers
// them from thousands of small trial arrays) has each one on its way to the device while it prepares the next:
//   kp_traj_create (device blocks), kp_traj_put (which = 0 Y, 1 U, 2 Yv, 3 Uv: ONE host-to-device copy enqueued on the
//   context's stream - asynchronous when `host` is page-locked, e.g. a kp_host_alloc block, which must then stay untouched
//   until kp_traj_finish), kp_traj_finish (scaling on the device, stream synchronised: the object is ready).
extern "C" int kp_traj_create(kp_ctx* ctx, int nb, int ntrials, int T, int n, int m, int Tv, kp_traj** out) {
  if (!ctx || !out || nb < 1 || ntrials < 1 || T < 3 || n < 1 || m < 1 || Tv < 2 || (int64_t)ntrials * T >= (1 << 24))
    return ctx ? ctx->fail(KP_ERR_ARG, "kp_traj_upload: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  kp_traj* t = new kp_traj;
  t->ctx = ctx; t->nb = nb; t->ntrials = ntrials; t->T = T; t->n = n; t->m = m; t->Tv = Tv;
  const size_t rows = (size_t)ntrials * T;
  const size_t bY = (size_t)nb * rows * n * 8, bU = (size_t)nb * rows * m * 8, bYv = (size_t)nb * Tv * n * 8, bUv = (size_t)nb * Tv * m * 8;
  hipError_t e = traj_alloc(ctx, &t->Y, bY, &t->cap[0]);
  if (e == hipSuccess) e = traj_alloc(ctx, &t->U, bU, &t->cap[1]);
  if (e == hipSuccess) e = traj_alloc(ctx, &t->Yv, bYv, &t->cap[2]);
  if (e == hipSuccess) e = traj_alloc(ctx, &t->Uv, bUv, &t->cap[3]);
  if (e == hipSuccess) e = traj_alloc(ctx, &t->sc, (size_t)nb * 2 * (n + m) * 8, &t->cap[4]);
  if (e != hipSuccess) {
    double* bufs[] = {t->Y, t->U, t->Yv, t->Uv, t->sc};
    for (double* p : bufs)
      if (p) (void)hipFree(p);
    delete t;
    return ctx->fail(KP_ERR_HIP, std::string("kp_traj_upload: ") + hipGetErrorString(e));
  }
  *out = t;
  return KP_OK;
}

extern "C" int kp_traj_put(kp_traj* t, int which, const double* host) {
  if (!t || !host || which < 0 || which > 3) return t ? t->ctx->fail(KP_ERR_ARG, "kp_traj_put: bad argument") : KP_ERR_ARG;
  kp_ctx* ctx = t->ctx;
  // a finished object holds SCALED data: a raw block put behind kp_traj_finish would never be rescaled (finish is a no-op
  // then) and the sweep would run on a mix of scaled and raw values
  if (t->have == 31) return ctx->fail(KP_ERR_ARG, "kp_traj_put: the object is finished (scaled); create a new one");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t rows = (size_t)t->ntrials * t->T;
  double* dst[4] = {t->Y, t->U, t->Yv, t->Uv};
  const size_t bytes[4] = {(size_t)t->nb * rows * t->n * 8, (size_t)t->nb * rows * t->m * 8, (size_t)t->nb * t->Tv * t->n * 8,
                           (size_t)t->nb * t->Tv * t->m * 8};
  KP_HIP(ctx, hipMemcpyAsync(dst[which], host, bytes[which], hipMemcpyHostToDevice, ctx->stream));
  t->have |= 1 << which;
  return KP_OK;
}

extern "C" int kp_traj_finish(kp_traj* t) {
  if (!t) return KP_ERR_ARG;
  kp_ctx* ctx = t->ctx;
  if (t->have == 31) return KP_OK;
  if (t->have != 15) return ctx->fail(KP_ERR_ARG, "kp_traj_finish: a block is missing (kp_traj_put of Y, U, Yv, Uv)");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int nb = t->nb, n = t->n, m = t->m, Tv = t->Tv;
  const size_t rows = (size_t)t->ntrials * t->T;
  hipLaunchKernelGGL(kp_traj_scale_kernel, dim3(nb), dim3(256), 0, s, t->Y, t->U, (int)rows, n, m, t->sc);
  hipLaunchKernelGGL(kp_traj_apply_scale_kernel, dim3(16, nb), dim3(256), 0, s, t->Y, (int)rows, n, n, m, 0, t->sc);
  hipLaunchKernelGGL(kp_traj_apply_scale_kernel, dim3(16, nb), dim3(256), 0, s, t->U, (int)rows, m, n, m, 1, t->sc);
  hipLaunchKernelGGL(kp_traj_apply_scale_kernel, dim3(2, nb), dim3(256), 0, s, t->Yv, Tv, n, n, m, 0, t->sc);
  hipLaunchKernelGGL(kp_traj_apply_scale_kernel, dim3(2, nb), dim3(256), 0, s, t->Uv, Tv, m, n, m, 1, t->sc);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipStreamSynchronize(s));
  t->have = 31;                                       // scaled: a second finish would scale again
  return KP_OK;
}

extern "C" int kp_traj_upload(kp_ctx* ctx, const double* Y, const double* U, int nb, int ntrials, int T, int n, int m, const double* Yv,
                              const double* Uv, int Tv, kp_traj** out) {
  if (!ctx || !Y || !U || !Yv || !Uv || !out) return ctx ? ctx->fail(KP_ERR_ARG, "kp_traj_upload: bad argument") : KP_ERR_ARG;
  kp_traj* t = nullptr;
  int rc = kp_traj_create(ctx, nb, ntrials, T, n, m, Tv, &t);
  if (rc) return rc;
  const double* src[4] = {Y, U, Yv, Uv};
  for (int w = 0; w < 4 && rc == KP_OK; ++w) rc = kp_traj_put(t, w, src[w]);
  if (rc == KP_OK) rc = kp_traj_finish(t);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    double* bufs[] = {t->Y, t->U, t->Yv, t->Uv, t->sc};
    for (double* p : bufs)
      if (p) (void)hipFree(p);
    delete t;
    return rc;
  }
  *out = t;
  return KP_OK;
}

extern "C" int kp_traj_destroy(kp_traj* t) {
  if (!t) return KP_OK;
  double* bufs[] = {t->Y, t->U, t->Yv, t->Uv, t->sc};
  // the blocks go back to the context's pool - once the device is done with them (every entry point that used them has
  // synchronised its stream before returning; an object abandoned between kp_traj_put and kp_traj_finish may still have
  // copies in flight)
  if (t->have != 0 && t->have != 31 && t->ctx) {
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
  }
  for (int i = 0; i < 5; ++i) traj_release(t->ctx, bufs[i], t->cap[i]);
  delete t;
  return KP_OK;
}

extern "C" int kp_traj_dims(const kp_traj* t, int* nb, int* ntrials, int* T, int* n, int* m, int* Tv) {
  if (!t) return KP_ERR_ARG;
  if (nb) *nb = t->nb;
  if (ntrials) *ntrials = t->ntrials;
  if (T) *T = t->T;
  if (n) *n = t->n;
  if (m) *m = t->m;
  if (Tv) *Tv = t->Tv;
  return KP_OK;
}

extern "C" int kp_traj_scale(kp_traj* t, double* sc_out) {
  if (!t || !sc_out) return KP_ERR_ARG;
  kp_ctx* ctx = t->ctx;
  if (t->have != 31) return ctx->fail(KP_ERR_ARG, "kp_traj_scale: trajectory object not finished (kp_traj_finish)");
  KP_HIP(ctx, hipMemcpy(sc_out, t->sc, (size_t)t->nb * 2 * (t->n + t->m) * 8, hipMemcpyDeviceToHost));
  return KP_OK;
}

// Validation rollout + error of one fitted model per system, one wave per system (N <= 16): val_model / val_BLmodel /
// val_NLmodel (Ksysid.m:1623-1879) on the scaled validation trial, then evaluate_rand_models.m:70-72:
// err_j = mean_t |y_sim - y_real|_j / (sum_t |y_real|_j / T).  Lane r owns component r of the lifted state.
// Ks: this system's K (column-major, leading dimension W); Aps / Bps: its projected A (N x N), B (N x m) for linear models
__device__ __forceinline__ void sweep_rollout_body(const BasisDev& b, const double* __restrict__ Ks, const double* __restrict__ Aps,
                                                   const double* __restrict__ Bps, const double* __restrict__ yv,
                                                   const double* __restrict__ uv, int Tv, int bad_fit, double* __restrict__ err_sys,
                                                   double* vsh) {
  const int lane = threadIdx.x;
  const int N = b.N, W = b.W, n = b.nzeta, m = b.m, mt = b.model_type;
  const int r = lane < N ? lane : 0;
  double arow[16], brow[3][16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    arow[c] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) brow[i][c] = 0.0;
  }
  if (mt == KP_MODEL_LINEAR) {          // projected model M A, M B (get_model, Ksysid.m:1224-1225)
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c < N) arow[c] = Aps[(size_t)c * N + r];
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < m) brow[i][0] = Bps[(size_t)i * N + r];
  } else if (mt == KP_MODEL_BILINEAR) { // A = UT(1:N,1:N), B = UT(1:N,N+1:end), UT = K' (get_BLmodel, Ksysid.m:1250-1259)
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < N) {
        arow[c] = Ks[c + (size_t)r * W];
#pragma unroll
        for (int i = 0; i < 3; ++i) if (i < m) brow[i][c] = Ks[N + N * i + c + (size_t)r * W];
      }
  } else {                              // F = K(:,1:nzeta)' basis (get_NLmodel, Ksysid.m:1325-1329): lane c keeps column c of Kf
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < n) arow[q] = Ks[r + (size_t)q * W];
  }
  // scaled first validation row -> lifted state
  if (lane < n) vsh[lane] = yv[(size_t)lane * Tv];
  if (mt == KP_MODEL_NONLINEAR && lane < m) vsh[n + lane] = uv[(size_t)lane * Tv];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double z = 0.0;
  if (mt != KP_MODEL_NONLINEAR) { if (lane < N) z = kp_eval_col(b, b.cols[lane], vsh, 1); }
  else if (lane < n) z = vsh[lane];
  double acc_e = 0.0, acc_a = 0.0;
  for (int t = 0; t < Tv; ++t) {
    if (lane < n) {
      const double yr = yv[(size_t)lane * Tv + t];
      if (t > 0) acc_e += fabs(z - yr);               // the first simulated row is the measured one (Ksysid.m:1654)
      acc_a += fabs(yr);
    }
    if (t == Tv - 1) break;
    double ut[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < m) ut[i] = uv[(size_t)i * Tv + t];
    double zn = 0.0;
    if (mt == KP_MODEL_LINEAR) {
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < N) zn += arow[c] * sb_bcast(z, c);
#pragma unroll
      for (int i = 0; i < 3; ++i) if (i < m) zn += brow[i][0] * ut[i];
    } else if (mt == KP_MODEL_BILINEAR) {              // z+ = A z + sum_i u_i B_i z (val_BLmodel, Ksysid.m:1772-1787)
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < N) {
          double w = arow[c];
#pragma unroll
          for (int i = 0; i < 3; ++i) if (i < m) w += ut[i] * brow[i][c];
          zn += w * sb_bcast(z, c);
        }
    } else {                                           // zeta+ = Kf psi([zeta; u]) (val_NLmodel, Ksysid.m:1848-1863)
      if (lane < n) vsh[lane] = z;
      if (lane < m) vsh[n + lane] = ut[lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double psi = lane < N ? kp_eval_col(b, b.cols[lane], vsh, 1) : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < n) {
          double sacc = lane < N ? arow[q] * psi : 0.0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
          if (lane == q) zn = sacc;
        }
      __builtin_amdgcn_wave_barrier();
    }
    z = zn;
  }
  if (lane < n) {
    const double bad = bad_fit ? __builtin_nan("") : 0.0;
    err_sys[lane] = (acc_e / Tv) / (acc_a / Tv) + bad;
  }
}


__global__ __launch_bounds__(64) void kp_sweep_rollout_kernel(BasisDev b, const double* __restrict__ K, const double* __restrict__ Ap,
                                                              const double* __restrict__ Bp, const double* __restrict__ Yv,
                                                              const double* __restrict__ Uv, const double* __restrict__ scv, int Tv,
                                                              const int* __restrict__ fit_status, double* __restrict__ err) {
  __shared__ double vsh[KP_MAX_VARS];
  const int sys = blockIdx.x, n = b.nzeta, m = b.m;
  sweep_rollout_body(b, K + (size_t)sys * b.W * b.W, Ap ? Ap + (size_t)sys * b.N * b.N : nullptr, Bp ? Bp + (size_t)sys * b.N * m : nullptr,
                     Yv + (size_t)sys * Tv * n, Uv + (size_t)sys * Tv * m, Tv, fit_status && fit_status[sys], err + (size_t)sys * n, vsh);
}

__global__ void kp_l1_flag_kernel(const double* __restrict__ K, int W, double t, int* __restrict__ flags) {
  const int sys = blockIdx.x;
  double s = 0.0;
  for (int e = threadIdx.x; e < W * W; e += 64) s += fabs(K[(size_t)sys * W * W + e]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) flags[sys] = s > t ? 1 : 0;
}

extern "C" int kp_sweep_eval(kp_ctx* ctx, const kp_traj* traj, const kp_basis* basis, double lasso, double* err_out, double* K_out,
                             int* status_out) {
  if (!ctx || !traj || !basis || !err_out) return ctx ? ctx->fail(KP_ERR_ARG, "kp_sweep_eval: bad argument") : KP_ERR_ARG;
  if (traj->have != 31) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval: trajectory object not finished (kp_traj_finish)");
  const BasisDev& b = basis->dev;
  if (b.nzeta != traj->n || b.m != traj->m) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval: trajectory / dictionary dimension mismatch");
  if (b.W > SB_W || b.k_pcs != 0 || b.N != b.nfull || b.m > 3) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval: needs W <= 16, m <= 3 and no dimension reduction");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int nb = traj->nb, W = b.W, N = b.N, n = traj->n, m = traj->m;
  const int Ns = traj->ntrials * (traj->T - 1) - 1;            // get_snapshotPairs: #good - 1 (Ksysid.m:960)
  const size_t bW = (size_t)nb * W * W * 8, bA = (size_t)nb * N * N * 8, bB = (size_t)nb * N * m * 8;
  char* ws = (char*)ctx->workspace(6, 3 * bW + bA + bB + (size_t)nb * (n * 8 + 12) + 256);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_sweep_eval: out of device memory");
  double *dK = (double*)ws, *dG = (double*)(ws + bW), *dC = (double*)(ws + 2 * bW);
  double *dA = (double*)(ws + 3 * bW), *dB = (double*)(ws + 3 * bW + bA), *dE = (double*)(ws + 3 * bW + bA + bB);
  int* dS = (int*)(dE + (size_t)nb * n);
  int* dF = dS + nb;
  hipStream_t s = ctx->stream;
  size_t lds = ((size_t)(2 * b.nvars + (b.m > 0 ? b.m : 1)) * SB_TS + 2 * SB_TS * SB_LD + 5 * 16 * SB_LD + 16) * sizeof(double);
  static const int refine = [] { const char* e = getenv("KP_BATCH_REFINE"); return e ? std::max(0, std::min(4, atoi(e))) : 1; }();
  const size_t pw_bytes = (size_t)2 * b.nvars * basis->pow_depth * SB_TS * sizeof(double);
  const bool use_rec = basis->fast && basis->d_recipes && basis->pow_depth >= 1 && lds + pw_bytes <= 64 * 1024;
  if (use_rec) lds += pw_bytes;
  TrajView tv = traj_view(traj);
  KP_HIP(ctx, hipEventRecord(ctx->ev0, s));
  hipLaunchKernelGGL(kp_small_fit_kernel, dim3(nb), dim3(256), lds, s, b, nullptr, nullptr, nullptr, (int64_t)0, Ns, dK, dG, dC, dS,
                     use_rec ? (const uint32_t*)basis->d_recipes : nullptr, basis->pow_depth, basis->max_factors > 0 ? basis->max_factors : 1,
                     refine, tv, 0);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, s));
  if (lasso < 1e6) {
    // lasso rows of the sweep (nonlinear models, lasso = 4, evaluate_rand_models.m:122): the L1 row is active only when
    // ||K_LS||_1 > t = lasso N, which the generated and shipped systems never reach; flagged systems are re-solved
    const double t = lasso * N;
    hipLaunchKernelGGL(kp_l1_flag_kernel, dim3(nb), dim3(64), 0, s, dK, W, t, dF);
    KP_HIP(ctx, hipGetLastError());
    std::vector<int> flags(nb);
    KP_HIP(ctx, hipMemcpyAsync(flags.data(), dF, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    for (int q = 0; q < nb; ++q)
      if (flags[q]) {
        int rc = kp_lasso_dev(ctx, dG + (size_t)q * W * W, dC + (size_t)q * W * W, W, W, t, 20000, 1e-10, dK + (size_t)q * W * W, nullptr);
        if (rc && rc != KP_ERR_NOT_CONVERGED) return rc;
      }
  }
  if (b.model_type == KP_MODEL_LINEAR) {
    hipLaunchKernelGGL(kp_small_project_kernel, dim3(nb), dim3(256), 0, s, dK, dG, dC, N, m, dA, dB, (double*)nullptr, (int*)nullptr);
    KP_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(kp_sweep_rollout_kernel, dim3(nb), dim3(64), 0, s, b, dK, dA, dB, traj->Yv, traj->Uv, traj->sc, traj->Tv, dS, dE);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipMemcpyAsync(err_out, dE, (size_t)nb * n * 8, hipMemcpyDeviceToHost, s));
  if (K_out) KP_HIP(ctx, hipMemcpyAsync(K_out, dK, bW, hipMemcpyDeviceToHost, s));
  if (status_out) KP_HIP(ctx, hipMemcpyAsync(status_out, dS, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[0] = ms;
  return KP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Gram pass of the nested sweep: G = Px'Px, C = Px'Py of every system straight from its (scaled) trajectories, monomial /
// Chebyshev dictionaries through the power-table recipes, W <= 16.  One workgroup per system, 128 snapshot pairs per tile:
//   * thread (side, p) lifts ONE side of ONE pair: fills its own column of the power table (no barrier between the fill
//     and the reads: a thread only reads what it wrote), forms the N columns and writes its row of Px or Py;
//   * the raw values of the NEXT tile are loaded into registers before the current tile is processed (the global-memory
//     latency is off the critical path; the per-tile chain raw load -> barrier -> table -> barrier -> lift -> barrier of
//     kp_small_fit_kernel is what kept that kernel at a tenth of the f64 rate);
//   * accumulation as there: a lane owns one 4 x 4 block of G and of C for 8 of the 128 pairs, three 16-byte LDS reads
//     per 32 FMAs, the 16 snapshot groups summed in a fixed order at the end.
// Two barriers per 128 pairs.  No solve here: kp_sweep_sub_kernel factors the sub-blocks of every degree.
// ---------------------------------------------------------------------------------------------------------------------
#define TG_TS 128
__global__ __launch_bounds__(256) void kp_traj_gram_kernel(BasisDev b, const uint32_t* __restrict__ recipes, int D, int nfmax, TrajView tv,
                                                           int Ns, int cheb, double* __restrict__ Gout, double* __restrict__ Cout) {
  extern __shared__ __align__(16) double sm[];
  // LDS: Px[TS][LD] | Py[TS][LD] | pw[side][v * D + e - 1][TS] | Gs[16][LD] | Cs[16][LD]
  const int nv = b.nvars, m = b.m, N = b.N, W = b.W, nz = b.nzeta;
  double* Px = sm;
  double* Py = Px + TG_TS * SB_LD;
  double* pw = Py + TG_TS * SB_LD;
  double* Gs = pw + 2 * nv * D * TG_TS;
  double* Cs = Gs + 16 * SB_LD;
  __shared__ uint32_t recs[SB_W];
  const int tid = threadIdx.x, sys = blockIdx.x;
  if (tid < N) recs[tid] = recipes[tid];
  for (int e = tid; e < 2 * TG_TS * SB_LD; e += 256) Px[e] = 0.0;          // unused columns stay zero
  const int side = tid >> 7, p = tid & (TG_TS - 1);
  const int blk = tid & 15, grp = tid >> 4;
  const int bi = (blk >> 2) * 4, bj = (blk & 3) * 4;
  const int Tm1 = tv.T - 1;
  double g[4][4], c[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) g[x][y] = c[x][y] = 0.0;
  // raw values of a pair for this thread's side: the nv dictionary variables ([y] or [y; u]), then the m inputs
  double raw[KP_MAX_VARS > 8 ? 8 : KP_MAX_VARS], rin[3];
  auto load_raw = [&](int r0) {
    const int pair = r0 + p;
    const bool ok = pair < Ns;
    const int tr = ok ? (int)(((unsigned long long)pair * tv.div_magic) >> 40) : 0;
    const int row = ok ? tr * tv.T + (pair - tr * Tm1) : 0;
#pragma unroll
    for (int v = 0; v < 8; ++v)
      if (v < nv) {
        // no arithmetic on the loaded value here (not even the tail mask): the wave would wait for the load on the spot
        // instead of at its use, a whole accumulation phase later
        raw[v] = v < nz ? tv.Y[((size_t)sys * tv.n + v) * tv.rows + row + side] : tv.U[((size_t)sys * tv.m + (v - nz)) * tv.rows + row];
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < m) rin[i] = tv.U[((size_t)sys * tv.m + i) * tv.rows + row];
    return ok;
  };
  bool ok_next = load_raw(0);
  __syncthreads();
  for (int r0 = 0; r0 < Ns; r0 += TG_TS) {
    // ---- lift of this thread's (side, pair) from the registers loaded one tile ago ----
    const bool ok = ok_next;
    double* mypw = pw + (size_t)side * nv * D * TG_TS + p;
#pragma unroll
    for (int v = 0; v < 8; ++v)
      if (v < nv) {
        const double x = ok ? raw[v] : 0.0;               // pairs past Ns: zero row (the constant column carries `ok` too)
        double q = x, qm = 1.0;
        for (int k = 0; k < D; ++k) {
          mypw[(v * D + k) * TG_TS] = q;
          const double qn = cheb ? 2.0 * x * q - qm : q * x;
          qm = q;
          q = qn;
        }
      }
    double uin[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) uin[i] = (i < m && ok) ? rin[i] : 0.0;
    double* P = (side ? Py : Px) + p * SB_LD;
    for (int col = 0; col < N; ++col) {
      const uint32_t rc = recs[col];
      double val = ok ? 1.0 : 0.0;
      for (int f = 0; f < nfmax; ++f) {
        const uint32_t id = (rc >> (8 * f)) & 255u;
        const double t = mypw[(id == 255u ? 0u : id) * TG_TS];
        val *= id == 255u ? 1.0 : t;
      }
      P[col] = val;
      if (b.model_type == KP_MODEL_BILINEAR)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < m) P[(i + 1) * N + col] = val * uin[i];
    }
    if (b.model_type == KP_MODEL_LINEAR)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < m) P[N + i] = uin[i];
    // the next tile's raw values: in flight while this tile is accumulated
    ok_next = load_raw(r0 + TG_TS);
    __syncthreads();
    // ---- accumulation: 16 blocks x 16 snapshot groups of 8 ----
#pragma unroll 2
    for (int s_ = 0; s_ < TG_TS / 16; ++s_) {
      const double* xr = Px + (grp * (TG_TS / 16) + s_) * SB_LD;
      const double* yr = Py + (grp * (TG_TS / 16) + s_) * SB_LD;
      const double2 xa = *reinterpret_cast<const double2*>(xr + bi), xb = *reinterpret_cast<const double2*>(xr + bi + 2);
      const double2 ja = *reinterpret_cast<const double2*>(xr + bj), jb = *reinterpret_cast<const double2*>(xr + bj + 2);
      const double2 ya = *reinterpret_cast<const double2*>(yr + bj), yb = *reinterpret_cast<const double2*>(yr + bj + 2);
      const double xi[4] = {xa.x, xa.y, xb.x, xb.y}, xj[4] = {ja.x, ja.y, jb.x, jb.y}, yj[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          g[x][y] += xi[x] * xj[y];
          c[x][y] += xi[x] * yj[y];
        }
    }
    __syncthreads();
  }
  // sum of the 16 snapshot groups in a fixed order
  const int gi = tid >> 4, gj = tid & 15;
  Gs[gi * SB_LD + gj] = 0.0;
  Cs[gi * SB_LD + gj] = 0.0;
  for (int t = 0; t < 16; ++t) {
    __syncthreads();
    if (grp == t) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          Gs[(bi + x) * SB_LD + bj + y] += g[x][y];
          Cs[(bi + x) * SB_LD + bj + y] += c[x][y];
        }
    }
  }
  __syncthreads();
  if (gi < W && gj < W) {
    Gout[(size_t)sys * W * W + (size_t)gj * W + gi] = Gs[gi * SB_LD + gj];
    Cout[(size_t)sys * W * W + (size_t)gj * W + gi] = Cs[gi * SB_LD + gj];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same Gram pass on the matrix pipe (round 3).  kp_traj_gram_kernel above writes the lifted rows Px, Py to LDS and every
// thread re-reads them for its register block (three 16-byte reads per 32 FMAs): the LDS was busy 60 % of the time and the pass
// ran at 0.15-0.23 of the f64 rate.  Here NO lifted row exists: the power table (T_e(x_v) or x_v^e, the inputs, the tail
// mask, 1 and 0 as extra entries, per (side, pair)) is the only thing in LDS, and a lane builds the operands of
// v_mfma_f64_4x4x4_4b itself as products of F table entries through byte offsets fixed at kernel start:
//   A (row group g)  psi_x[pair k][col 4g + i],  i = lane & 3, k = lane >> 4       (4 values)
//   B                psi_x | psi_y [pair k][col 4 blk + i],  blk = (lane >> 2) & 3   (1 + 1 values)
// 6 F ds_read_b64 and 8 MFMAs per 4 pairs and wave (G and C, 16 x 16 each), no address arithmetic (the k-step enters the
// immediate offset); entry stride 129 doubles: the 16 lanes of a read pass that want 16 different entries of one pair hit 16
// different bank pairs.  A wave owns every fourth 4-pair step of a tile; the four partial sums meet in LDS at the end.
// One barrier per 128 pairs (the table is double buffered).  F = factors per operand value: bilinear columns psi_c u_i get
// their own table entries (UPRO: the filling thread multiplies once per pair what every lane would multiply per step).
// Measured per pass of 1024 systems x 9999 pairs (linear degree 13 | bilinear 6 | nonlinear 4): 0.74 | 0.59 | 0.88 ms ->
// 0.29 | 0.30 | 0.35 ms.  Timing-only ablations of the linear pass: without the MFMA phase 0.12 ms, with an empty tile loop
// 0.04 ms - the matrix phase (0.17 ms) runs at the instruction's rate (4 workgroups per CU x 78 tiles x 64 MFMAs x 16.5
// cycles), and since nothing on the VALU overlaps an f64 MFMA on the same SIMD the table fill and the loads ADD to it.  What
// mattered on the way: raw values 4 tiles ahead instead of 1, and ONE copy of the tile body (unrolled over the prefetch
// stages, the variables and the tail case the kernel was 115 KB of code and no faster than 0.33 ms).
// ---------------------------------------------------------------------------------------------------------------------
#define TGM_STR 129
#define TGM_PD 4                                      // tiles of raw values in flight per thread
#define TGM_NV 4                                      // dictionary variables ([zeta] or [zeta; u]) this kernel takes
template <int F, int DC, int NVC, bool UPRO>
__global__ __launch_bounds__(256, 2) void kp_traj_gram_mfma_kernel(BasisDev b, const uint32_t* __restrict__ recipes, int D_rt, TrajView tv, int Ns,
                                                                    double* __restrict__ Gout, double* __restrict__ Cout) {
  extern __shared__ __align__(16) double sm[];
  const int nv = b.nvars, m = b.m, N = b.N, W = b.W, nz = b.nzeta;
  const int D = DC > 0 ? DC : D_rt;                   // DC: the table depth at compile time (the sweep's 13 / 6 / 4): unrolled fill
  // table entries per (side, pair): powers | inputs | mask | 1 | 0 [| u_i x powers, i < m: `upro`, bilinear dictionaries - a
  // column psi_c u_i then costs the lanes no extra factor (6 multiplies per 4 pairs and lane) but the filling thread D]
  const int E = nv * D + m + 3 + (UPRO ? m * nv * D : 0);   // powers | inputs | mask | 0 | 1 [| input x powers]
  // (the 0 entry sits before the 1 entry: the padding columns of a B read use it, and with the sweep's linear dictionary
  //  - 13 powers, input, mask - it is then entry 15 of a read that spans entries 0 .. 15, one bank pair each; as entry 16 it
  //  shared a bank pair with entry 0: 20 % of the kernel's LDS cycles were conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT)
  const int id_u = nv * D, id_mask = id_u + m, id_zero = id_mask + 1, id_one = id_mask + 2, id_up = id_mask + 3;
  const int buf_doubles = 2 * E * TGM_STR;            // one buffer: [side][entry][TGM_STR]
  double* tab = sm;                                   // two buffers
  double* Gs = sm;                                    // [4 waves][2][16][16]: the partial sums, over the table once it is dead
  __shared__ uint32_t recs[SB_W];
  const int tid = threadIdx.x, sys = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < N) recs[tid] = recipes[tid];
  const int side = tid >> 7, p = tid & (TG_TS - 1);
  // constant entries of both buffers
  for (int e = tid; e < 2 * 2 * 2 * TGM_STR; e += 256) {
    const int pp = e % TGM_STR, which = (e / TGM_STR) & 1, sd = (e / (2 * TGM_STR)) & 1, bf = e / (4 * TGM_STR);
    tab[bf * buf_doubles + (sd * E + (which ? id_zero : id_one)) * TGM_STR + pp] = which ? 0.0 : 1.0;
  }
  __syncthreads();
  // ---- this lane's six operand values as F byte offsets each (entries of pair lane >> 4 of k-step `wave`) ----
  const int li = lane & 3, kq = lane >> 4, blk = (lane >> 2) & 3;
  int off[6][F];
#pragma unroll
  for (int sl = 0; sl < 6; ++sl) {
    const int c = sl < 4 ? 4 * sl + li : 4 * blk + li;          // column of Px (slots 0-4) or Py (slot 5)
    const int sd = sl == 5 ? 1 : 0;
    int ids[F];
#pragma unroll
    for (int f = 0; f < F; ++f) ids[f] = id_one;
    if (c >= W) ids[0] = id_zero;
    else {
      int pc = c, iu = -1;                                      // dictionary column and input factor of Px column c
      if (b.model_type == KP_MODEL_LINEAR) { if (c >= N) { pc = -1; iu = c - N; } }
      else if (b.model_type == KP_MODEL_BILINEAR) { pc = c % N; iu = c / N - 1; }
      int nf = 0;
      if (pc >= 0) {
        const uint32_t rc = recs[pc];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int id = (int)((rc >> (8 * f)) & 255u);
          if (id != 255 && nf < F) ids[nf++] = id;
        }
      }
      if (iu >= 0) {
        if (UPRO && nf > 0) ids[0] = id_up + iu * nv * D + ids[0];
        else if (nf < F) ids[nf++] = id_u + iu;
      }
      if (nf == 0) ids[0] = id_mask;                            // the constant column: 1 inside the data, 0 past it
    }
#pragma unroll
    for (int f = 0; f < F; ++f) off[sl][f] = (((sd * E + ids[f]) * TGM_STR) + 4 * wave + kq) * 8;
  }
  double accG[4] = {0.0, 0.0, 0.0, 0.0}, accC[4] = {0.0, 0.0, 0.0, 0.0};
  // Raw values of this thread's (side, pair), TGM_PD tiles ahead: a tile is ~0.6 us of matrix work now, less than a trip to
  // HBM under load - fetched ONE tile ahead (as kp_traj_gram_kernel does for its 4.7 us tiles) every tile waited for its data.
  // The stages rotate through register moves (a handful per tile): ONE copy of the tile body - unrolled over the stages the
  // kernel was 115 KB of code.
  constexpr int NVL = NVC > 0 ? NVC : TGM_NV;         // variables at compile time (the sweep: 1 or 2), else up to TGM_NV
  double raw[TGM_PD][NVL], rin[TGM_PD][3];
  double okq[TGM_PD];
  // base addresses of this thread's columns (the per-tile part of an address is the row alone)
  const double* pv[NVL];
  const double* pu[3];
#pragma unroll
  for (int v = 0; v < NVL; ++v)
    pv[v] = v < nz ? tv.Y + ((size_t)sys * tv.n + v) * tv.rows + side : tv.U + ((size_t)sys * tv.m + (v < nv ? v - nz : 0)) * tv.rows;
#pragma unroll
  for (int i = 0; i < 3; ++i) pu[i] = tv.U + ((size_t)sys * tv.m + (i < m ? i : 0)) * tv.rows;
  auto load_raw = [&](int r0, double (&rw)[NVL], double (&ri)[3]) {
    const int pair = r0 + p;
    const bool ok = pair < Ns;
    const int tr = ok ? (int)(((unsigned long long)pair * tv.div_magic) >> 40) : 0;
    const int row = ok ? tr + pair : 0;                // = tr T + (pair - tr (T - 1))
#pragma unroll
    for (int v = 0; v < NVL; ++v)
      if (v < nv) rw[v] = pv[v][row];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < m) ri[i] = pu[i][row];
    return ok ? 1.0 : 0.0;
  };
#pragma unroll
  for (int st = 0; st < TGM_PD; ++st) okq[st] = load_raw(st * TG_TS, raw[st], rin[st]);
  int bufsel = 0;
  for (int rr = 0; rr < Ns; rr += TG_TS, bufsel ^= 1) {
    // ---- table of this thread's (side, pair): T_1 .. T_D of every variable (the Chebyshev internal basis of the nested
    // sweep), the inputs, the mask.  Entries of a pair past Ns are 0 (times the mask), so that every product is.
    const double okf = okq[0];
    double* my = tab + bufsel * buf_doubles + side * E * TGM_STR + p;
    double ui[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ui[i] = i < m ? rin[0][i] * okf : 0.0;
    {
#pragma unroll
      for (int v = 0; v < NVL; ++v)
        if (v < nv) {
          const double x = raw[0][v], x2 = x + x;
          double q = x * okf, qm = okf;                // masked recurrence: T_k(x) okf for every k
          auto put = [&](int k, double val) {          // entry (v, k) and, for bilinear dictionaries, its products with the inputs
            my[(v * D + k) * TGM_STR] = val;
            if constexpr (UPRO) {
#pragma unroll
              for (int i = 0; i < 3; ++i)
                if (i < m) my[(id_up + i * nv * D + v * D + k) * TGM_STR] = val * ui[i];
            }
          };
          if constexpr (DC > 0) {
#pragma unroll
            for (int k = 0; k < DC; ++k) {
              put(k, q);
              const double qn = x2 * q - qm;
              qm = q;
              q = qn;
            }
          } else {
            for (int k = 0; k < D; ++k) {
              put(k, q);
              const double qn = x2 * q - qm;
              qm = q;
              q = qn;
            }
          }
        }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < m) my[(id_u + i) * TGM_STR] = ui[i];
      my[id_mask * TGM_STR] = okf;
    }
    // rotate the stages, refill the last one
#pragma unroll
    for (int st = 0; st + 1 < TGM_PD; ++st) {
      okq[st] = okq[st + 1];
#pragma unroll
      for (int v = 0; v < NVL; ++v) raw[st][v] = raw[st + 1][v];
#pragma unroll
      for (int i = 0; i < 3; ++i) rin[st][i] = rin[st + 1][i];
    }
    okq[TGM_PD - 1] = load_raw(rr + TGM_PD * TG_TS, raw[TGM_PD - 1], rin[TGM_PD - 1]);
    __syncthreads();
    // ---- 8 of the tile's 32 four-pair steps on this wave ----
    const char* tb = (const char*)(tab + bufsel * buf_doubles);
    {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double v6[6];
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
          double pr = *reinterpret_cast<const double*>(tb + off[sl][0] + j * 128);
#pragma unroll
          for (int f = 1; f < F; ++f) pr *= *reinterpret_cast<const double*>(tb + off[sl][f] + j * 128);
          v6[sl] = pr;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          accG[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(v6[g], v6[4], accG[g], 0, 0, 0);
          accC[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(v6[g], v6[5], accC[g], 0, 0, 0);
        }
      }
    }
  }
  // ---- the four waves' partial sums, added in a fixed order; D layout: row 4 g + (lane >> 4), column 4 blk + (lane & 3) ----
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    Gs[((wave * 2 + 0) * 16 + 4 * g + kq) * 16 + 4 * blk + li] = accG[g];
    Gs[((wave * 2 + 1) * 16 + 4 * g + kq) * 16 + 4 * blk + li] = accC[g];
  }
  __syncthreads();
  const int gi = tid >> 4, gj = tid & 15;
  if (gi < W && gj < W) {
    double sg = 0.0, sc = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sg += Gs[((w * 2 + 0) * 16 + gi) * 16 + gj];
      sc += Gs[((w * 2 + 1) * 16 + gi) * 16 + gj];
    }
    Gout[(size_t)sys * W * W + (size_t)gj * W + gi] = sg;
    Cout[(size_t)sys * W * W + (size_t)gj * W + gi] = sc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the Gram pass of the sweep's own shapes (1 state, 1 input; linear / bilinear / nonlinear polynomial dictionaries in
// the order of def_polyLift) with the COLUMNS of the rows in the table.  kp_traj_gram_mfma_kernel above ran at 0.37-0.45 of the
// f64 matrix rate with 1.3-2.1 VALU instructions beside every MFMA (rocprofv3: SQ_INSTS_VALU / SQ_INSTS_VALU_MFMA_MOPS_F64 =
// 2.3 | 2.4 | 3.1), and on this chip nothing on the VALU overlaps an f64 MFMA of the same SIMD.  Its ISA says where they came
// from: (1) the prefetch stages rotated through register moves - and the move out of the newest stage made every tile wait
// for the load issued one tile earlier (`s_waitcnt vmcnt(0)` at the top of the loop: the "4 tiles ahead" were 1); (2) LDS
// addresses re-formed per tile for the run-time buffer index and column count; (3) selects on the run-time input count;
// (4) for the nonlinear dictionary, every LANE multiplied two table entries per operand value, 48 v_mul_f64 per tile.
// Here: the thread that owns a (side, pair) writes the row's W <= 16 column values themselves (Chebyshev recurrence in
// registers, mixed monomials multiplied ONCE per pair), every MFMA operand is one ds_read with an immediate offset, the tile
// loop is unrolled over the four prefetch stages (fixed registers, no moves, loads really 4 tiles ahead; 6 KB of code), the
// buffer index and the dictionary are compile-time.  Left per tile and wave: the D recurrence steps, <= D products, a 5-
// instruction row index - against 64 MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
template <int MT, int D>
struct TgcShape {
  static constexpr int NV = MT == KP_MODEL_NONLINEAR ? 2 : 1;
  static constexpr int N = MT == KP_MODEL_NONLINEAR ? (D + 1) * (D + 2) / 2 : D + 1;        // dictionary functions, constant included
  static constexpr int W = MT == KP_MODEL_LINEAR ? N + 1 : MT == KP_MODEL_BILINEAR ? 2 * N : N;
  static constexpr int E = W < 16 ? W + 1 : W;                                                 // + one zero entry for the padding columns
};

template <int MT, int D>
__global__ __launch_bounds__(256, 4) void kp_traj_gram_cols_kernel(TrajView tv, int Ns, double* __restrict__ Gout, double* __restrict__ Cout) {
  using S = TgcShape<MT, D>;
  constexpr int W = S::W, E = S::E, N = S::N;
  static_assert(W <= 16, "one 16 x 16 tile");
  // ONE table buffer [side][entry][TGM_STR] and NO barrier in the tile loop: a wave owns pairs [32 wave, 32 wave + 32) of every
  // 128-pair tile - its 64 lanes write exactly the 2 x 32 (side, pair) rows its own eight 4-pair steps read, and the LDS
  // executes one wave's operations in order.  33 KB per workgroup: four workgroups (16 waves) per CU, the whole sweep resident.
  extern __shared__ __align__(16) double sm[];
  double* Gs = sm;                                    // the four waves' partial sums, over the table once it is dead
  const int tid = threadIdx.x, sys = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int side = lane >> 5, p = 32 * wave + (lane & 31);
  if (E > W)                                          // the zero entry of both sides
    for (int e = tid; e < 2 * TGM_STR; e += 256) sm[((e / TGM_STR) * E + W) * TGM_STR + e % TGM_STR] = 0.0;
  // ---- MFMA operands.  The four blocks of v_mfma_f64_4x4x4_4b take four different PAIR groups (block = pairs 4 blk .. 4 blk + 3
  // of a 16-pair step), not four column groups: one instruction then adds 16 pairs to ONE 4 x 4 output block (g, h), its four
  // blocks being partial sums that meet in the epilogue.  The A fragment of column group g and the B fragment of the same group
  // are the same register (lane (blk, k, i): psi[pair 4 blk + k][column 4 g + i]), so a 16-pair step reads the four groups of
  // psi_x and of psi_y ONCE - 8 ds_read_b64 for 26 MFMAs (10 for the upper triangle of G = Px'Px, 16 for C = Px'Py) where the
  // column-group form read 24 values for 32 MFMAs and kept the CU's LDS pipe as busy as its matrix pipes ----
  const int li = lane & 3, kq = lane >> 4, blk = (lane >> 2) & 3;
  // column groups 0-2 are always inside the row (W >= 12): base + immediate; group 3 may run into the padding (zero entry)
  // Which of the wave's 32 pairs a (step jj, block, k) slot takes is free (every pair once per tile): position
  // 16 (kq & 1) + 4 blk + 2 jj + (kq >> 1), so that the 32 lanes of a read pass (kq in {0, 1} or {2, 3}; entry stride = 1 mod 32
  // bank pairs) hit bank pairs li + 4 blk + 16 (kq & 1) + const - all different.  (With position 4 blk + kq the two k of a
  // pass overlapped on 15 of 16 bank pairs: rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25.)
  const int ppos = 32 * wave + 16 * (kq & 1) + 4 * blk + (kq >> 1);
  const int offa = li * TGM_STR + ppos;
  const int off3 = ((12 + li) < W ? 12 + li : W) * TGM_STR + ppos;
  static_assert(W >= 12, "column groups 0-2 full");
  double accG[10], accC[16];
#pragma unroll
  for (int k = 0; k < 10; ++k) accG[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) accC[k] = 0.0;
  // ---- raw values of this thread's (side, pair), four tiles ahead in FIXED registers (the loop below is unrolled over them) ----
  const int Tm1 = tv.T - 1;
  const double* py = tv.Y + (size_t)sys * tv.rows + side;      // side 1 lifts the successor row (Py)
  const double* pu = tv.U + (size_t)sys * tv.rows;
  // pair -> row = pair + pair / (T - 1), advanced tile by tile without a division: (trial, offset in trial)
  int tr = p / Tm1, rem = p - tr * Tm1, pair = p;
  const int q128 = TG_TS / Tm1, r128 = TG_TS - q128 * Tm1;
  constexpr int PD = 4;                                // tiles of raw values in flight (fixed registers; the mask is re-derived from the pair index)
  double ry[PD], ru[PD];
  auto load_next = [&](double& y, double& u) __attribute__((always_inline)) {
    const bool in = pair < Ns;
    const int row = in ? pair + tr : 0;
    y = py[row];
    u = pu[row];
    pair += TG_TS;
    tr += q128;
    rem += r128;
    if (rem >= Tm1) { rem -= Tm1; ++tr; }
  };
#pragma unroll
  for (int st = 0; st < PD; ++st) load_next(ry[st], ru[st]);
  __syncthreads();                                     // (the zero entries)
  double* const my = sm + side * E * TGM_STR + p;
  auto tile = [&](auto st_c) __attribute__((always_inline)) {
    constexpr int ST = decltype(st_c)::value;
    // ---- the row of this (side, pair): Chebyshev internal basis, T_e(x_v) where the monomial has x_v^e; every entry carries
    // the mask (0 for pairs past Ns), so that every product does ----
    {
      const double okf = (pair - PD * TG_TS) < Ns ? 1.0 : 0.0;      // `pair` runs PD tiles ahead of the tile being lifted
      const double x = ry[ST], uu = ru[ST] * okf, x2 = x + x;
      if constexpr (MT == KP_MODEL_LINEAR) {            // [T_1 .. T_D, 1, u]: written as the recurrence produces them
        double q = x * okf, qm = okf;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          my[k * TGM_STR] = q;
          const double qn = x2 * q - qm;
          qm = q;
          q = qn;
        }
        my[D * TGM_STR] = okf;
        my[(D + 1) * TGM_STR] = uu;
      } else if constexpr (MT == KP_MODEL_BILINEAR) {   // [psi, u psi], psi = [T_1 .. T_D, 1]
        double q = x * okf, qm = okf;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          my[k * TGM_STR] = q;
          my[(N + k) * TGM_STR] = q * uu;
          const double qn = x2 * q - qm;
          qm = q;
          q = qn;
        }
        my[D * TGM_STR] = okf;
        my[(N + D) * TGM_STR] = uu;
      } else {                                          // monomials of [y; u] by total degree, exponent of u ascending inside a degree
        double ty[D], tu[D];
        {
          double q = x * okf, qm = okf;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            ty[k] = q;
            const double qn = x2 * q - qm;
            qm = q;
            q = qn;
          }
        }
        {
          const double xu = ru[ST], xu2 = xu + xu;
          double q = xu * okf, qm = okf;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            tu[k] = q;
            const double qn = xu2 * q - qm;
            qm = q;
            q = qn;
          }
        }
        int c = 0;
#pragma unroll
        for (int d = 1; d <= D; ++d)
#pragma unroll
          for (int e2 = 0; e2 <= d; ++e2) {
            const int e1 = d - e2;
            my[c * TGM_STR] = e2 == 0 ? ty[e1 - 1] : e1 == 0 ? tu[e2 - 1] : ty[e1 - 1] * tu[e2 - 1];
            ++c;
          }
        my[c * TGM_STR] = okf;
      }
    }
    load_next(ry[ST], ru[ST]);                           // this stage's registers are free again: the tile PD ahead
    // ---- this wave's eight 4-pair steps (its own 32 pairs): 6 reads, 8 MFMAs each.  No barrier: the rows were written by
    // this wave, and a wave's LDS operations complete in order ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {                        // this wave's 32 pairs: two 16-pair steps
      double v[8];
#pragma unroll
      for (int sl = 0; sl < 8; ++sl)
        v[sl] = sm[((sl & 3) < 3 ? offa + 4 * (sl & 3) * TGM_STR : off3) + (sl >= 4 ? E * TGM_STR : 0) + jj * 2];
      int k = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = g; h < 4; ++h) {
          accG[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[g], v[h], accG[k], 0, 0, 0);
          ++k;
        }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) accC[4 * g + h] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[g], v[4 + h], accC[4 * g + h], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                 // one 16-pair step's operands in registers at a time (128 VGPRs: 4 workgroups per CU)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the reads above precede the next tile's writes of the same rows
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int ntiles = (Ns + TG_TS - 1) / TG_TS;
  for (int t = 0; t < ntiles; t += PD) {                 // tiles past Ns inside the last group are all mask: they add zeros
    tile(std::integral_constant<int, 0>{});
    tile(std::integral_constant<int, 1>{});
    tile(std::integral_constant<int, 2>{});
    tile(std::integral_constant<int, 3>{});
  }
  // ---- epilogue: the four blocks of an accumulator (lanes that differ in blk) are partial sums of one 4 x 4 output block -
  // added first (fixed order: xor 4, then xor 8), then the four waves' sums through LDS, in wave order.  Output block (g, h):
  // row 4 g + (lane >> 4), column 4 h + (lane & 3); G is mirrored from its upper blocks (exactly symmetric) ----
  auto blk_sum = [&](double x) __attribute__((always_inline)) {
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 8, 64);
    return x;
  };
  __syncthreads();                                     // the table is dead: Gs = [wave][26][16] doubles (13 KB)
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const double sg = blk_sum(accG[k]);
    if (blk == 0) Gs[(wave * 26 + k) * 16 + kq * 4 + li] = sg;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double sc = blk_sum(accC[k]);
    if (blk == 0) Gs[(wave * 26 + 10 + k) * 16 + kq * 4 + li] = sc;
  }
  __syncthreads();
  const int gi = tid >> 4, gj = tid & 15;                // output entry (row gi, column gj)
  if (gi < W && gj < W) {
    const int gq = gi >> 2, hq = gj >> 2;
    // G: block (g, h) with g <= h is stored; the entry below the block diagonal comes from the transposed block
    const int ga = gq <= hq ? gq : hq, gb = gq <= hq ? hq : gq;
    const int ri = gq <= hq ? gi & 3 : gj & 3, rj = gq <= hq ? gj & 3 : gi & 3;
    const int kg = ga * 4 - ga * (ga - 1) / 2 + (gb - ga);                  // index of (ga, gb) in the order g = 0..3, h = g..3
    double sg = 0.0, sc = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sg += Gs[(w * 26 + kg) * 16 + ri * 4 + rj];
      sc += Gs[(w * 26 + 10 + 4 * gq + hq) * 16 + (gi & 3) * 4 + (gj & 3)];
    }
    Gout[(size_t)sys * W * W + (size_t)gj * W + gi] = sg;
    Cout[(size_t)sys * W * W + (size_t)gj * W + gi] = sc;
  }
}

// the recipes def_polyLift gives a degree-D dictionary over nv <= 2 variables (what kp_traj_gram_cols_kernel hard-codes)
static bool tgc_canonical(const kp_basis* basis, int nv, int D) {
  if (!basis->fast || (int)basis->h_recipes.size() != basis->dev.nfull || basis->pow_depth != D) return false;
  std::vector<uint32_t> want;
  for (int d = 1; d <= D; ++d)
    for (int e2 = 0; e2 <= (nv == 2 ? d : 0); ++e2) {
      const int e1 = d - e2;
      uint32_t r = 0xffffffffu;
      int nf = 0;
      auto push = [&](int v, int e) { r = (r & ~(0xffu << (8 * nf))) | ((uint32_t)(v * D + e - 1) << (8 * nf)); ++nf; };
      if (e1 > 0) push(0, e1);
      if (e2 > 0) push(1, e2);
      want.push_back(r);
    }
  want.push_back(0xffffffffu);                         // the constant
  if (want.size() != basis->h_recipes.size()) return false;
  for (size_t i = 0; i < want.size(); ++i)
    if (want[i] != basis->h_recipes[i]) return false;
  return true;
}

// =====================================================================================================================
// All degrees of one model type from ONE pass over the data (evaluate_rand_models.m loops degree by degree, :47-143).
//  * The degree-j polynomial dictionary is a column subset of the degree-D one: def_polyLift orders the monomials by
//    total degree (Ksysid.m:645-648), so fullBasis_j = the first N_j - 1 columns of fullBasis_D plus the constant, and
//    Px_j'Px_j, Px_j'Py_j are sub-blocks of the degree-D Grams.
//  * The Grams are accumulated in a CHEBYSHEV internal basis (the data are scaled to [-1, 1], get_scale): psi_c has
//    T_e(x_v) where the monomial has x_v^e.  psi_m = S psi_c with an exactly known S (x^a = sum_b s1[a][b] T_b(x), all
//    coefficients positive: no cancellation), hence  K_m = T^-1 K_c T (T = S'),  G_m = S G_c S',  C_m = S C_c S'.
//    cond(Px_c) ~ 20-40 where cond(Px_m) ~ 1e5 for the degree-13 dictionaries, so the normal equations in the internal
//    basis give K_m to ~5e-12 of the QR solution that MATLAB's `\` (Ksysid.m:1069) returns - without the second sweep
//    over the snapshots that iterative refinement needs.
// =====================================================================================================================
struct SweepDeg {
  int N, W, pad0, pad1;
  int idx[16];          // column of the degree-D Px for each column of the degree-j Px
  double Sf[256];       // [r * 16 + c]: psi_m = Sf psi_c (identity on the padding)
  double Tinv[256];     // (Sf')^-1
};

__global__ __launch_bounds__(256) void kp_sweep_sub_kernel(const double* __restrict__ Gc, const double* __restrict__ Cc, int Wmax, int nb,
                                                           const SweepDeg* __restrict__ degs, const int* __restrict__ fit_status,
                                                           double* __restrict__ Kall, double* __restrict__ Gall,
                                                           double* __restrict__ Call, int* __restrict__ stat) {
  __shared__ double Gs[16 * SB_LD], Cs[16 * SB_LD], Xs[16 * SB_LD], Ls[16 * SB_LD], Dd[16], T1[16 * SB_LD], T2[16 * SB_LD], Sf[16 * SB_LD],
      Ti[16 * SB_LD];
  __shared__ int bad;
  const int tid = threadIdx.x, i = tid >> 4, j = tid & 15, sys = blockIdx.x, dj = blockIdx.y;
  const SweepDeg& d = degs[dj];
  const int W = d.W;
  const double* G0 = Gc + (size_t)sys * Wmax * Wmax;
  const double* C0 = Cc + (size_t)sys * Wmax * Wmax;
  const bool in = i < W && j < W;
  Gs[i * SB_LD + j] = in ? G0[d.idx[i] + (size_t)d.idx[j] * Wmax] : (i == j ? 1.0 : 0.0);
  Cs[i * SB_LD + j] = in ? C0[d.idx[i] + (size_t)d.idx[j] * Wmax] : 0.0;
  Sf[i * SB_LD + j] = d.Sf[i * 16 + j];
  Ti[i * SB_LD + j] = d.Tinv[i * 16 + j];
  if (tid == 0) bad = 0;
  __syncthreads();
  sb_spd_solve16(Gs, Cs, Xs, Ls, Dd, &bad);                       // Xs = K_c
  double a = 0.0, g = 0.0, c = 0.0;                                // X T, G S', C S'  (T[k][j] = Sf[j][k])
  for (int k = 0; k < 16; ++k) {
    const double sjk = Sf[j * SB_LD + k];
    a += Xs[i * SB_LD + k] * sjk;
    g += Gs[i * SB_LD + k] * sjk;
    c += Cs[i * SB_LD + k] * sjk;
  }
  __syncthreads();
  T1[i * SB_LD + j] = a; T2[i * SB_LD + j] = g; Ls[i * SB_LD + j] = c;
  __syncthreads();
  double km = 0.0, gm = 0.0, cm = 0.0;
  for (int k = 0; k < 16; ++k) {
    km += Ti[i * SB_LD + k] * T1[k * SB_LD + j];
    gm += Sf[i * SB_LD + k] * T2[k * SB_LD + j];
    cm += Sf[i * SB_LD + k] * Ls[k * SB_LD + j];
  }
  const size_t o = ((size_t)dj * nb + sys) * 256;
  const int isbad = bad || (fit_status && fit_status[sys]);
  if (in) {
    Kall[o + (size_t)j * W + i] = isbad ? __builtin_nan("") : km;
    Gall[o + (size_t)j * W + i] = gm;
    Call[o + (size_t)j * W + i] = cm;
  }
  if (tid == 0) stat[(size_t)dj * nb + sys] = isbad;
}

// kp_small_project_kernel for every (system, degree) of a nested linear sweep
__global__ __launch_bounds__(256) void kp_sweep_project_kernel(const double* __restrict__ Kall, const double* __restrict__ Gall,
                                                               const double* __restrict__ Call, int nb, int m, const SweepDeg* __restrict__ degs,
                                                               double* __restrict__ Aall, double* __restrict__ Ball) {
  __shared__ double Ks[16 * SB_LD], Gs[16 * SB_LD], Cs[16 * SB_LD], Ts[16 * SB_LD], LL[16 * SB_LD], LR[16 * SB_LD], Ms[16 * SB_LD], Ls[16 * SB_LD], Dd[16];
  __shared__ int bad;
  const int tid = threadIdx.x, i = tid >> 4, j = tid & 15, sys = blockIdx.x, dj = blockIdx.y;
  const int N = degs[dj].N, W = N + m;
  const size_t off = ((size_t)dj * nb + sys) * 256;
  const bool in = i < W && j < W;
  Ks[i * SB_LD + j] = in ? Kall[off + (size_t)j * W + i] : 0.0;
  Gs[i * SB_LD + j] = in ? Gall[off + (size_t)j * W + i] : 0.0;
  Cs[i * SB_LD + j] = in ? Call[off + (size_t)j * W + i] : 0.0;
  if (tid == 0) bad = 0;
  __syncthreads();
  double t = 0.0;
  if (j < N)
    for (int w = 0; w < W; ++w) t += Gs[i * SB_LD + w] * Ks[w * SB_LD + j];
  Ts[i * SB_LD + j] = t;
  __syncthreads();
  double ll = 0.0, lr = 0.0;
  if (i < N && j < N)
    for (int w = 0; w < W; ++w) {
      ll += Ks[w * SB_LD + i] * Ts[w * SB_LD + j];
      lr += Ks[w * SB_LD + i] * Cs[w * SB_LD + j];
    }
  LL[i * SB_LD + j] = (i < N && j < N) ? ll : (i == j ? 1.0 : 0.0);
  LR[i * SB_LD + j] = (i < N && j < N) ? lr : 0.0;
  __syncthreads();
  sb_spd_solve16(LL, LR, Ms, Ls, Dd, &bad);
  const double nanv = __builtin_nan("");
  double a = 0.0, b = 0.0;
  if (i < N)
    for (int k = 0; k < N; ++k) {
      const double mik = Ms[k * SB_LD + i];
      if (j < N) a += mik * Ks[j * SB_LD + k];
      if (j < m) b += mik * Ks[(N + j) * SB_LD + k];
    }
  if (i < N && j < N) Aall[off + (size_t)j * N + i] = bad ? nanv : a;
  if (i < N && j < m) Ball[off + (size_t)j * N + i] = bad ? nanv : b;
}

__global__ __launch_bounds__(64) void kp_sweep_rollout_nested_kernel(BasisDev b, const SweepDeg* __restrict__ degs, const ColDesc* __restrict__ cols_all,
                                                                     int nb, const double* __restrict__ Kall, const double* __restrict__ Aall,
                                                                     const double* __restrict__ Ball, const double* __restrict__ Yv,
                                                                     const double* __restrict__ Uv, int Tv, const int* __restrict__ stat,
                                                                     double* __restrict__ err) {
  __shared__ double vsh[KP_MAX_VARS];
  const int sys = blockIdx.x, dj = blockIdx.y, n = b.nzeta, m = b.m;
  BasisDev bj = b;
  bj.N = degs[dj].N; bj.W = degs[dj].W; bj.nfull = bj.N; bj.cols = cols_all + (size_t)dj * 16;
  const size_t off = ((size_t)dj * nb + sys) * 256;
  sweep_rollout_body(bj, Kall + off, Aall ? Aall + off : nullptr, Ball ? Ball + off : nullptr, Yv + (size_t)sys * Tv * n, Uv + (size_t)sys * Tv * m,
                     Tv, stat[(size_t)dj * nb + sys], err + ((size_t)dj * nb + sys) * n, vsh);
}

// Validation rollouts of the nested sweep, FOUR (system, degree) jobs per wave: a job lives in one 16-lane row (N <= 16),
// lane r of the row owns component r of the lifted state.  z+ = A z column by column: z[c] is broadcast inside every row by
// a DPP row_newbcast move and lane r adds A[r][c] z[c] - N moves + N FMAs per step for all four jobs, instead of sixteen
// v_readlane broadcasts per job.
template <int CTRL>
__device__ __forceinline__ double sb_row_ror(double v) {       // lane l of a 16-lane row receives from lane (l - n) mod 16
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sb_row_sum(double v) {        // sum over the 16 lanes of a row, in every lane
  v += sb_row_ror<0x128>(v);
  v += sb_row_ror<0x124>(v);
  v += sb_row_ror<0x122>(v);
  v += sb_row_ror<0x121>(v);
  return v;
}

// MT: the model type at compile time - the row of B_i (96 registers) exists only in the bilinear instantiation; with the type
// a run-time value every instantiation carried it, 2 waves per SIMD fitted and the 3328 waves of the linear pass (13 degrees
// x 1024 systems) ran in two rounds.
template <int MT>
__global__ __launch_bounds__(64) void kp_sweep_rollout4_kernel(BasisDev b, const SweepDeg* __restrict__ degs, const ColDesc* __restrict__ cols_all,
                                                               int nb, int n_deg, const double* __restrict__ Kall, const double* __restrict__ Aall,
                                                               const double* __restrict__ Ball, const double* __restrict__ Yv,
                                                               const double* __restrict__ Uv, int Tv, const int* __restrict__ stat,
                                                               double* __restrict__ err) {
  __shared__ double vsh[4][KP_MAX_VARS];
  const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
  const int njobs = nb * n_deg;
  int job = blockIdx.x * 4 + q;
  const bool live = job < njobs;
  if (!live) job = njobs - 1;
  const int dj = job / nb, sys = job - dj * nb;
  const int n = b.nzeta, m = b.m;
  constexpr int mt = MT;
  const int N = degs[dj].N, W = degs[dj].W;
  BasisDev bj = b;
  bj.N = N; bj.W = W; bj.nfull = N; bj.cols = cols_all + (size_t)dj * 16;
  const size_t off = ((size_t)dj * nb + sys) * 256;
  const double* Ks = Kall + off;
  const double* yv = Yv + (size_t)sys * Tv * n;
  const double* uv = Uv + (size_t)sys * Tv * m;
  const bool rin = r < N;
  // row r of the model matrices: ak[c] = A[r][c], bk[i][c] = B_i[r][c]
  constexpr int NBK = MT == KP_MODEL_BILINEAR ? 3 : 1;      // (a zero-length array is not allowed)
  double ak[16], bk[NBK][16], bl[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const bool in = rin && c < N;
    double a = 0.0;
    if (mt == KP_MODEL_LINEAR) a = in ? Aall[off + (size_t)c * N + r] : 0.0;
    else if (mt == KP_MODEL_BILINEAR) a = in ? Ks[c + (size_t)r * W] : 0.0;
    ak[c] = a;
    if constexpr (MT == KP_MODEL_BILINEAR) {
#pragma unroll
      for (int i = 0; i < 3; ++i) bk[i][c] = (i < m && in) ? Ks[N + N * i + c + (size_t)r * W] : 0.0;
    } else {
      bk[0][c] = 0.0;
    }
  }
  // columns to visit: the widest of the wave's four jobs (consecutive jobs share their degree except at a boundary)
  int Nw = N;
  Nw = max(Nw, __builtin_amdgcn_readlane(N, 16));
  Nw = max(Nw, __builtin_amdgcn_readlane(N, 32));
  Nw = max(Nw, __builtin_amdgcn_readlane(N, 48));
  Nw = __builtin_amdgcn_readfirstlane(max(Nw, __builtin_amdgcn_readlane(N, 0)));
  if (mt == KP_MODEL_LINEAR)
#pragma unroll
    for (int i = 0; i < 3; ++i) bl[i] = (i < m && rin) ? Ball[off + (size_t)i * N + r] : 0.0;
  double kf[4] = {0.0, 0.0, 0.0, 0.0};             // nonlinear: lane c keeps column c of Kf = K(:, 1:n)' (n <= 4 outputs)
  if (mt == KP_MODEL_NONLINEAR)
#pragma unroll
    for (int o = 0; o < 4; ++o) kf[o] = (o < n && rin) ? Ks[r + (size_t)o * W] : 0.0;
  // nonlinear: column r of the dictionary over [zeta; u] as packed exponents (4 bits per variable, <= 8 variables), so that a
  // step evaluates psi from registers - zeta_i by a DPP broadcast from lane i of the row, u_i from the lane's own copy -
  // instead of kp_eval_col's exponent bytes from global memory and the trip through LDS (two wave barriers per step)
  unsigned epack = 0;
  bool nl_fast = false;
  if constexpr (MT == KP_MODEL_NONLINEAR) {
    bool ok = n <= 4 && m <= 3;
    if (rin && ok) {
      const ColDesc c = bj.cols[r];
      if (c.kind == COL_VAR) epack = 1u << (4 * c.arg);
      else if (c.kind == COL_MONO) {
        for (int i = 0; i < bj.nvars; ++i) {
          const unsigned e = bj.exps[(size_t)c.arg * bj.nvars + i];
          ok = ok && e < 16;
          epack |= (e & 15u) << (4 * i);
        }
      } else if (c.kind != COL_CONST) ok = false;
    }
    nl_fast = __all(ok);
  }
  // first validation row -> lifted state
  if (r < n) vsh[q][r] = yv[(size_t)r * Tv];
  if (mt == KP_MODEL_NONLINEAR && r < m) vsh[q][n + r] = uv[(size_t)r * Tv];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double z = 0.0;
  if (mt != KP_MODEL_NONLINEAR) { if (rin) z = kp_eval_col(bj, bj.cols[r], vsh[q], 1); }
  else if (r < n) z = vsh[q][r];
  double acc_e = 0.0, acc_a = 0.0;
  double yr = r < n ? yv[(size_t)r * Tv] : 0.0;
  double ut[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ut[i] = i < m ? uv[(size_t)i * Tv] : 0.0;
  // The time loop with the column count of the wave's widest job as a compile-time constant: with `if (c < Nw)` around every
  // column the 16 scalar compare-and-branch pairs of a step cost a lone wave more than its arithmetic (0.4 us per step at
  // degree 13).  (Also measured: the trial fetched in 4-step register chunks a chunk ahead, or staged in LDS in 16-step
  // chunks, instead of the one-step-ahead loads below - no faster: the step time is the dependent chain, not the loads.)
  auto run = [&](auto nwc) {
    constexpr int NW = decltype(nwc)::value;
    for (int t = 0; t < Tv; ++t) {
      if (r < n) {
        if (t > 0) acc_e += fabs(z - yr);               // the first simulated row is the measured one (Ksysid.m:1654)
        acc_a += fabs(yr);
      }
      if (t == Tv - 1) break;
      const double yr_n = r < n ? yv[(size_t)r * Tv + t + 1] : 0.0;      // values of the next step: in flight during this one
      double ut_n[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) ut_n[i] = i < m ? uv[(size_t)i * Tv + t + 1] : 0.0;
      double zn = 0.0;
      if constexpr (MT != KP_MODEL_NONLINEAR) {
        // z+ = (A + sum_i u_i B_i) z  (val_model :1685 / val_BLmodel :1783); linear: B u added below.  z[c] reaches every
        // lane of its 16-lane row by ONE DP-ALU DPP move (v_mov_b64_dpp row_newbcast:c).  (Round 2 broadcast it with two
        // ds_swizzle per column: with 13 waves per CU the LDS crossbar was the bound.)  Two partial sums: half the chain.
        double zn1 = 0.0;
#define KP_COL_STEP(CC, ACC)                                                                                   \
        if constexpr (CC < NW) {                                                                                 \
          double w = ak[CC];                                                                                     \
          if constexpr (MT == KP_MODEL_BILINEAR) { w += ut[0] * bk[0][CC]; if (m > 1) w += ut[1] * bk[1][CC]; if (m > 2) w += ut[2] * bk[2][CC]; } \
          ACC += w * __builtin_amdgcn_update_dpp(0.0, z, 0x150 + (CC), 0xf, 0xf, true);                         \
        }
        KP_COL_STEP(0, zn) KP_COL_STEP(1, zn1) KP_COL_STEP(2, zn) KP_COL_STEP(3, zn1) KP_COL_STEP(4, zn) KP_COL_STEP(5, zn1)
        KP_COL_STEP(6, zn) KP_COL_STEP(7, zn1) KP_COL_STEP(8, zn) KP_COL_STEP(9, zn1) KP_COL_STEP(10, zn) KP_COL_STEP(11, zn1)
        KP_COL_STEP(12, zn) KP_COL_STEP(13, zn1) KP_COL_STEP(14, zn) KP_COL_STEP(15, zn1)
#undef KP_COL_STEP
        zn += zn1;
        if constexpr (MT == KP_MODEL_LINEAR) zn += bl[0] * ut[0] + bl[1] * ut[1] + bl[2] * ut[2];
      } else {                                           // zeta+ = Kf psi([zeta; u]) (val_NLmodel, Ksysid.m:1848-1863)
        double psi;
        if (nl_fast) {
          psi = rin ? 1.0 : 0.0;
#define KP_NL_VAR(I, X)                                                                  \
          {                                                                               \
            const double x_ = X;                                                          \
            const int e_ = (int)((epack >> (4 * (I))) & 15u);                             \
            for (int k_ = 0; k_ < e_; ++k_) psi *= x_;                                    \
          }
          KP_NL_VAR(0, __builtin_amdgcn_update_dpp(0.0, z, 0x150, 0xf, 0xf, true))
          if (n > 1) KP_NL_VAR(1, __builtin_amdgcn_update_dpp(0.0, z, 0x151, 0xf, 0xf, true))
          if (n > 2) KP_NL_VAR(2, __builtin_amdgcn_update_dpp(0.0, z, 0x152, 0xf, 0xf, true))
          if (n > 3) KP_NL_VAR(3, __builtin_amdgcn_update_dpp(0.0, z, 0x153, 0xf, 0xf, true))
          KP_NL_VAR(n, ut[0])
          if (m > 1) KP_NL_VAR(n + 1, ut[1])
          if (m > 2) KP_NL_VAR(n + 2, ut[2])
#undef KP_NL_VAR
        } else {
          if (r < n) vsh[q][r] = z;
          if (r < m) vsh[q][n + r] = ut[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          psi = rin ? kp_eval_col(bj, bj.cols[r], vsh[q], 1) : 0.0;
          __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < n) {
            const double sacc = sb_row_sum(kf[o] * psi);
            if (r == o) zn = sacc;
          }
      }
      z = zn;
      yr = yr_n;
#pragma unroll
      for (int i = 0; i < 3; ++i) ut[i] = ut_n[i];
    }
  };
  if constexpr (MT == KP_MODEL_NONLINEAR) {
    run(std::integral_constant<int, 0>{});
  } else {
    switch (Nw) {
#define KP_NW_CASE(V) case V: run(std::integral_constant<int, V>{}); break;
      KP_NW_CASE(1) KP_NW_CASE(2) KP_NW_CASE(3) KP_NW_CASE(4) KP_NW_CASE(5) KP_NW_CASE(6) KP_NW_CASE(7) KP_NW_CASE(8)
      KP_NW_CASE(9) KP_NW_CASE(10) KP_NW_CASE(11) KP_NW_CASE(12) KP_NW_CASE(13) KP_NW_CASE(14) KP_NW_CASE(15)
#undef KP_NW_CASE
      default: run(std::integral_constant<int, 16>{}); break;
    }
  }
  if (live && r < n) {
    const double bad = stat[(size_t)dj * nb + sys] ? __builtin_nan("") : 0.0;
    err[((size_t)dj * nb + sys) * n + r] = (acc_e / Tv) / (acc_a / Tv) + bad;
  }
}

__global__ void kp_l1_flag_all_kernel(const double* __restrict__ Kall, const SweepDeg* __restrict__ degs, int nb, double lasso, int* __restrict__ flags) {
  const int sys = blockIdx.x, dj = blockIdx.y, W = degs[dj].W;
  const double* K = Kall + ((size_t)dj * nb + sys) * 256;
  double s = 0.0;
  for (int e = threadIdx.x; e < W * W; e += 64) s += fabs(K[e]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) flags[(size_t)dj * nb + sys] = s > lasso * degs[dj].N ? 1 : 0;
}

// 1-D monomial -> Chebyshev coefficients: x^a = sum_b s1[a][b] T_b(x)   (x T_b = (T_(b+1) + T_|b-1|) / 2, x T_0 = T_1)
static void cheb_table(int dmax, std::vector<std::vector<double>>& s1) {
  s1.assign(dmax + 1, std::vector<double>(dmax + 1, 0.0));
  s1[0][0] = 1.0;
  for (int a = 0; a < dmax; ++a)
    for (int bq = 0; bq <= a; ++bq) {
      const double c = s1[a][bq];
      if (c == 0.0) continue;
      if (bq == 0) s1[a + 1][1] += c;
      else { s1[a + 1][bq + 1] += 0.5 * c; s1[a + 1][bq - 1] += 0.5 * c; }
    }
}

extern "C" int kp_sweep_eval_nested(kp_ctx* ctx, const kp_traj* traj, const kp_basis* basis, double lasso, int n_deg, double* err_out,
                                    int* status_out) {
  if (!ctx || !traj || !basis || !err_out || n_deg < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: bad argument") : KP_ERR_ARG;
  if (traj->have != 31) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: trajectory object not finished (kp_traj_finish)");
  const BasisDev& b = basis->dev;
  if (b.nzeta != traj->n || b.m != traj->m) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: trajectory / dictionary dimension mismatch");
  if (b.W > SB_W || b.k_pcs != 0 || b.N != b.nfull || b.m > 3 || !basis->fast || basis->h_recipes.size() != (size_t)b.nfull)
    return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: needs a polynomial dictionary with W <= 16, m <= 3 and no dimension reduction");
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  const int nb = traj->nb, Wmax = b.W, Nmax = b.N, n = traj->n, m = traj->m, nv = b.nvars, Dp = basis->pow_depth;
  // exponent multi-index and total degree of every column (from the power-table recipes; the constant is all zeros)
  std::vector<std::vector<int>> ex(Nmax, std::vector<int>(nv, 0));
  std::vector<int> deg(Nmax, 0);
  for (int c = 0; c < Nmax; ++c) {
    const uint32_t r = basis->h_recipes[c];
    for (int f = 0; f < 4; ++f) {
      const uint32_t id = (r >> (8 * f)) & 255u;
      if (id == 255u) continue;
      ex[c][id / Dp] += (int)(id % Dp) + 1;
    }
    for (int v = 0; v < nv; ++v) deg[c] += ex[c][v];
  }
  if (deg[Nmax - 1] != 0) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: the last dictionary column must be the constant");
  for (int c = 0; c + 2 < Nmax; ++c)
    if (deg[c] > deg[c + 1] || deg[c] < 1) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: monomials must be ordered by total degree");
  const int dmax = deg[Nmax - 2];
  if (n_deg > dmax) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: n_deg exceeds the degree of the dictionary");
  std::vector<std::vector<double>> s1;
  cheb_table(dmax, s1);
  std::vector<SweepDeg> degs(n_deg);
  std::vector<ColDesc> cols_all((size_t)n_deg * 16, ColDesc{COL_CONST, 0, 0, 0});
  for (int dj = 0; dj < n_deg; ++dj) {
    const int j = dj + 1;
    SweepDeg& d = degs[dj];
    std::vector<int> psi;                            // columns of psi_D that make psi_j: degree <= j, then the constant
    for (int c = 0; c + 1 < Nmax; ++c)
      if (deg[c] <= j) psi.push_back(c);
    psi.push_back(Nmax - 1);
    const int Nj = (int)psi.size();
    d.N = Nj;
    d.W = b.model_type == KP_MODEL_LINEAR ? Nj + m : b.model_type == KP_MODEL_BILINEAR ? Nj * (m + 1) : Nj;
    d.pad0 = d.pad1 = 0;
    // S_psi (Nj x Nj) and its layout inside Sf
    std::vector<long double> Sp((size_t)Nj * Nj, 0.0L);
    for (int r = 0; r < Nj; ++r)
      for (int c = 0; c < Nj; ++c) {
        long double p = 1.0L;
        for (int v = 0; v < nv; ++v) {
          const int a = ex[psi[r]][v], bq = ex[psi[c]][v];
          p *= bq <= a ? (long double)s1[a][bq] : 0.0L;
        }
        Sp[(size_t)r * Nj + c] = p;
      }
    std::vector<long double> Sf((size_t)256, 0.0L);
    for (int q = 0; q < 16; ++q) Sf[q * 16 + q] = 1.0L;
    for (int q = 0; q < 16; ++q) d.idx[q] = 0;
    const int nblk = b.model_type == KP_MODEL_BILINEAR ? m + 1 : 1;
    for (int blk = 0; blk < nblk; ++blk)
      for (int r = 0; r < Nj; ++r) {
        d.idx[blk * Nj + r] = blk * Nmax + psi[r];
        for (int c = 0; c < Nj; ++c) Sf[(size_t)(blk * Nj + r) * 16 + blk * Nj + c] = Sp[(size_t)r * Nj + c];
      }
    if (b.model_type == KP_MODEL_LINEAR)
      for (int i = 0; i < m; ++i) d.idx[Nj + i] = Nmax + i;
    // Tinv = (Sf')^-1 by Gauss-Jordan in extended precision (Sf is triangular up to the column order, diagonal 2^(1-deg))
    std::vector<long double> A((size_t)256), I((size_t)256, 0.0L);
    for (int r = 0; r < 16; ++r)
      for (int c = 0; c < 16; ++c) A[r * 16 + c] = Sf[c * 16 + r];
    for (int q = 0; q < 16; ++q) I[q * 16 + q] = 1.0L;
    for (int k = 0; k < 16; ++k) {
      int piv = k;
      for (int r = k + 1; r < 16; ++r)
        if (fabsl(A[r * 16 + k]) > fabsl(A[piv * 16 + k])) piv = r;
      if (A[piv * 16 + k] == 0.0L) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: singular basis change");
      for (int c = 0; c < 16; ++c) { std::swap(A[k * 16 + c], A[piv * 16 + c]); std::swap(I[k * 16 + c], I[piv * 16 + c]); }
      const long double inv = 1.0L / A[k * 16 + k];
      for (int c = 0; c < 16; ++c) { A[k * 16 + c] *= inv; I[k * 16 + c] *= inv; }
      for (int r = 0; r < 16; ++r)
        if (r != k && A[r * 16 + k] != 0.0L) {
          const long double f = A[r * 16 + k];
          for (int c = 0; c < 16; ++c) { A[r * 16 + c] -= f * A[k * 16 + c]; I[r * 16 + c] -= f * I[k * 16 + c]; }
        }
    }
    for (int q = 0; q < 256; ++q) { d.Sf[q] = (double)Sf[q]; d.Tinv[q] = (double)I[q]; }
    // column descriptors of psi_j for the rollout's lift
    for (int r = 0; r < Nj; ++r) {
      const int c = psi[r];
      ColDesc cd{COL_CONST, 0, 0, 0};
      if (c < nv) cd = ColDesc{COL_VAR, c, 0, 0};
      else if (c + 1 < Nmax) cd = ColDesc{COL_MONO, c - nv, 0, 0};
      cols_all[(size_t)dj * 16 + r] = cd;
    }
  }
  // device buffers: Gc, Cc, Kc (degree D), per (degree, system) K_m, G_m, C_m, A, B (256 doubles each), errors, flags, tables
  const size_t bW = (size_t)nb * Wmax * Wmax * 8, bAll = (size_t)n_deg * nb * 256 * 8;
  const size_t b_tab = (size_t)n_deg * sizeof(SweepDeg), b_cols = cols_all.size() * sizeof(ColDesc);
  char* ws = (char*)ctx->workspace(6, 3 * bW + 5 * bAll + (size_t)n_deg * nb * (n * 8 + 8) + (size_t)nb * 4 + b_tab + b_cols + 1024);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_sweep_eval_nested: out of device memory");
  char* q = ws;
  double* dKc = (double*)q; q += bW;
  double* dGc = (double*)q; q += bW;
  double* dCc = (double*)q; q += bW;
  double* dK = (double*)q; q += bAll;
  double* dG = (double*)q; q += bAll;
  double* dC = (double*)q; q += bAll;
  double* dA = (double*)q; q += bAll;
  double* dB = (double*)q; q += bAll;
  double* dE = (double*)q; q += (size_t)n_deg * nb * n * 8;
  int* dS0 = (int*)q; q += (size_t)nb * 4;
  int* dS = (int*)q; q += (size_t)n_deg * nb * 4;
  int* dF = (int*)q; q += (size_t)n_deg * nb * 4;
  q = (char*)(((uintptr_t)q + 63) & ~(uintptr_t)63);
  SweepDeg* dT = (SweepDeg*)q; q += b_tab;
  q = (char*)(((uintptr_t)q + 63) & ~(uintptr_t)63);
  ColDesc* dCols = (ColDesc*)q;
  hipStream_t s = ctx->stream;
  KP_HIP(ctx, hipMemcpyAsync(dT, degs.data(), b_tab, hipMemcpyHostToDevice, s));
  KP_HIP(ctx, hipMemcpyAsync(dCols, cols_all.data(), b_cols, hipMemcpyHostToDevice, s));
  const int Ns = traj->ntrials * (traj->T - 1) - 1;
  if (!basis->d_recipes || nv > 8) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: dictionary not supported by the power-table lift");
  const size_t lds = ((size_t)2 * TG_TS * SB_LD + (size_t)2 * nv * Dp * TG_TS + 2 * 16 * SB_LD) * sizeof(double);
  // the matrix-pipe form: factors per operand value = monomial factors (+ the input of a bilinear column); two table buffers
  const int nfac0 = basis->max_factors > 0 ? basis->max_factors : 1;
  const size_t lds_up = (size_t)2 * 2 * (nv * Dp * (1 + m) + m + 3) * TGM_STR * sizeof(double);
  const int upro = b.model_type == KP_MODEL_BILINEAR && lds_up <= 78 * 1024;        // input products in the table
  const int nfac = nfac0 + (b.model_type == KP_MODEL_BILINEAR && !upro ? 1 : 0);
  const size_t lds_m = std::max(upro ? lds_up : (size_t)2 * 2 * (nv * Dp + m + 3) * TGM_STR * sizeof(double), (size_t)4 * 2 * 256 * sizeof(double));
  static const bool gram_old = getenv("KP_SWEEP_GRAM_OLD") != nullptr;
  const bool use_mfma = !gram_old && nfac <= 5 && nv <= TGM_NV && lds_m <= 78 * 1024;      // two workgroups per CU
  if (!use_mfma && lds > 150 * 1024) return ctx->fail(KP_ERR_ARG, "kp_sweep_eval_nested: dictionary too large for the power-table lift");
  KP_HIP(ctx, hipMemsetAsync(dS0, 0, (size_t)nb * 4, s));
  // round 4: the sweep's own shapes (1 state, 1 input, canonical polynomial dictionary) with the row's columns in the table
  static const bool cols_off = getenv("KP_SWEEP_GRAM_V1") != nullptr;
  bool use_cols = false;
  if (!cols_off && !gram_old && n == 1 && m == 1 && b.nzeta == 1 && b.k_pcs == 0 && tgc_canonical(basis, nv, Dp)) {
#define KP_TGC(MT_, D_)                                                                                                           \
    {                                                                                                                             \
      static KpLdsCache tgc_lds;                                                                                                  \
      const size_t lds_c = std::max((size_t)2 * TgcShape<MT_, D_>::E * TGM_STR * sizeof(double), (size_t)4 * 26 * 16 * sizeof(double)); \
      KP_HIP(ctx, kp_ensure_lds(tgc_lds, (const void*)kp_traj_gram_cols_kernel<MT_, D_>, lds_c));                                 \
      KP_HIP(ctx, hipEventRecord(ctx->ev0, s));                                                                                   \
      hipLaunchKernelGGL((kp_traj_gram_cols_kernel<MT_, D_>), dim3(nb), dim3(256), lds_c, s, traj_view(traj), Ns, dGc, dCc);      \
      use_cols = true;                                                                                                            \
      ctx->timers[10] = 26.0 * 512.0 / 16.0;   /* executed on the matrix pipe per pair: 10 + 16 MFMAs per 16 pairs */            \
    }
    if (b.model_type == KP_MODEL_LINEAR && Dp == 13) KP_TGC(KP_MODEL_LINEAR, 13)
    else if (b.model_type == KP_MODEL_BILINEAR && Dp == 6) KP_TGC(KP_MODEL_BILINEAR, 6)
    else if (b.model_type == KP_MODEL_NONLINEAR && Dp == 4) KP_TGC(KP_MODEL_NONLINEAR, 4)
#undef KP_TGC
  }
  if (use_cols) {
  } else if (use_mfma) {
    ctx->timers[10] = 8.0 * 512.0 / 4.0;         // 8 MFMAs per 4 pairs (one padded 16 x 16 tile for G and for C)
#define KP_TGM(F_, DC_, NV_, UP_)                                                                                               \
    {                                                                                                                             \
      static KpLdsCache tgm_lds;                                                                                                  \
      KP_HIP(ctx, kp_ensure_lds(tgm_lds, (const void*)kp_traj_gram_mfma_kernel<F_, DC_, NV_, UP_>, lds_m));                       \
      KP_HIP(ctx, hipEventRecord(ctx->ev0, s));                                                                                   \
      hipLaunchKernelGGL((kp_traj_gram_mfma_kernel<F_, DC_, NV_, UP_>), dim3(nb), dim3(256), lds_m, s, b,                         \
                         (const uint32_t*)basis->d_recipes, Dp, traj_view(traj), Ns, dGc, dCc);                                   \
    }
    // the shapes of evaluate_rand_models.m (1-D systems: linear degree 13, bilinear 6, nonlinear 4) with the table depth and
    // the variable count at compile time; everything else with run-time values
    if (nfac == 1 && Dp == 13 && nv == 1 && !upro) KP_TGM(1, 13, 1, false)
    else if (nfac == 1 && Dp == 6 && nv == 1 && upro) KP_TGM(1, 6, 1, true)
    else if (nfac == 2 && Dp == 4 && nv == 2 && !upro) KP_TGM(2, 4, 2, false)
    else if (upro) switch (nfac) {
      case 1: KP_TGM(1, 0, 0, true) break;
      case 2: KP_TGM(2, 0, 0, true) break;
      case 3: KP_TGM(3, 0, 0, true) break;
      default: KP_TGM(4, 0, 0, true) break;
    }
    else switch (nfac) {
      case 1: KP_TGM(1, 0, 0, false) break;
      case 2: KP_TGM(2, 0, 0, false) break;
      case 3: KP_TGM(3, 0, 0, false) break;
      case 4: KP_TGM(4, 0, 0, false) break;
      default: KP_TGM(5, 0, 0, false) break;
    }
#undef KP_TGM
  } else {
    static KpLdsCache tg_lds;
    KP_HIP(ctx, kp_ensure_lds(tg_lds, (const void*)kp_traj_gram_kernel, lds));
    KP_HIP(ctx, hipEventRecord(ctx->ev0, s));
    hipLaunchKernelGGL(kp_traj_gram_kernel, dim3(nb), dim3(256), lds, s, b, (const uint32_t*)basis->d_recipes, Dp, basis->max_factors > 0 ? basis->max_factors : 1,
                       traj_view(traj), Ns, 1, dGc, dCc);
  }
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, s));
  (void)dKc;
  const dim3 grid(nb, n_deg);
  hipLaunchKernelGGL(kp_sweep_sub_kernel, grid, dim3(256), 0, s, dGc, dCc, Wmax, nb, dT, (const int*)nullptr, dK, dG, dC, dS);
  KP_HIP(ctx, hipGetLastError());
  if (lasso < 1e6) {
    hipLaunchKernelGGL(kp_l1_flag_all_kernel, grid, dim3(64), 0, s, dK, dT, nb, lasso, dF);
    KP_HIP(ctx, hipGetLastError());
    std::vector<int> flags((size_t)n_deg * nb);
    KP_HIP(ctx, hipMemcpyAsync(flags.data(), dF, flags.size() * 4, hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    for (size_t e = 0; e < flags.size(); ++e)
      if (flags[e]) {                                 // L1 row active (never for the shipped / generated systems): lasso solve on (G_m, C_m)
        const int W = degs[e / nb].W;
        int rc = kp_lasso_dev(ctx, dG + e * 256, dC + e * 256, W, W, lasso * degs[e / nb].N, 20000, 1e-10, dK + e * 256, nullptr);
        if (rc && rc != KP_ERR_NOT_CONVERGED) return rc;
      }
  }
  if (b.model_type == KP_MODEL_LINEAR) {
    hipLaunchKernelGGL(kp_sweep_project_kernel, grid, dim3(256), 0, s, dK, dG, dC, nb, m, dT, dA, dB);
    KP_HIP(ctx, hipGetLastError());
  }
  static const bool rollout4 = getenv("KP_SWEEP_NO_ROLLOUT4") == nullptr;
  if (rollout4 && n <= 4) {
#define KP_ROLL4(MT_) hipLaunchKernelGGL(kp_sweep_rollout4_kernel<MT_>, dim3((nb * n_deg + 3) / 4), dim3(64), 0, s, b, dT, dCols, nb, n_deg, dK, \
                                         MT_ == KP_MODEL_LINEAR ? dA : nullptr, MT_ == KP_MODEL_LINEAR ? dB : nullptr, traj->Yv, traj->Uv, traj->Tv, dS, dE)
    if (b.model_type == KP_MODEL_LINEAR) KP_ROLL4(KP_MODEL_LINEAR);
    else if (b.model_type == KP_MODEL_BILINEAR) KP_ROLL4(KP_MODEL_BILINEAR);
    else KP_ROLL4(KP_MODEL_NONLINEAR);
#undef KP_ROLL4
  } else
    hipLaunchKernelGGL(kp_sweep_rollout_nested_kernel, grid, dim3(64), 0, s, b, dT, dCols, nb, dK, b.model_type == KP_MODEL_LINEAR ? dA : nullptr,
                       b.model_type == KP_MODEL_LINEAR ? dB : nullptr, traj->Yv, traj->Uv, traj->Tv, dS, dE);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipMemcpyAsync(err_out, dE, (size_t)n_deg * nb * n * 8, hipMemcpyDeviceToHost, s));
  if (status_out) KP_HIP(ctx, hipMemcpyAsync(status_out, dS, (size_t)n_deg * nb * 4, hipMemcpyDeviceToHost, s));
  KP_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[0] = ms;
  return KP_OK;
}

// K (degree j, monomial basis) of one nested evaluation, for parity checks: call right after kp_sweep_eval_nested
extern "C" int kp_sweep_nested_get_K(kp_ctx* ctx, int nb, int Wmax, int n_deg, int deg_index, int W, double* K_out) {
  if (!ctx || !K_out || deg_index < 0 || deg_index >= n_deg || !ctx->ws[6]) return ctx ? ctx->fail(KP_ERR_ARG, "kp_sweep_nested_get_K: bad argument") : KP_ERR_ARG;
  const size_t bW = (size_t)nb * Wmax * Wmax * 8;
  const double* dK = (const double*)((char*)ctx->ws[6] + 3 * bW) + (size_t)deg_index * nb * 256;
  std::vector<double> tmp((size_t)nb * 256);
  KP_HIP(ctx, hipMemcpy(tmp.data(), dK, tmp.size() * 8, hipMemcpyDeviceToHost));
  for (int q = 0; q < nb; ++q)
    for (int e = 0; e < W * W; ++e) K_out[(size_t)q * W * W + e] = tmp[(size_t)q * 256 + e];
  return KP_OK;
}

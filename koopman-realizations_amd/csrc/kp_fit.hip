// Normal-equation solve  G K = C  (K = Px \ Py of Ksysid.m:1069 for full-rank Px) and the
// host entry points of the EDMD fit.
//
//   kp_pad_kernel      copies G,C into 16-padded buffers (identity / zero padding)
//   kp_chol_kernel     one workgroup: right-looking blocked Cholesky (panel 16), writes L in
//                      the lower and L' in the upper triangle plus the inverses of the
//                      diagonal blocks
//   kp_trsm_kernel     one wave per 16 right-hand sides: forward + backward substitution by
//                      blocks with v_mfma_f64_16x16x4_f64, X block resident in LDS
#include <cmath>
#include <cstdlib>

#include <vector>
#include <algorithm>
#include <cstring>

#include "kp_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
extern "C" int kp_synchronize(kp_ctx* ctx);

// blockIdx.y = system of a batch: inputs gc_stride doubles apart (G and C of a fit are one [G | C] block), padded
// outputs n*n / n*ncp apart
// grid (row chunks of 256, n + ncp columns - G's then C's -, systems): no index arithmetic beyond a multiply (until round 6 one
// thread per element of a flat index: two 64-bit divisions each - 17.8 us for the 1.8 MB of one W = 336 system)
__global__ __launch_bounds__(256) void kp_pad_kernel(const double* __restrict__ G, const double* __restrict__ C, int W, int ncols, int n, int ncp,
                                                     double* __restrict__ Gp, double* __restrict__ Cp, size_t gc_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x, jj = blockIdx.y;
  if (i >= n) return;
  G += blockIdx.z * gc_stride; C += blockIdx.z * gc_stride; Gp += blockIdx.z * (size_t)n * n; Cp += blockIdx.z * (size_t)n * ncp;
  if (jj < n) {
    Gp[(size_t)jj * n + i] = (i < W && jj < W) ? G[(size_t)jj * W + i] : (i == jj ? 1.0 : 0.0);
  } else {
    const int j = jj - n;
    Cp[(size_t)j * n + i] = (i < W && j < ncols) ? C[(size_t)j * W + i] : 0.0;
  }
}

// batch: system y writes K + ((k_first + y) % k_cap) * W * ncols (result ring); k_cap = 0: K itself
// grid (row chunks of 256, ncols columns, systems)
__global__ __launch_bounds__(256) void kp_unpad_kernel(const double* __restrict__ Xp, int n, int W, int ncols, double* __restrict__ K, int ncp, int k_first, int k_cap) {
  const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (i >= W) return;
  Xp += blockIdx.z * (size_t)n * ncp;
  if (k_cap > 0) K += (size_t)((k_first + (int)blockIdx.z) % k_cap) * W * ncols;
  K[(size_t)j * W + i] = Xp[(size_t)j * n + i];
}

#define PS 17  // LDS row stride of the 16-wide panels

__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

#ifndef KP_CHOL_ABL
#define KP_CHOL_ABL 0
#endif
#define CH_NT 512   // 8 waves: two per SIMD, 256 VGPRs each (the 8 x 8 micro tiles of the trailing update hold 64 accumulators)

// 16 x 16 diagonal block at (k0, k0) by ONE wave, entirely in registers: lane r (< 16) owns row r; pivots and
// pivot-column entries travel by v_readlane, so the 16 serial pivot steps need no LDS round trip and no
// workgroup barrier.  Writes L (lower) / L' (upper) into A, the block D (LDS, lower part) and the reciprocals
// Dd of its diagonal.  The inverses of the diagonal blocks, which only the TRSM kernel needs, are formed after the
// factorisation by kp_chol_finish_kernel, off the critical path.
__device__ __forceinline__ void chol_diag_block(double* __restrict__ A, int n, int k0, double (*D)[16], double* Dd, int* bad,
                                                const double* __restrict__ odiag) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15;
  double row[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) row[c] = A[(size_t)(k0 + c) * n + k0 + r];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    double d = lane_bcast(row[c], c);
    // a pivot that is rounding noise of its original diagonal entry (n * 8 eps of it) is as singular as a non-positive
    // one: the caller falls back to the rank-revealing solve
    if (!(d > odiag[k0 + c])) {
      if (lane == 0) *bad = 1;
      d = 1.0;
    }
    // 1/sqrt(d): hardware estimate + two Newton steps (full f64 accuracy, no division)
    double id = __builtin_amdgcn_rsq(d);
    id = id * (1.5 - 0.5 * d * id * id);
    id = id * (1.5 - 0.5 * d * id * id);
    if (lane == c) Dd[c] = id;
    const double l = row[c] * id;          // lane c: sqrt(d); lanes r > c: L_rc
    row[c] = l;
#pragma unroll
    for (int cc = c + 1; cc < 16; ++cc) row[cc] -= l * lane_bcast(l, cc);   // entries with cc <= r are used
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double lrc = c <= r ? row[c] : 0.0;
      D[r][c] = lrc;
      if (c <= r) {
        A[(size_t)(k0 + c) * n + k0 + r] = lrc;             // L (lower)
        A[(size_t)(k0 + r) * n + k0 + c] = lrc;             // L' (upper)
      }
    }
  }
}

// 8 x 8 micro tile (ti >= tj) of the trailing update A22 -= L21 L21': L21 is in LDS as Lt[q][row] (the 8 values
// of a tile row are contiguous: 16-byte LDS reads, no bank conflicts between lanes with neighbouring tiles)
__device__ __forceinline__ void chol_tile_update(double* __restrict__ A22, int n, const double* __restrict__ Lt, int ld, int ti, int tj,
                                                 double* __restrict__ Pnext) {
  // the tile is loaded first (32 independent 16-byte loads in flight), updated in registers and stored once
  double acc[8][8];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const double2* src = reinterpret_cast<const double2*>(A22 + (size_t)(8 * tj + y) * n + 8 * ti);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const double2 v = src[h];
      acc[2 * h][y] = v.x;
      acc[2 * h + 1][y] = v.y;
    }
  }
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    double rv[8], cv[8];
    const double2* pr = reinterpret_cast<const double2*>(Lt + (size_t)q * ld + 8 * ti);
    const double2* pc = reinterpret_cast<const double2*>(Lt + (size_t)q * ld + 8 * tj);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const double2 a = pr[h], c = pc[h];
      rv[2 * h] = a.x; rv[2 * h + 1] = a.y;
      cv[2 * h] = c.x; cv[2 * h + 1] = c.y;
    }
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) acc[x][y] -= rv[x] * cv[y];
  }
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    double2* dst = reinterpret_cast<double2*>(A22 + (size_t)(8 * tj + y) * n + 8 * ti);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      double2 v;
      v.x = acc[2 * h][y];
      v.y = acc[2 * h + 1][y];
      dst[h] = v;
      // the first 16 columns of A22 are the next panel's A21: keep them in LDS ([q][row - 16]) for the next step
      if (tj <= 1) *reinterpret_cast<double2*>(Pnext + (size_t)(8 * tj + y) * ld + 8 * ti - 16 + 2 * h) = v;
    }
  }
}

// After the factorisation: L' into the upper triangle (the TRSM kernel reads both orientations coalesced), one 16 x 16
// block pair per workgroup, and the inverses of the diagonal blocks (one wave each), Dinv[kb][col][row].
__global__ __launch_bounds__(256) void kp_chol_finish_kernel(double* __restrict__ A, int n, int npair, double* __restrict__ Dinv) {
  __shared__ double T[16][17];
  A += blockIdx.y * (size_t)n * n; Dinv += blockIdx.y * (size_t)(n / 16) * 256;
  const int p = blockIdx.x, t = threadIdx.x;
  if (p >= npair) {                                  // inverse of diagonal block kb: lane j builds column j
    if (t >= 64) return;
    const int kb = p - npair, k0 = 16 * kb, r = t & 15;
    double row[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = A[(size_t)(k0 + c) * n + k0 + r];     // L_rc for c <= r
#pragma unroll
    for (int i = 0; i < 16; ++i) {                   // x_i = -(sum_{q=j}^{i-1} L_iq x_q) / L_ii, x_j = 1 / L_jj
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < i; ++q) {
        const double liq = lane_bcast(row[q], i);    // L_iq lives in lane i, register q
        sacc += (q >= r) ? liq * x[q] : 0.0;
      }
      const double idg = 1.0 / lane_bcast(row[i], i);
      x[i] = i == r ? idg : (i > r ? -sacc * idg : 0.0);
    }
    if (t < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) Dinv[(size_t)kb * 256 + r * 16 + c] = x[c];   // inverse(c, r)
    }
    return;
  }
  int bi = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);      // pair index -> block (bi > bj)
  while (bi * (bi - 1) / 2 > p) --bi;
  while ((bi + 1) * bi / 2 <= p) ++bi;
  const int bj = p - bi * (bi - 1) / 2;
  T[t >> 4][t & 15] = A[(size_t)(16 * bj + (t >> 4)) * n + 16 * bi + (t & 15)];     // L block: column t>>4, row t&15
  __syncthreads();
  A[(size_t)(16 * bi + (t >> 4)) * n + 16 * bj + (t & 15)] = T[t & 15][t >> 4];
}

__global__ __launch_bounds__(CH_NT) void kp_chol_kernel(double* __restrict__ A, int n, double* __restrict__ Dinv, int* __restrict__ info,
                                                        int* __restrict__ sticky, int prof) {
  long long tph[5] = {0, 0, 0, 0, 0}, tlast = 0;   // KP_CHOL_PROF=1: cycles per phase, printed by thread 0
#define CH_TICK(i) do { if (prof) { long long tnow = clock64(); tph[i] += tnow - tlast; tlast = tnow; } } while (0)
  extern __shared__ __align__(16) double sm[];
  __shared__ __align__(16) double D[16][16];
  __shared__ double Dd[16];
  __shared__ int bad;
  double* Pin = sm;                     // A21 as [q][row], leading dimension n
  double* Lt = sm + (size_t)16 * n;     // L21 as [q][row]
  const int tid = threadIdx.x, wave = tid >> 6;
  A += blockIdx.y * (size_t)n * n; Dinv += blockIdx.y * (size_t)(n / 16) * 256; info += blockIdx.y;   // system of a batch
  __shared__ double odiag[512];         // pivot thresholds: n * 8 eps * original diagonal (0 for the identity padding)
  if (tid == 0) bad = 0;
  for (int i = tid; i < n; i += CH_NT) odiag[i] = fmax(A[(size_t)i * n + i], 0.0) * ((double)n * 8.0 * 2.220446049250313e-16);
  __syncthreads();
  const int nt = n / 16;
  if (wave == 0) chol_diag_block(A, n, 0, D, Dd, &bad, odiag);
  __syncthreads();
  if (prof) tlast = clock64();
  for (int kb = 0; kb < nt; ++kb) {
    const int k0 = kb * 16;
    const int R = n - k0 - 16;
    if (R <= 0) break;
    // A21 -> LDS as [q][row] (consecutive threads = consecutive rows: coalesced, conflict free).  Only for the first
    // panel: afterwards the trailing update of the previous step has left the panel in LDS already.
    if (kb == 0) {
      for (int e = tid; e < R * 16; e += CH_NT) {
        const int r = e % R, q = e / R;
        Pin[q * n + r] = A[(size_t)(k0 + q) * n + k0 + 16 + r];
      }
      __syncthreads();
    }
    CH_TICK(0);
    // L21 = A21 * L^-T by forward substitution, one thread per row (the block L and 1/diag are broadcast LDS reads)
#if KP_CHOL_ABL != 3
    for (int r = tid; r < R; r += CH_NT) {
      double l[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double sacc = Pin[c * n + r];
#pragma unroll
        for (int q = 0; q < c; ++q) sacc -= l[q] * D[c][q];
        l[c] = sacc * Dd[c];
        Lt[c * n + r] = l[c];
        A[(size_t)(k0 + c) * n + k0 + 16 + r] = l[c];   // L (lower); L' (upper) is filled in once at the end
      }
    }
#endif
    __syncthreads();
    CH_TICK(1);
    // trailing update of the lower triangle in 8 x 8 micro tiles.  Look-ahead: the next diagonal block is updated first; then wave 0 factors that block while the other waves update the rest.
    const int ntb = R / 8;                          // even (R is a multiple of 16)
    double* A22 = A + (size_t)(k0 + 16) * n + k0 + 16;
    if (tid < 256) {                                // next diagonal block = micro tiles (0,0), (1,0), (1,1)
      const int r = tid & 15, c = tid >> 4;
      if (r >= c) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += Lt[q * n + r] * Lt[q * n + c];
        A22[(size_t)c * n + r] -= s;
      }
    }
    __syncthreads();
    CH_TICK(2);
    if (wave == 0) {
#if KP_CHOL_ABL != 2
      chol_diag_block(A, n, k0 + 16, D, Dd, &bad, odiag);
#endif
    } else {
      // the triangle of ntb x ntb micro tiles is folded into an (ntb+1) x (ntb/2) rectangle; tiles (0,0),(1,0),(1,1) are done
      const int total = (ntb + 1) * (ntb / 2);
      for (int e = tid - 64; e < total; e += CH_NT - 64) {
        const int i = e % (ntb + 1), j = e / (ntb + 1);
        int ti, tj;
        if (i > j) {
          ti = i - 1;
          tj = j;
        } else {
          ti = ntb - 1 - i;
          tj = ntb - 1 - j;
        }
        if (ti <= 1) continue;                      // (0,0), (1,0), (1,1)
#if KP_CHOL_ABL != 1
        chol_tile_update(A22, n, Lt, n, ti, tj, Pin);
#endif
      }
    }
    if (prof && wave == 0) CH_TICK(3);       // wave 0: the diagonal block alone
    __syncthreads();
    if (prof && wave == 0) CH_TICK(4); else CH_TICK(3);
  }
  if (prof && (tid == 0 || tid == 64))
    printf("chol prof tid %d: A21 load %lld  L21 %lld  next-diag update %lld  [wave0: diag | others: tiles] %lld  wait %lld\n", tid, tph[0], tph[1], tph[2],
           tph[3], tph[4]);
  if (tid == 0) {
    *info = bad;
    if (bad && sticky) *sticky = 1;
  }
}

// Forward + backward block substitution for 16 right-hand sides by a 4-wave workgroup.
// Block row i:  X_i = Dinv_i (C_i - sum_{j<i} L_ij X_j).  The sum over j is dealt round-robin to
// the 4 waves (j = wave, wave+4, ...), partial tiles are combined through LDS by wave 0, which
// applies the inverse diagonal block.  The L operands of step i+1 do not depend on X, so they are
// loaded from global memory (L2) during step i (software pipeline across steps).
#define TR_MAXJ 8   // block products per wave and step: ceil(nt/4) <= 8  =>  n <= 512
__global__ __launch_bounds__(256) void kp_trsm_kernel(const double* __restrict__ LU, const double* __restrict__ Dinv, int n,
                                                     double* __restrict__ X) {
  extern __shared__ double xs[];  // [n][16] X block (row = 16 consecutive doubles) | part[4][256] partial tiles
  double* part = xs + (size_t)n * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x;
  const int nt = n / 16;
  const int lr = lane >> 4, lc = lane & 15;
  LU += blockIdx.y * (size_t)n * n; Dinv += blockIdx.y * (size_t)nt * 256; X += blockIdx.y * (size_t)n * gridDim.x * 16;   // system of a batch
  double* Xb = X + (size_t)cb * 16 * n;
  for (int e = tid; e < n * 16; e += 256) {
    int row = e % n, col = e / n;
    xs[row * 16 + col] = Xb[(size_t)col * n + row];       // C block -> LDS (overwritten by Y, then K)
  }
  __syncthreads();
  double an[TR_MAXJ][4];                                   // -L operands of the NEXT step
  for (int dir = 0; dir < 2; ++dir) {                      // 0: L Y = C (forward), 1: L' K = Y (backward)
    // operands of block row i against block column j: A[r][k] = LU[(16j + k) * n + 16i + r]
    // (lower triangle holds L, upper triangle L', so the same formula serves both sweeps)
    auto load_ops = [&](int i) {
#pragma unroll
      for (int s = 0; s < TR_MAXJ; ++s) {
        const int jj = wave + 4 * s;
        const int j = dir == 0 ? jj : nt - 1 - jj;
        const bool on = dir == 0 ? (j < i) : (j > i);
        if (i >= 0 && i < nt && on) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) an[s][kk] = -LU[(size_t)(j * 16 + kk * 4 + lr) * n + i * 16 + lc];
        }
      }
    };
    const int i0 = dir == 0 ? 0 : nt - 1, istep = dir == 0 ? 1 : -1;
    load_ops(i0);
    for (int it = 0; it < nt; ++it) {
      const int i = i0 + it * istep;
      double ac[TR_MAXJ][4];
#pragma unroll
      for (int s = 0; s < TR_MAXJ; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ac[s][kk] = an[s][kk];
      load_ops(i + istep);                                  // prefetch the next step's L tiles
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < TR_MAXJ; ++s) {
        const int jj = wave + 4 * s;
        const int j = dir == 0 ? jj : nt - 1 - jj;
        const bool on = dir == 0 ? (j < i) : (j > i);
        if (on) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double bv = xs[(j * 16 + kk * 4 + lr) * 16 + lc];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[s][kk], bv, acc, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave * 256 + (lr + 4 * r) * 16 + lc] = acc[r];
      __syncthreads();
      if (wave == 0) {
        double4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = kk * 4 + lr;
          // forward: Dinv_i[lc][k]; backward: (Dinv_i')[lc][k] = Dinv_i[k][lc]
          const double av = dir == 0 ? Dinv[(size_t)i * 256 + k * 16 + lc] : Dinv[(size_t)i * 256 + lc * 16 + k];
          const int o = k * 16 + lc;
          const double bv = xs[(i * 16 + k) * 16 + lc] + part[o] + part[256 + o] + part[512 + o] + part[768 + o];
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, y, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(i * 16 + lr + 4 * r) * 16 + lc] = y[r];
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < n * 16; e += 256) {
    int row = e % n, col = e / n;
    Xb[(size_t)col * n + row] = xs[row * 16 + col];
  }
}

static bool trsm_old_sel() {
  static const bool v = getenv("KP_TRSM_OLD") != nullptr;
  return v;
}

// ---- block substitution, second form: right-looking, one barrier per step, v_mfma_f64_4x4x4 --------------------------
// Forward  L Y = C  and backward  L' K = Y  for 16 right-hand sides by a 4-wave workgroup.  Row block i belongs to wave
// i % 4, which keeps its residual R_i = C_i - sum_j L_ij Y_j in accumulators.  Step j: the owner of row j applies the pending
// update with Y_{j-1} to that row first, multiplies by the inverse diagonal block and publishes Y_j (LDS); everyone else
// meanwhile applies Y_{j-1} to the rest of their rows; ONE barrier (LDS-only: the L tiles of column j, requested just before
// it, stay in flight).  The left-looking form above has two barriers, a 4-way partial-sum exchange and conditional loads (a
// drained queue) per step - and uses v_mfma_f64_16x16x4, which runs at half the rate of the 4x4x4 form on gfx950 (140 against
// 4 x 16.5 cycles per 2048 flop): a 16 x 16 x 16 tile product is 16 MFMAs here (blocks = the 4 row groups of the tile; the B
// operand - a 4 x 4 block of Y_j - is shared by the blocks).
#define TR2_MAXR 8   // row blocks per wave: ceil(32 / 4)  =>  n <= 512
namespace {

// LDS column order of a block of 4 NJ right-hand sides: a lane's NJ values (columns 4 J + jj) are contiguous
template <int NJ>
__device__ __forceinline__ int tr2_col(int c) { return (c & 3) * NJ + (c >> 2); }

// an[q][K] = -(A operand of tile (row(q), column block jc)): rows 4 blk + jj of the tile, column 4 K + kq
template <int CNT>
__device__ __forceinline__ void tr2_load(double (&an)[TR2_MAXR][4], const double* __restrict__ LU, int64_t n, int jc, const int (&rowof)[TR2_MAXR], int kq,
                                         int blk, int jj) {
#pragma unroll
  for (int q = 0; q < CNT; ++q)
#pragma unroll
    for (int K = 0; K < 4; ++K) an[q][K] = -LU[(size_t)(jc * 16 + 4 * K + kq) * n + rowof[q] * 16 + 4 * blk + jj];
}

// slots [q0, CNT): R -= L Y  (4 NJ MFMAs per tile: NJ independent chains of 4)
template <int CNT, int NJ>
__device__ __forceinline__ void tr2_update(double (&acc)[TR2_MAXR][NJ], const double (&an)[TR2_MAXR][4], const double (&bv)[4][NJ], int q0) {
#pragma unroll
  for (int q = 0; q < CNT; ++q) {
    if (q < q0) continue;
#pragma unroll
    for (int K = 0; K < 4; ++K)
#pragma unroll
      for (int J = 0; J < NJ; ++J) acc[q][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[q][K], bv[K][J], acc[q][J], 0, 0, 0);
  }
}

// the NJ values of a lane in row `row` of xs (row stride 4 NJ doubles)
template <int NJ>
__device__ __forceinline__ void tr2_get(const double* __restrict__ xs, int row, int jj, double (&v)[NJ]) {
  const double* p = xs + row * (4 * NJ) + jj * NJ;
  if constexpr (NJ == 4) {
    const double2 v01 = reinterpret_cast<const double2*>(p)[0], v23 = reinterpret_cast<const double2*>(p)[1];
    v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
  } else if constexpr (NJ == 2) {
    const double2 v01 = reinterpret_cast<const double2*>(p)[0];
    v[0] = v01.x; v[1] = v01.y;
  } else {
    v[0] = p[0];
  }
}

template <int NJ>
__device__ __forceinline__ void tr2_put(double* __restrict__ xs, int row, int jj, const double (&v)[NJ]) {
  double* p = xs + row * (4 * NJ) + jj * NJ;
  if constexpr (NJ == 4) {
    double2 v01, v23;
    v01.x = v[0]; v01.y = v[1]; v23.x = v[2]; v23.y = v[3];
    reinterpret_cast<double2*>(p)[0] = v01; reinterpret_cast<double2*>(p)[1] = v23;
  } else if constexpr (NJ == 2) {
    double2 v01;
    v01.x = v[0]; v01.y = v[1];
    reinterpret_cast<double2*>(p)[0] = v01;
  } else {
    p[0] = v[0];
  }
}

// B operands of a 16-row block of xs: b[K][J] = X(4 K + kq, 4 J + jj)
template <int NJ>
__device__ __forceinline__ void tr2_read_b(const double* __restrict__ xs, int j, int kq, int jj, double (&b)[4][NJ]) {
#pragma unroll
  for (int K = 0; K < 4; ++K) tr2_get<NJ>(xs, j * 16 + 4 * K + kq, jj, b[K]);
}

}  // namespace

#define TR2_SWITCH(CALL)                                                                   \
  switch (cnt) {                                                                           \
    case 1: { constexpr int C_ = 1; CALL; } break;                                         \
    case 2: { constexpr int C_ = 2; CALL; } break;                                         \
    case 3: { constexpr int C_ = 3; CALL; } break;                                         \
    case 4: { constexpr int C_ = 4; CALL; } break;                                         \
    case 5: { constexpr int C_ = 5; CALL; } break;                                         \
    case 6: { constexpr int C_ = 6; CALL; } break;                                         \
    case 7: { constexpr int C_ = 7; CALL; } break;                                         \
    case 8: { constexpr int C_ = 8; CALL; } break;                                         \
    default: break;                                                                        \
  }

// NJ = right-hand sides per workgroup / 4.  One system: NJ = 1 (W / 4 workgroups, the shortest serial chain per step); a batch
// of systems: NJ = 4 (every workgroup reads all of L: fewer, wider workgroups keep that traffic down).
// Kout (or nullptr): the solution goes there unpadded (W x ncols, leading dimension W) instead of back into X - the single-
// system path, which then needs no unpad launch behind it.
// ldl / ldx: leading dimensions of LU and X (the blocked factorisation of wide systems, kp_wide.hip, substitutes against a
// diagonal block that sits inside a larger matrix and right-hand sides that are rows of one); dirs: 1 = forward only,
// 2 = backward only, 3 = both.
template <int NJ>
__global__ __launch_bounds__(256) void kp_trsm2_kernel(const double* __restrict__ LU, const double* __restrict__ Dinv, int n, int ncp,
                                                      double* __restrict__ X, double* __restrict__ Kout, int W, int ncols, int64_t ldl, int64_t ldx,
                                                      int dirs, const int* __restrict__ info) {
  // (info: the factorisation's status word(s), or nullptr - a factor that hit a non-positive pivot is not substituted with:
  // the caller reads the same word and takes the rank-revealing path)
  if (info && info[blockIdx.y]) return;
  extern __shared__ __align__(16) double xs[];  // [n][4 NJ] X block, columns in tr2_col order
  constexpr int NC = 4 * NJ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.x;
  const int nt = n / 16;
  const int kq = lane >> 4, blk = (lane >> 2) & 3, jj = lane & 3;
  LU += blockIdx.y * (size_t)n * n; Dinv += blockIdx.y * (size_t)nt * 256; X += blockIdx.y * (size_t)ldx * ncp;   // system of a batch
  double* Xb = X + (size_t)cb * NC * ldx;
  for (int e = tid; e < n * NC; e += 256) {
    int row = e % n, col = e / n;
    xs[row * NC + tr2_col<NJ>(col)] = Xb[(size_t)col * ldx + row];       // C block -> LDS (overwritten by Y, then K)
  }
  __syncthreads();
  const int nown = wave < nt ? (nt - wave + 3) / 4 : 0;    // row blocks wave, wave + 4, ... < nt
  for (int dir = 0; dir < 2; ++dir) {                      // 0: L Y = C (forward), 1: L' K = Y (backward)
    if (!((dirs >> dir) & 1)) continue;
    // slot q <-> row block, the rows that retire LAST first, so that the rows a step still reaches are always the slots
    // [0, cnt) and the row that retires next is slot cnt - 1: forward rows retire ascending (slot 0 = the largest row),
    // backward descending (slot 0 = the smallest)
    int rowof[TR2_MAXR];
#pragma unroll
    for (int q = 0; q < TR2_MAXR; ++q) rowof[q] = q < nown ? (dir == 0 ? wave + 4 * (nown - 1 - q) : wave + 4 * q) : 0;
    double acc[TR2_MAXR][NJ];                              // R(4 blk + kq, 4 J + jj) of the slot's row block
#pragma unroll
    for (int q = 0; q < TR2_MAXR; ++q) tr2_get<NJ>(xs, rowof[q] * 16 + 4 * blk + kq, jj, acc[q]);
    // owned rows strictly beyond row j in sweep direction (= the rows the update with X_j reaches)
    auto beyond = [&](int j) {
      const int c = dir == 0 ? nown - (j < wave ? 0 : (j - wave) / 4 + 1) : (j <= wave ? 0 : (j - wave + 3) / 4);
      return min(max(c, 0), nown);
    };
    const int j0 = dir == 0 ? 0 : nt - 1, jstep = dir == 0 ? 1 : -1;
    // inverse diagonal block (A operand: row 4 blk + jj, column 4 K + kq; transposed in the backward sweep) of the next
    // row this wave retires, requested one ownership period ahead
    auto load_dinv = [&](int j, double (&dv)[4]) {
      const int jc = min(max(j, 0), nt - 1);
#pragma unroll
      for (int K = 0; K < 4; ++K) {
        const int r = 4 * blk + jj, c = 4 * K + kq;
        dv[K] = dir == 0 ? Dinv[(size_t)jc * 256 + c * 16 + r] : Dinv[(size_t)jc * 256 + r * 16 + c];
      }
    };
    double dv[4];
    {
      int jf = j0;                                          // first row this wave retires
      while (jf >= 0 && jf < nt && (jf & 3) != wave) jf += jstep;
      load_dinv(jf, dv);
    }
    double an[TR2_MAXR][4];
    double bv[4][NJ];
    int cnt = nown;                                        // rows reached by "X_{j0 - jstep}": all of them (nothing pending)
    for (int it = 0; it < nt; ++it) {
      const int j = j0 + it * jstep;
      const bool owner = (j & 3) == wave;
      // here: an / bv = tiles of column j - jstep for the slots [0, cnt) and X_{j - jstep} (it > 0), not yet applied
      if (owner) {
        const int qj = cnt - 1;                            // slot of row j
        if (it > 0) { TR2_SWITCH((tr2_update<C_, NJ>(acc, an, bv, C_ - 1))); }      // row j alone
        double rj[NJ];
#pragma unroll
        for (int J = 0; J < NJ; ++J) rj[J] = acc[0][J];
#pragma unroll
        for (int q = 1; q < TR2_MAXR; ++q)
          if (q == qj) {
#pragma unroll
            for (int J = 0; J < NJ; ++J) rj[J] = acc[q][J];
          }
        // R_j through LDS into the B-operand layout (only this wave touches block j of xs now), times the inverse block
        tr2_put<NJ>(xs, j * 16 + 4 * blk + kq, jj, rj);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double br[4][NJ];
        tr2_read_b<NJ>(xs, j, kq, jj, br);
        double y[NJ];
#pragma unroll
        for (int J = 0; J < NJ; ++J) y[J] = 0.0;
#pragma unroll
        for (int K = 0; K < 4; ++K)
#pragma unroll
          for (int J = 0; J < NJ; ++J) y[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(dv[K], br[K][J], y[J], 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                    // all lanes have read R_j before Y_j replaces it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        tr2_put<NJ>(xs, j * 16 + 4 * blk + kq, jj, y);
        load_dinv(j + 4 * jstep, dv);
        if (it > 0) { TR2_SWITCH((tr2_update<C_ - 1, NJ>(acc, an, bv, 0))); }      // the rest of this wave's rows
      } else if (it > 0) {
        TR2_SWITCH((tr2_update<C_, NJ>(acc, an, bv, 0)));
      }
      // tiles of column j for the rows beyond it: requested now, used after the barrier
      cnt = beyond(j);
      TR2_SWITCH((tr2_load<C_>(an, LU, ldl, j, rowof, kq, blk, jj)));
      // LDS-only barrier: the tile loads just issued stay in flight across it (a __syncthreads would wait for them)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      tr2_read_b<NJ>(xs, j, kq, jj, bv);
    }
    __syncthreads();
  }
  for (int e = tid; e < n * NC; e += 256) {
    int row = e % n, col = e / n;
    const double v = xs[row * NC + tr2_col<NJ>(col)];
    if (Kout) {
      const int gc = cb * NC + col;
      if (row < W && gc < ncols) Kout[(size_t)gc * W + row] = v;
    } else {
      Xb[(size_t)col * ldx + row] = v;
    }
  }
}

// ---- systems beyond one workgroup's reach (n > 512): blocked right-looking factorisation over MANY workgroups ------------------
// The reference solves `K = Px \ Py` (Ksysid.m:1069) for whatever dictionary the user configured; its fourier dictionary on the
// arm's six states has 738 (linear) / 2 940 (bilinear) columns (Ksysid.m:694-731).  Structure (block = KP_WIDE_BS columns, 256):
//   for each diagonal block k:  D = A_kk (gathered from the upper triangle)  ->  kp_chol_ll_kernel (the one-workgroup
//   factorisation of the narrow path, with the pivot thresholds of the ORIGINAL diagonal)  ->  kp_chol_finish_kernel (L' and the
//   inverses of the 16 x 16 diagonal blocks)  ->  back into A;   U_12 = L_kk^-1 A_12: kp_trsm2_kernel<4>, forward sweep only, the
//   right-hand sides are the COLUMNS of the upper block row (contiguous along the contraction index of what follows);
//   A_22 -= U_12' U_12: kp_tn_gemm (upper tiles only, every CU).
// Only the upper triangle is ever updated; it is mirrored once at the end (the backward sweep reads L = U').  The solve is the
// same three kernels: per block a forward (then backward) substitution against the diagonal block and one TN product that
// takes the block's solution out of all remaining rows.  The one-workgroup chain (~0.5 us per column) is then only the
// diagonal blocks; everything O(n^3) runs on all CUs.
__global__ __launch_bounds__(256) void kp_wide_thresh_kernel(const double* __restrict__ A, int n, double* __restrict__ thr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) thr[i] = fmax(A[(size_t)i * n + i], 0.0) * ((double)n * 8.0 * 2.220446049250313e-16);
}

// D (b x b) = symmetric block of A at (k0, k0), read from the UPPER triangle;  back: both triangles of the factored block
__global__ __launch_bounds__(256) void kp_wide_diag_kernel(double* __restrict__ A, int n, int k0, int b, double* __restrict__ D, int back) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= b * b) return;
  const int i = e % b, j = e / b;
  if (back) A[(size_t)(k0 + j) * n + k0 + i] = D[e];
  else D[e] = A[(size_t)(k0 + max(i, j)) * n + k0 + min(i, j)];
}

static int wide_block() {               // (read per call: the tests vary it)
  const char* e = getenv("KP_WIDE_BS");
  const int b = e ? atoi(e) : 256;
  return std::max(16, std::min(352, b)) / 16 * 16;
}

#include "kp_tn_gemm.h"

// one direction of the block substitution against a diagonal block (b x b at Akk, leading dimension n) for `nrhs` right-hand sides
// at X (leading dimension n): 4 right-hand sides per workgroup (the shortest serial chain) while that gives at most ~2 rounds of
// workgroups, 16 beyond (every workgroup reads the whole block)
static hipError_t wide_trsm(hipStream_t st, const double* Akk, const double* Dinv_k, int b, int nrhs, double* X, int n, int dirs) {
  if (nrhs <= 4096)
    hipLaunchKernelGGL((kp_trsm2_kernel<1>), dim3(nrhs / 4, 1), dim3(256), (size_t)b * 4 * 8, st, Akk, Dinv_k, b, nrhs, X, (double*)nullptr, 0, 0, (int64_t)n, (int64_t)n,
                       dirs, (const int*)nullptr);
  else
    hipLaunchKernelGGL((kp_trsm2_kernel<4>), dim3(nrhs / 16, 1), dim3(256), (size_t)b * 16 * 8, st, Akk, Dinv_k, b, nrhs, X, (double*)nullptr, 0, 0, (int64_t)n, (int64_t)n,
                       dirs, (const int*)nullptr);
  return hipGetLastError();
}

// Cp := (L L')^-1 Cp for a factor as chol_solve_wide leaves it (L below, L' above the diagonal, Dinv = inverses of the 16 x 16
// diagonal blocks): per block a forward (then backward) substitution against the diagonal block and one TN product that takes
// the block's solution out of all remaining rows.
static int wide_substitute(kp_ctx* ctx, double* Gp, double* Cp, double* Dinv, int n, int ncp, hipStream_t st, bool forward_done = false) {
  const int bs = wide_block();
  static KpLdsCache trsm_lds;
  KP_HIP(ctx, kp_ensure_lds(trsm_lds, (const void*)kp_trsm2_kernel<4>, (size_t)352 * 16 * 8));
  // forward: L Y = C, block rows ascending; the block's Y leaves all later rows by one product
  for (int k0 = 0; k0 < n && !forward_done; k0 += bs) {
    const int b = std::min(bs, n - k0), R = n - k0 - b;
    KP_HIP(ctx, wide_trsm(st, Gp + (size_t)k0 * n + k0, Dinv + (size_t)(k0 / 16) * 256, b, ncp, Cp + k0, n, 1));
    if (R > 0) KP_HIP(ctx, kp_tn_gemm(st, Gp + (size_t)(k0 + b) * n + k0, n, Cp + k0, n, R, ncp, b, Cp + k0 + b, n, -1.0, 1.0, 0, 1, nullptr));
  }
  // backward: L' K = Y, block rows descending; K_k leaves the rows above through L (the mirrored lower triangle)
  for (int k0 = (n - 1) / bs * bs; k0 >= 0; k0 -= bs) {
    const int b = std::min(bs, n - k0);
    KP_HIP(ctx, wide_trsm(st, Gp + (size_t)k0 * n + k0, Dinv + (size_t)(k0 / 16) * 256, b, ncp, Cp + k0, n, 2));
    if (k0 > 0) KP_HIP(ctx, kp_tn_gemm(st, Gp + k0, n, Cp + k0, n, k0, ncp, b, Cp, n, -1.0, 1.0, 0, 1, nullptr));
  }
  return KP_OK;
}

// Gp (n x n, padded, identity beyond W), Cp (n x ncp): factor Gp in place (L below, L' above the diagonal, as the narrow path
// leaves it), Dinv = inverses of the 16 x 16 diagonal blocks, Cp := Gp^-1 Cp.  *info is set when a pivot falls below its threshold.
static int chol_solve_wide(kp_ctx* ctx, double* Gp, double* Cp, double* Dinv, int n, int ncp, int* info, int* sticky, hipStream_t st) {
  const int bs = wide_block();
  double* scr = (double*)ctx->workspace(18, ((size_t)bs * bs + n) * 8 + 64);
  if (!scr) return ctx->fail(KP_ERR_HIP, "kp_fit_solve: out of device memory");
  double* D = scr;
  double* thr = scr + (size_t)bs * bs;
  int* dummy = (int*)(thr + n);
  static KpLdsCache trsm_lds;
  KP_HIP(ctx, kp_ensure_lds(trsm_lds, (const void*)kp_trsm2_kernel<4>, (size_t)352 * 16 * 8));
  // The right-hand sides directly behind the matrix (same leading dimension: [Gp | Cp] is ONE n x (n + ncp) array, as
  // kp_chol_solve_batch_dev lays them out): the forward substitution rides on the factorisation - C is a few more columns of
  // the block row A12 and of the trailing matrix - instead of repeating its launches afterwards (a W = 2 940 solve: 12 fewer
  // substitution launches of ~38 us each, and the late, small trailing updates fill the chip with the right-hand sides' tiles).
  // Same operations per element in the same order.  KP_WIDE_SEPARATE_FORWARD=1 (read per call) keeps the two phases apart.
  const bool fwd = Cp == Gp + (size_t)n * n && !getenv("KP_WIDE_SEPARATE_FORWARD");
  KP_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int), st));
  hipLaunchKernelGGL(kp_wide_thresh_kernel, dim3((n + 255) / 256), dim3(256), 0, st, Gp, n, thr);
  for (int k0 = 0; k0 < n; k0 += bs) {
    const int b = std::min(bs, n - k0), R = n - k0 - b;
    double* Akk = Gp + (size_t)k0 * n + k0;
    double* A12 = Gp + (size_t)(k0 + b) * n + k0;
    hipLaunchKernelGGL(kp_wide_diag_kernel, dim3((b * b + 255) / 256), dim3(256), 0, st, Gp, n, k0, b, D, 0);
    KP_HIP(ctx, kp_chol_ll_launch(D, b, 1, dummy, info, 0, st, thr + k0));
    {
      const int nbk = b / 16, npair = nbk * (nbk - 1) / 2;
      hipLaunchKernelGGL(kp_chol_finish_kernel, dim3(npair + nbk, 1), dim3(256), 0, st, D, b, npair, Dinv + (size_t)(k0 / 16) * 256);
    }
    hipLaunchKernelGGL(kp_wide_diag_kernel, dim3((b * b + 255) / 256), dim3(256), 0, st, Gp, n, k0, b, D, 1);
    KP_HIP(ctx, hipGetLastError());
    if (fwd) {
      // [A12 | C_k]: the columns right of the block and the right-hand sides in ONE substitution, [A22 | C_rest] in ONE product
      // (the right-hand sides lie right of the diagonal of every tile row: never skipped by the upper-tiles rule)
      KP_HIP(ctx, wide_trsm(st, Akk, Dinv + (size_t)(k0 / 16) * 256, b, R + ncp, A12, n, 1));
      if (R > 0) KP_HIP(ctx, kp_tn_gemm(st, A12, n, A12, n, R, R + ncp, b, Gp + (size_t)(k0 + b) * n + k0 + b, n, -1.0, 1.0, 1, 1, nullptr));
    } else if (R > 0) {
      KP_HIP(ctx, wide_trsm(st, Akk, Dinv + (size_t)(k0 / 16) * 256, b, R, A12, n, 1));
      KP_HIP(ctx, kp_tn_gemm(st, A12, n, A12, n, R, R, b, Gp + (size_t)(k0 + b) * n + k0 + b, n, -1.0, 1.0, 1, 1, nullptr));
    }
  }
  hipLaunchKernelGGL(kp_mirror_upper_kernel, dim3(n / 16, n / 16), dim3(256), 0, st, Gp, n, (int64_t)n);
  KP_HIP(ctx, hipGetLastError());
  if (sticky) {   // (the deferred pipeline's word; wide fits run synchronously, so this is only for symmetry with the narrow path)
    KP_HIP(ctx, hipMemcpyAsync(sticky, info, sizeof(int), hipMemcpyDeviceToDevice, st));
  }
  return wide_substitute(ctx, Gp, Cp, Dinv, n, ncp, st, fwd);
}

// For a caller that already holds a Cholesky factor (the rank-revealing solve, kp_pivchol.hip): Lp = n x n (n a multiple of
// 16), L in the lower triangle; Cp (n x ncp, ncp a multiple of 16) := (L L')^-1 Cp.  Dinv: room for (n / 16) * 256 doubles.
int kp_factor_substitute_dev(kp_ctx* ctx, double* Lp, int n, double* Cp, int ncp, double* Dinv, hipStream_t st) {
  if (!st) st = ctx->stream;
  const int nbk = n / 16, npair = nbk * (nbk - 1) / 2;
  hipLaunchKernelGGL(kp_chol_finish_kernel, dim3(npair + nbk, 1), dim3(256), 0, st, Lp, n, npair, Dinv);
  KP_HIP(ctx, hipGetLastError());
  if (n > 16 * 4 * TR_MAXJ) return wide_substitute(ctx, Lp, Cp, Dinv, n, ncp, st);
  static KpLdsCache trsm2_lds;
  KP_HIP(ctx, kp_ensure_lds(trsm2_lds, (const void*)kp_trsm2_kernel<4>, (size_t)352 * 16 * 8 > (size_t)n * 16 * 8 ? (size_t)352 * 16 * 8 : (size_t)n * 16 * 8));
  if (ncp <= 512)
    hipLaunchKernelGGL((kp_trsm2_kernel<1>), dim3(ncp / 4, 1), dim3(256), (size_t)n * 4 * 8, st, Lp, Dinv, n, ncp, Cp, (double*)nullptr, 0, 0, (int64_t)n, (int64_t)n, 3, (const int*)nullptr);
  else
    hipLaunchKernelGGL((kp_trsm2_kernel<4>), dim3(ncp / 16, 1), dim3(256), (size_t)n * 16 * 8, st, Lp, Dinv, n, ncp, Cp, (double*)nullptr, 0, 0, (int64_t)n, (int64_t)n, 3, (const int*)nullptr);
  KP_HIP(ctx, hipGetLastError());
  return KP_OK;
}

// nb systems [G | C] gc_stride doubles apart (W x W each; ncols = W when nb > 1); system y's K goes to
// K_dev + ((k_first + y) % k_cap) W ncols (k_cap = 0: K_dev).  One launch sequence for the whole batch.
int kp_chol_solve_batch_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, int nb, size_t gc_stride, double* K_dev,
                            int k_first, int k_cap, hipStream_t st, hipEvent_t pad_done, int* sticky) {
  if (!st) st = ctx->stream;
  const int n = (W + 15) / 16 * 16, ncp = (ncols + 15) / 16 * 16;
  size_t bG = (size_t)n * n * 8, bC = (size_t)n * ncp * 8, bD = (size_t)(n / 16) * 256 * 8;
  char* ws = (char*)ctx->workspace(5, (size_t)nb * (bG + bC + bD) + (size_t)nb * 4 + 64);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_solve: out of device memory");
  double* Gp = (double*)ws;
  double* Cp = (double*)(ws + (size_t)nb * bG);
  double* Dinv = (double*)(ws + (size_t)nb * (bG + bC));
  int* info = (int*)(ws + (size_t)nb * (bG + bC + bD));   // nb = 1: kp_chol_info_offset(W, ncols), the readers' formula
  size_t lds_chol = (size_t)2 * 16 * n * 8;
  size_t lds_trsm = ((size_t)n * 16 + 1024) * 8;
  const bool wide = n > 16 * 4 * TR_MAXJ;
  if (wide && nb != 1) return ctx->fail(KP_ERR_ARG, "kp_fit_solve: systems wider than 512 are solved one at a time");
  if (!wide && (lds_chol > 160 * 1024 - 8192 || lds_trsm > 160 * 1024)) return ctx->fail(KP_ERR_ARG, "kp_fit_solve: W too large");
  hipLaunchKernelGGL(kp_pad_kernel, dim3((n + 255) / 256, n + ncp, nb), dim3(256), 0, st, G_dev, C_dev, W, ncols, n, ncp, Gp, Cp, gc_stride);
  KP_HIP(ctx, hipGetLastError());
  if (pad_done) KP_HIP(ctx, hipEventRecord(pad_done, st));
  if (wide) {      // blocked factorisation and substitution over all CUs
    int rc = chol_solve_wide(ctx, Gp, Cp, Dinv, n, ncp, info, sticky, st);
    if (rc) return rc;
    double* Kdst = K_dev + (k_cap > 0 ? (size_t)(k_first % k_cap) * W * ncols : 0);
    hipLaunchKernelGGL(kp_unpad_kernel, dim3((W + 255) / 256, ncols, 1), dim3(256), 0, st, Cp, n, W, ncols, Kdst, ncp, 0, 0);
    KP_HIP(ctx, hipGetLastError());
    return KP_OK;
  }
  static KpLdsCache chol_lds, trsm_lds;
  static const int chol_prof = getenv("KP_CHOL_PROF") ? atoi(getenv("KP_CHOL_PROF")) : 0;
  if (kp_chol_ll_applicable(n)) {                 // n <= 352: left-looking factorisation on the matrix pipe (kp_chol_ll.hip)
    KP_HIP(ctx, kp_chol_ll_launch(Gp, n, nb, info, sticky, chol_prof, st));
  } else {
    KP_HIP(ctx, kp_ensure_lds(chol_lds, (const void*)kp_chol_kernel, lds_chol));
    hipLaunchKernelGGL(kp_chol_kernel, dim3(1, nb), dim3(CH_NT), lds_chol, st, Gp, n, Dinv, info, sticky, chol_prof);
    KP_HIP(ctx, hipGetLastError());
  }
  {
    const int nbk = n / 16, npair = nbk * (nbk - 1) / 2;
    hipLaunchKernelGGL(kp_chol_finish_kernel, dim3(npair + nbk, nb), dim3(256), 0, st, Gp, n, npair, Dinv);
  }
  KP_HIP(ctx, hipGetLastError());
  if (trsm_old_sel()) {
    KP_HIP(ctx, kp_ensure_lds(trsm_lds, (const void*)kp_trsm_kernel, lds_trsm));
    hipLaunchKernelGGL(kp_trsm_kernel, dim3(ncp / 16, nb), dim3(256), lds_trsm, st, Gp, Dinv, n, Cp);
  } else if (nb == 1) {       // one system: 4 right-hand sides per workgroup - the shortest serial chain per step (LDS <= 16 KB)
    // K written in place (no unpad launch): the ring offset kp_unpad_kernel would apply is applied here - a lone queued fit
    // flushed at pend_first % k_cap != 0 must land in ITS slot of the result ring, not in slot 0
    double* Kdst = K_dev + (k_cap > 0 ? (size_t)(k_first % k_cap) * W * ncols : 0);
    hipLaunchKernelGGL((kp_trsm2_kernel<1>), dim3(ncp / 4, nb), dim3(256), (size_t)n * 4 * 8, st, Gp, Dinv, n, ncp, Cp, Kdst, W, ncols, (int64_t)n,
                       (int64_t)n, 3, (const int*)info);
    KP_HIP(ctx, hipGetLastError());
    return KP_OK;
  } else {                    // a batch: 16 per workgroup, every workgroup reads all of L
    static KpLdsCache trsm2_lds;
    KP_HIP(ctx, kp_ensure_lds(trsm2_lds, (const void*)kp_trsm2_kernel<4>, (size_t)n * 16 * 8));
    hipLaunchKernelGGL((kp_trsm2_kernel<4>), dim3(ncp / 16, nb), dim3(256), (size_t)n * 16 * 8, st, Gp, Dinv, n, ncp, Cp, (double*)nullptr, W, ncols,
                       (int64_t)n, (int64_t)n, 3, (const int*)info);
  }
  KP_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(kp_unpad_kernel, dim3((W + 255) / 256, ncols, nb), dim3(256), 0, st, Cp, n, W, ncols, K_dev, ncp, k_first,
                     k_cap);
  KP_HIP(ctx, hipGetLastError());
  return KP_OK;
}

// G_dev, C_dev: W x W / W x ncols column-major on the device (not modified); K_dev: W x ncols.
int kp_chol_solve_dev(kp_ctx* ctx, double* G_dev, double* C_dev, int W, int ncols, double* K_dev, hipStream_t st, hipEvent_t pad_done,
                      int* sticky) {
  return kp_chol_solve_batch_dev(ctx, G_dev, C_dev, W, ncols, 1, 0, K_dev, 0, 0, st, pad_done, sticky);
}

// smallest pivot of the factorisation relative to its original diagonal entry, min_i L_ii^2 / G_ii: ~1 / cond(G) - what
// the normal equations lose against the QR solve of `\` is cond(G) eps, so the caller can tell whether K needs the
// refinement pass over the data (kp_fit_refine) at all
// host_out (page-locked, device-mapped words of the context, or nullptr): [0] = the factorisation's info word, [1] = the ratio -
// written by the kernel itself, so the read-back needs no copy command behind it (4 us of the one-fit latency)
__global__ __launch_bounds__(256) void kp_pivot_ratio_kernel(const double* __restrict__ Lp, int n, const double* __restrict__ G, int W,
                                                             double* __restrict__ out, const int* __restrict__ info, double* host_out) {
  __shared__ double red[4];
  double r = 1e300;
  for (int i = threadIdx.x; i < W; i += 256) {
    const double l = Lp[(size_t)i * n + i], g = G[(size_t)i * W + i];
    r = fmin(r, g > 0.0 ? l * l / g : 0.0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r = fmin(r, __shfl_xor(r, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    out[0] = v;
    if (host_out) {
      host_out[1] = v;
      *reinterpret_cast<int*>(host_out) = info[0];
      __threadfence_system();
    }
  }
}

static int check_info(kp_ctx* ctx) {
  // info word sits behind the padded buffers of workspace 5; read it back (stream is synced by callers)
  return KP_OK;
}

// phase 1: only queue the ratio kernel (it stores the info word and the ratio into the context's page-locked words itself);
// phase 2: it was queued before - wait for the stream and read the words; 0: both.
static int read_chol_info(kp_ctx* ctx, int W, int ncols, int* bad, const double* G_dev = nullptr, hipStream_t st = nullptr, int phase = 0) {
  if (!st) st = ctx->stream;
  const int n = (W + 15) / 16 * 16;
  const size_t off = kp_chol_info_offset(W, ncols);
  double* ratio_dev = (double*)((char*)ctx->ws[5] + off + 8);
  if (G_dev && phase != 2) {     // the factor is still in the padded buffer at the head of workspace 5
    hipLaunchKernelGGL(kp_pivot_ratio_kernel, dim3(1), dim3(256), 0, st, (const double*)ctx->ws[5], n, G_dev, W, ratio_dev,
                       (const int*)((char*)ctx->ws[5] + off), ctx->pin_small);
    KP_HIP(ctx, hipGetLastError());
  }
  // info word (offset `off`) and pivot ratio (off + 8) sit side by side: the ratio kernel stores both into the context's
  // page-locked words itself; without it (no G) one 16-byte DMA brings them (two staged copies into pageable words were
  // 25 us of the one-fit latency, tools/fit_timeline.py)
  if (phase == 1) return KP_OK;
  if (ctx->pin_small) {
    if (!G_dev) KP_HIP(ctx, hipMemcpyAsync(ctx->pin_small, (char*)ctx->ws[5] + off, 16, hipMemcpyDeviceToHost, st));
    KP_HIP(ctx, hipStreamSynchronize(st));
    int info = 0;
    memcpy(&info, ctx->pin_small, sizeof(int));
    *bad = info;
    if (G_dev) ctx->last_pivot_ratio = ctx->pin_small[1];
  } else {
    if (G_dev) KP_HIP(ctx, hipMemcpyAsync(&ctx->last_pivot_ratio, ratio_dev, sizeof(double), hipMemcpyDeviceToHost, st));
    KP_HIP(ctx, hipMemcpyAsync(bad, (char*)ctx->ws[5] + off, sizeof(int), hipMemcpyDeviceToHost, st));
    KP_HIP(ctx, hipStreamSynchronize(st));
  }
  (void)check_info;
  return KP_OK;
}

// timers: 0 = fused lift+Gram kernel, 6 = partial-tile reduction, 1 = solve (when run)
static void collect_gram_timers(kp_ctx* ctx, bool solved) {
  float ms = 0;
  if (ctx->ring_n > 0) {                          // pipelined fits: mean over the last ring_n Gram launches
    double sum = 0.0;
    int cnt = 0;
    for (int i = 0; i < ctx->ring_n; ++i) {
      const int p = (ctx->ring_pos - 1 - i + 2 * 64) % 64;
      if (hipEventElapsedTime(&ms, ctx->ring[2 * p], ctx->ring[2 * p + 1]) == hipSuccess) {
        sum += ms;
        ++cnt;
      }
    }
    if (cnt) ctx->timers[0] = (float)(sum / cnt);
    ctx->timers[7] = (float)cnt;
    ctx->ring_n = 0;
  } else if (hipEventElapsedTime(&ms, ctx->evp[0], ctx->evp[1]) == hipSuccess) {
    ctx->timers[0] = ms;
    ctx->timers[7] = 1.0f;
  }
  if (hipEventElapsedTime(&ms, ctx->evp[ctx->reduce_timed_from], ctx->evp[2]) == hipSuccess) ctx->timers[6] = ms;
  if (solved && hipEventElapsedTime(&ms, ctx->evp[2], ctx->evp[3]) == hipSuccess) ctx->timers[1] = ms;
  (void)hipGetLastError();   // an event pair that was not recorded in this mode leaves a sticky error behind: not a failure
}

static int ensure_gc(kp_ctx* ctx, int W, int slots = 2) {
  size_t need = (size_t)std::max(2, slots) * 2 * W * W * 8;   // [G | C] buffers: asynchronous fits alternate between two / queue in a ring
  if (ctx->GC_bytes < need) {
    if (ctx->GC) (void)hipFree(ctx->GC);
    ctx->GC = nullptr;
    ctx->GC_bytes = 0;
    KP_HIP(ctx, hipMalloc((void**)&ctx->GC, need));
    ctx->GC_bytes = need;
  }
  ctx->GC_W = W;
  return KP_OK;
}

static int ensure_kres(kp_ctx* ctx, int W, int n) {
  size_t need = (size_t)n * W * W * 8;
  if (ctx->Kres_bytes < need) {
    if (ctx->Kres) (void)hipFree(ctx->Kres);
    ctx->Kres = nullptr;
    ctx->Kres_bytes = 0;
    KP_HIP(ctx, hipMalloc((void**)&ctx->Kres, need));
    ctx->Kres_bytes = need;
  }
  ctx->Kres_W = W;
  ctx->Kres_n = n;
  return KP_OK;
}

int kp_ensure_gc(kp_ctx* ctx, int W) { return ensure_gc(ctx, W); }

extern "C" int kp_fit_gram(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, double* G, double* C) {
  if (!ctx || !basis || !snaps) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_gram: NULL handle") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const int W = basis->dev.W;
  int rc = ctx->async_pending ? kp_synchronize(ctx) : KP_OK;
  if (rc) return rc;
  ctx->reserve_cus = 0;
  rc = ensure_gc(ctx, W);
  if (rc) return rc;
  rc = kp_gram_dispatch(ctx, basis, snaps, ctx->GC);
  if (rc) return rc;
  size_t bW = (size_t)W * W * 8;
  if (G) KP_HIP(ctx, hipMemcpyAsync(G, ctx->GC, bW, hipMemcpyDeviceToHost, ctx->stream));
  if (C) KP_HIP(ctx, hipMemcpyAsync(C, ctx->GC + (size_t)W * W, bW, hipMemcpyDeviceToHost, ctx->stream));
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  collect_gram_timers(ctx, false);
  return KP_OK;
}

extern "C" int kp_fit_solve(kp_ctx* ctx, const double* G, const double* C, int W, int ncols, double* K) {
  if (!ctx || !G || !C || !K || W < 1 || ncols < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_solve: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  size_t bG = (size_t)W * W * 8, bC = (size_t)W * ncols * 8;
  char* ws = (char*)ctx->workspace(6, bG + 2 * bC);
  if (!ws) return ctx->fail(KP_ERR_HIP, "kp_fit_solve: out of device memory");
  double* Gd = (double*)ws;
  double* Cd = (double*)(ws + bG);
  double* Kd = (double*)(ws + bG + bC);
  KP_HIP(ctx, hipMemcpyAsync(Gd, G, bG, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(Cd, C, bC, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = kp_chol_solve_dev(ctx, Gd, Cd, W, ncols, Kd);
  if (rc) return rc;
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(K, Kd, bC, hipMemcpyDeviceToHost, ctx->stream));
  int bad = 0;
  rc = read_chol_info(ctx, W, ncols, &bad, Gd);
  if (rc) return rc;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->timers[1] = ms;
  ctx->last_rank = W;
  if (bad) {
    // rank-deficient dictionary: MATLAB's `\` warns and returns a basic solution (QR with column pivoting); same here
    int r = 0;
    rc = kp_pivchol_solve_dev(ctx, Gd, Cd, W, ncols, Kd, &r);
    if (rc) return rc;
    ctx->last_rank = r;
    KP_HIP(ctx, hipMemcpyAsync(K, Kd, bC, hipMemcpyDeviceToHost, ctx->stream));
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->err = "warning: Gram matrix is rank deficient; basic solution returned (kp_fit_last_rank)";
  }
  return KP_OK;
}

extern "C" int kp_fit_last_pivot_ratio(const kp_ctx* ctx, double* ratio) {
  if (!ctx || !ratio) return KP_ERR_ARG;
  *ratio = ctx->last_pivot_ratio;
  return KP_OK;
}

extern "C" int kp_fit_last_rank(const kp_ctx* ctx, int* rank) {
  if (!ctx || !rank) return KP_ERR_ARG;
  *rank = ctx->last_rank;
  return KP_OK;
}

// Asynchronous pipeline, deferred solves (default): the queued Gram pairs are factored and solved by one batched launch
// sequence on the Gram stream (Cholesky: one workgroup per fit, all at once; TRSM: W/16 workgroups per fit).
static int solve_batch_size() {
  static const int v = [] { const char* e = getenv("KP_SOLVE_BATCH"); return e ? std::max(0, atoi(e)) : 128; }();   // (128 against 64: 0.4171 against 0.4191 ms per pipelined fit; capped by the result ring)
  return v;
}

static int flush_solves(kp_ctx* ctx) {
  if (ctx->pend_solves == 0) return KP_OK;
  const int W = ctx->pend_W;
  KP_HIP(ctx, hipEventRecord(ctx->ev_solve0, ctx->stream));
  int rc = kp_chol_solve_batch_dev(ctx, ctx->GC, ctx->GC + (size_t)W * W, W, W, ctx->pend_solves, (size_t)2 * W * W, ctx->Kres, ctx->pend_first,
                                   ctx->kring_cap, ctx->stream, nullptr, ctx->sticky_info);
  ctx->pend_solves = 0;
  if (rc) return rc;
  KP_HIP(ctx, hipEventRecord(ctx->ev_solve1, ctx->stream));
  return KP_OK;
}

int kp_flush_pending(kp_ctx* ctx) { return flush_solves(ctx); }

extern "C" int kp_fit(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps, const double* lasso, int n_lasso,
                      double* K_out) {
  // (a context whose [G | C] was handed in - kp_multi_fit's peers - needs no snapshot object)
  if (!ctx || !basis || (!snaps && !ctx->gc_preloaded) || n_lasso < 1) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  const int W = basis->dev.W, N = basis->dev.N;
  const bool all_ls = [&] {
    for (int i = 0; i < n_lasso; ++i)
      if (lasso && lasso[i] < 1e6) return false;
    return true;
  }();
  // (wide dictionaries, W > 512, take the synchronous path: a ring of 128 [G | C] pairs of that size would be tens of GB)
  const bool go_async = !K_out && n_lasso == 1 && all_ls && ctx->stream2 && ctx->sticky_info && !ctx->reduce_grams && !ctx->gc_preloaded && W <= 512 && !getenv("KP_NO_ASYNC");
  int rc;
  // The [G | C] ring may hold queued Gram pairs that are not solved yet (deferred, batched solves): drain the pipeline
  // BEFORE any buffer of it can be reallocated for another width - ensure_gc frees and reallocates GC when the new W needs
  // more room - and before a synchronous fit reuses it.
  if (ctx->async_pending && (!go_async || ctx->pend_basis != (const void*)basis || ctx->pend_Ns != snaps->Ns || ctx->pend_W != W)) {
    rc = kp_synchronize(ctx);
    if (rc) return rc;
  }
  rc = ensure_gc(ctx, W);
  if (rc) return rc;
  if (go_async) {
    // ---- asynchronous pipeline: this fit's solve (stream2) overlaps the next fit's Gram (stream) ----
    // two [G | C] buffers alternate, so this Gram's reduction only has to wait for the pad kernel (the solve's
    // copy of G | C) of the fit before the previous one -- never for the solve that is running right now
    // the two G | C halves and the split-partial buffers are addressed with THIS call's W and partial size: fits in
    // flight with another dictionary or snapshot count would overlap them, so such a change drains the pipeline first
    if (ctx->async_pending && (ctx->pend_basis != (const void*)basis || ctx->pend_Ns != snaps->Ns || ctx->pend_W != W)) {
      rc = kp_synchronize(ctx);
      if (rc) return rc;
    }
    if (ctx->batch_closed) {
      ctx->async_count = 0;
      ctx->batch_closed = false;
    }
    rc = ensure_kres(ctx, W, ctx->kring_cap);
    if (rc) return rc;
    ctx->kres_is_ring = true;
    ctx->pend_basis = (const void*)basis;
    ctx->pend_Ns = snaps->Ns;
    ctx->pend_W = W;
    const int sbatch = std::min(solve_batch_size(), ctx->kring_cap);
    if (sbatch >= 1) {
      // deferred solves: this fit's [G | C] goes into the next slot of the ring; nothing else runs beside the Gram kernel
      rc = ensure_gc(ctx, W, sbatch);
      if (rc) return rc;
      {   // the batched solve's padded workspace, sized for a full batch NOW: growing it at the first full flush would put a
          // hipFree / hipMalloc of ~150 MB (tens of ms) in the middle of the pipeline
        const int np_ = (W + 15) / 16 * 16;
        const size_t per = (size_t)np_ * np_ * 8 * 2 + (size_t)(np_ / 16) * 256 * 8;
        if (!ctx->workspace(5, (size_t)sbatch * per + (size_t)sbatch * 4 + 64)) return ctx->fail(KP_ERR_HIP, "kp_fit: out of device memory");
      }
      if (ctx->pend_solves == 0) ctx->pend_first = ctx->async_count;
      double* GCs = ctx->GC + (size_t)ctx->pend_solves * 2 * W * W;
      ctx->reserve_cus = 0;
      ctx->reduce_stream = nullptr;
      ctx->ring_timing = true;
      rc = kp_gram_dispatch(ctx, basis, snaps, GCs);
      ctx->ring_timing = false;
      if (rc) return rc;
      ++ctx->pend_solves;
      ++ctx->async_count;
      ctx->async_pending = true;
      if (ctx->pend_solves >= sbatch) return flush_solves(ctx);
      return KP_OK;
    }
    double* Kslot = ctx->Kres + (size_t)(ctx->async_count % ctx->kring_cap) * W * W;
    const int flip = ctx->gc_flip;
    ctx->gc_flip ^= 1;
    double* GCb = ctx->GC + (size_t)flip * 2 * W * W;
    hipEvent_t evp = flip ? ctx->ev_pad_done2 : ctx->ev_pad_done;
    bool& pend = flip ? ctx->pad_pending2 : ctx->pad_pending;
    if (pend) KP_HIP(ctx, hipStreamWaitEvent(ctx->stream, evp, 0));
    // CUs left free for the solve stream (Cholesky: one workgroup; TRSM: W/16 workgroups)
    static const int reserve = [] { const char* e = getenv("KP_RESERVE_CUS"); return e ? atoi(e) : 24; }();
    ctx->reserve_cus = reserve;
    // The Kronecker kernel's partial reduction runs on the solve stream: reduce + Cholesky + TRSM (0.43 ms, serial)
    // are shorter than the Gram kernel (0.46 ms), so the Gram stream issues kernels back to back.
    // KP_REDUCE_ON_GRAM_STREAM=1 keeps it behind the Gram kernel.
    static const bool reduce_on_solve = getenv("KP_REDUCE_ON_GRAM_STREAM") == nullptr;
    ctx->reduce_stream = reduce_on_solve ? ctx->stream2 : nullptr;
    ctx->part_flip = flip;
    ctx->solve_chained = false;
    rc = kp_gram_dispatch(ctx, basis, snaps, GCb);
    ctx->reduce_stream = nullptr;
    if (rc) return rc;
    if (!ctx->solve_chained) {
      KP_HIP(ctx, hipEventRecord(ctx->ev_gram_done, ctx->stream));
      KP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_gram_done, 0));
    }
    KP_HIP(ctx, hipEventRecord(ctx->ev_solve0, ctx->stream2));
    rc = kp_chol_solve_dev(ctx, GCb, GCb + (size_t)W * W, W, W, Kslot, ctx->stream2, evp, ctx->sticky_info);
    if (rc) return rc;
    KP_HIP(ctx, hipEventRecord(ctx->ev_solve1, ctx->stream2));
    ++ctx->async_count;
    pend = true;
    ctx->async_pending = true;
    return KP_OK;
  }
  if (ctx->async_pending) {
    rc = kp_synchronize(ctx);
    if (rc) return rc;
  }
  rc = ensure_kres(ctx, W, n_lasso);
  if (rc) return rc;
  ctx->reserve_cus = 0;
  ctx->kres_is_ring = false;
  ctx->batch_closed = true;
  if (!ctx->gc_preloaded) {
    rc = kp_gram_dispatch(ctx, basis, snaps, ctx->GC);  // records ev0/ev1 around gram+reduce
    if (rc) return rc;
  }
  if (ctx->reduce_grams) {   // one fit sharded over snapshots: the only exchange is this all-reduce of [G | C]
    rc = kp_comm_allreduce_dev(ctx, ctx->GC, (size_t)2 * W * W, ctx->stream);
    if (rc) return rc;
  }
  double* Gd = ctx->GC;
  double* Cd = ctx->GC + (size_t)W * W;
  bool need_ls = false;
  for (int i = 0; i < n_lasso; ++i) need_ls |= (!lasso || !(lasso[i] < 1e6));
  int ls_index = -1;
  if (ctx->test_hooks) {                    // (contexts created with KP_TEST_HOOKS set: a remembered rank that is wrong in a chosen way)
    if (const char* e = getenv("KP_RANK_HINT_TEST")) basis->rank_hint = atoi(e);
  }
  // (see below) least-squares values only, the narrow path, a second stream to run on
  const bool concurrent = basis->rank_hint > 0 && all_ls && ctx->stream2 && W <= 16 * 4 * TR_MAXJ && !ctx->reduce_grams && !ctx->gc_preloaded &&
                          !getenv("KP_NO_RANK_HINT");
  bool conc_done = false, conc_copied = false;
  int conc_bad = 0, conc_rank = 0;
  const size_t k_bytes = (size_t)n_lasso * W * W * 8;
  double* k_pin = (K_out && k_bytes <= ((size_t)64 << 20)) ? (double*)kp_pinned_scratch(ctx, k_bytes) : nullptr;
  // the least-squares solution is also the inactive-constraint answer of the lasso path; all lasso values run as one batch
  std::vector<double> tv;
  std::vector<double*> tdst;
  for (int i = 0; i < n_lasso; ++i) {
    double* Ki = ctx->Kres + (size_t)i * W * W;
    bool is_ls = (!lasso || !(lasso[i] < 1e6));  // Ksysid.m:1068 (Inf was mapped to 1e6 at :155-157)
    if (is_ls) {
      if (ls_index >= 0) {
        KP_HIP(ctx, hipMemcpyAsync(Ki, ctx->Kres + (size_t)ls_index * W * W, (size_t)W * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
      } else if (concurrent) {
        // The dictionary's last fit was rank deficient: the plain factorisation - still what DECIDES, so that no result depends
        // on the history - runs on the second stream into a scratch K, beside the rank-revealing solve that will most likely
        // be the answer, instead of in front of it with a host round trip between the two (~50 us of a 0.64 ms fit).
        double* Ks = (double*)ctx->workspace(19, (size_t)W * W * 8);
        if (!Ks) return ctx->fail(KP_ERR_HIP, "kp_fit: out of device memory");
        KP_HIP(ctx, hipEventRecord(ctx->ev_gram_done, ctx->stream));
        KP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_gram_done, 0));
        rc = kp_chol_solve_dev(ctx, Gd, Cd, W, W, Ks, ctx->stream2, nullptr, nullptr);
        if (rc) return rc;
        ls_index = i;
        if (ctx->pin_small) {                                                                   // (its verdict is ready long before it is read)
          rc = read_chol_info(ctx, W, W, &conc_bad, Gd, ctx->stream2, 1);
          if (rc) return rc;
        }
        // (one value: its K goes to the caller's page-locked block in front of the solve's own synchronisation, not behind it)
        conc_copied = K_out && n_lasso == 1;
        rc = kp_pivchol_solve_dev(ctx, Gd, Cd, W, W, Ki, &conc_rank, basis->rank_hint, conc_copied ? (k_pin ? (void*)k_pin : (void*)K_out) : nullptr,
                                  k_bytes, ctx->evp[3], k_pin ? 1 : 0);      // (synchronises the first stream)
        if (rc) return rc;
        rc = read_chol_info(ctx, W, W, &conc_bad, Gd, ctx->stream2, ctx->pin_small ? 2 : 0);  // (and the second)
        if (rc) return rc;
        if (!conc_bad) {                                                                        // full rank after all
          KP_HIP(ctx, hipMemcpyAsync(Ki, Ks, (size_t)W * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
          conc_copied = false;
        }
        conc_done = true;
      } else {
        rc = kp_chol_solve_dev(ctx, Gd, Cd, W, W, Ki);
        if (rc) return rc;
        ls_index = i;
      }
    } else {
      tv.push_back(lasso[i] * N);   // t = lasso*N, Ksysid.m:996
      tdst.push_back(Ki);
    }
  }
  if (!tv.empty()) {
    KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    rc = kp_lasso_batch_dev(ctx, Gd, Cd, W, W, tv.data(), (int)tv.size(), 20000, 1e-10, tdst.data(), nullptr, nullptr);
    if (rc) return rc;
    KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    // the lasso solvers take the same growable page-locked block for their records: a request larger than k_bytes has
    // replaced (and freed) the block k_pin pointed into - take it again (nothing of theirs is live any more)
    if (k_pin) k_pin = (double*)kp_pinned_scratch(ctx, k_bytes);
  }
  (void)need_ls;
  if (!conc_done) KP_HIP(ctx, hipEventRecord(ctx->evp[3], ctx->stream));
  // K to the caller: the caller's array is pageable, and a device-to-host copy into pageable memory goes through the
  // runtime's own staging (0.16 ms for the 0.9 MB of one W = 336 matrix); a direct DMA into the context's page-locked
  // block and a memcpy from there take 0.05 ms
  // (a dictionary whose previous fit was rank deficient: the copy of a K that will most likely be replaced waits for the verdict)
  const bool k_late = K_out && ls_index >= 0 && basis->rank_hint > 0 && !conc_done;
  // (the copy engine here, not kp_copy_to_host_async's stores: it runs beside the ratio kernel that follows - 0.815 against 0.823 ms)
  if (K_out && !k_late && !conc_copied) KP_HIP(ctx, hipMemcpyAsync(k_pin ? k_pin : K_out, ctx->Kres, k_bytes, hipMemcpyDeviceToHost, ctx->stream));
  int bad = 0;
  if (conc_done) {
    bad = conc_bad;
    if (!conc_copied) KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else if (ls_index >= 0) {
    rc = read_chol_info(ctx, W, W, &bad, Gd);
    if (rc) return rc;
    if (k_late && !bad) {
      KP_HIP(ctx, hipMemcpyAsync(k_pin ? k_pin : K_out, ctx->Kres, k_bytes, hipMemcpyDeviceToHost, ctx->stream));
      KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  } else {
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  collect_gram_timers(ctx, true);
  if (!tv.empty()) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->timers[3] = ms;
  }
  ctx->last_rank = W;
  if (!bad && ls_index >= 0) basis->rank_hint = 0;
  if (k_pin && (!bad || conc_done)) memcpy(K_out, k_pin, k_bytes);      // (conc_done: whichever K it is, it is final)
  if (bad && conc_done) {
    ctx->last_rank = conc_rank;
    basis->rank_hint = conc_rank < W ? conc_rank : 0;
    ctx->err = "warning: Gram matrix is rank deficient; basic solution returned (kp_fit_last_rank)";
    return KP_OK;
  }
  if (bad) {
    // rank-deficient dictionary (Ksysid.m:1069 on the arm data without dim_red): basic solution + rank, like MATLAB's `\`
    int r = 0;
    double* Kls = ctx->Kres + (size_t)ls_index * W * W;
    rc = kp_pivchol_solve_dev(ctx, Gd, Cd, W, W, Kls, &r, basis->rank_hint);
    if (rc) return rc;
    ctx->last_rank = r;
    basis->rank_hint = r < W ? r : 0;
    for (int i = 0; i < n_lasso; ++i)
      if (i != ls_index && (!lasso || !(lasso[i] < 1e6)))
        KP_HIP(ctx, hipMemcpyAsync(ctx->Kres + (size_t)i * W * W, Kls, (size_t)W * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (K_out) KP_HIP(ctx, hipMemcpyAsync(k_pin ? k_pin : K_out, ctx->Kres, k_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (k_pin) memcpy(K_out, k_pin, k_bytes);
    ctx->err = "warning: Gram matrix is rank deficient; basic solution returned (kp_fit_last_rank)";
  }
  return KP_OK;
}

extern "C" int kp_synchronize(kp_ctx* ctx) {
  if (!ctx) return KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rcf = flush_solves(ctx);
    if (rcf) return rcf;
  }
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream2) KP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  int bad = 0;
  if (ctx->async_pending) {
    collect_gram_timers(ctx, false);
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1) == hipSuccess) ctx->timers[1] = ms;
    KP_HIP(ctx, hipMemcpy(&bad, ctx->sticky_info, sizeof(int), hipMemcpyDeviceToHost));
    if (bad) KP_HIP(ctx, hipMemset(ctx->sticky_info, 0, sizeof(int)));
  }
  ctx->async_pending = false;
  ctx->pad_pending = false;
  ctx->pad_pending2 = false;
  ctx->batch_closed = true;
  if (bad) return ctx->fail(KP_ERR_NOT_SPD, "kp_synchronize: a deferred fit hit a Gram matrix that is not numerically positive definite");
  return KP_OK;
}

extern "C" int kp_fit_get_K(kp_ctx* ctx, int index, int W, double* K) {
  if (!ctx || !K || index < 0 || W != ctx->Kres_W) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_get_K: bad argument") : KP_ERR_ARG;
  KP_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rc = kp_synchronize(ctx);
    if (rc) return rc;
  }
  size_t slot = (size_t)index;
  if (ctx->kres_is_ring) {
    // asynchronous batch: fit number `index` since the previous kp_synchronize; the ring keeps the last kring_cap of them
    if (index >= ctx->async_count || index < ctx->async_count - ctx->kring_cap)
      return ctx->fail(KP_ERR_ARG, "kp_fit_get_K: that fit of the asynchronous batch is not (or no longer) in the result ring "
                                   "(kp_fit_async_slots sets its size)");
    slot = (size_t)(index % ctx->kring_cap);
  } else if (index >= ctx->Kres_n) {
    return ctx->fail(KP_ERR_ARG, "kp_fit_get_K: index out of range");
  }
  KP_HIP(ctx, hipMemcpy(K, ctx->Kres + slot * W * W, (size_t)W * W * 8, hipMemcpyDeviceToHost));
  return KP_OK;
}

extern "C" int kp_fit_async_slots(kp_ctx* ctx, int n_slots) {
  if (!ctx || n_slots < 1 || n_slots > (1 << 20)) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_async_slots: bad argument") : KP_ERR_ARG;
  int rc = kp_synchronize(ctx);
  if (rc) return rc;
  ctx->kring_cap = n_slots;
  ctx->kres_is_ring = false;   // whatever the ring held is addressed with the old size: start over
  ctx->Kres_n = 0;
  ctx->async_count = 0;
  return KP_OK;
}

// One fit whose snapshot pairs are sharded over the ranks of the communicator: every rank runs the fused Gram kernel on
// its shard, ONE all-reduce of [G | C] (2 W^2 doubles, device to device over RCCL) is the only exchange, and every
// rank solves the same system (identical K on all ranks).  Without a communicator this is kp_fit.
extern "C" int kp_fit_sharded(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps_local, const double* lasso, int n_lasso,
                              double* K_out) {
  if (!ctx) return KP_ERR_ARG;
  if (ctx->async_pending) {
    int rc0 = kp_synchronize(ctx);
    if (rc0) return rc0;
  }
  ctx->reduce_grams = true;
  std::vector<double> tmpK;
  double* dst = K_out;
  if (!dst && basis) {            // keep the synchronous path (K_out == NULL selects the asynchronous pipeline otherwise)
    tmpK.resize((size_t)std::max(1, n_lasso) * basis->dev.W * basis->dev.W);
    dst = tmpK.data();
  }
  int rc = kp_fit(ctx, basis, snaps_local, lasso, n_lasso, dst);
  ctx->reduce_grams = false;
  return rc;
}

// the Gram pair alone, summed over the ranks' shards
extern "C" int kp_fit_gram_sharded(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* snaps_local, double* G, double* C) {
  if (!ctx || !basis || !snaps_local) return ctx ? ctx->fail(KP_ERR_ARG, "kp_fit_gram_sharded: NULL handle") : KP_ERR_ARG;
  int rc = kp_fit_gram(ctx, basis, snaps_local, nullptr, nullptr);
  if (rc) return rc;
  const int W = basis->dev.W;
  rc = kp_comm_allreduce_dev(ctx, ctx->GC, (size_t)2 * W * W, ctx->stream);
  if (rc) return rc;
  size_t bW = (size_t)W * W * 8;
  if (G) KP_HIP(ctx, hipMemcpyAsync(G, ctx->GC, bW, hipMemcpyDeviceToHost, ctx->stream));
  if (C) KP_HIP(ctx, hipMemcpyAsync(C, ctx->GC + (size_t)W * W, bW, hipMemcpyDeviceToHost, ctx->stream));
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KP_OK;
}

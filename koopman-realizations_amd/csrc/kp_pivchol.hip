// Rank-revealing solve of the normal equations for RANK-DEFICIENT dictionaries.
//
// MATLAB's `K = Px \ Py` (Ksysid.m:1069) on a rank-deficient Px (the arm's marker coordinates without dim_red: poly-3
// bilinear has rank 252 of 336) is a QR solve with column pivoting: it warns, reports the rank and
// returns a BASIC solution - non-zero only in the rows of the r columns the pivoting selected.  The same greedy choice
// in Gram space is Cholesky with diagonal pivoting (pivot = largest remaining diagonal = largest remaining column norm).
// Every basic solution over a column subset that spans range(Px) has the same residual Px K - Py, which is what the
// parity test compares with LAPACK's pivoted QR.
//
// Round 5: BLOCKED and over many workgroups (rounds 2-4: one workgroup, left-looking, every step re-read all earlier
// columns of L from L2 - 85 MB through one CU at W = 336, 1.9 ms).  Now, per block of nb pivots (nb x W doubles of LDS):
//   kp_pivchol_panel_kernel  ONE workgroup, the serial part: pivot search over the remaining diagonal (one barrier per
//       step: every wave's candidate goes to LDS, everybody reduces the candidates), the pivot column from the TRAILING
//       matrix (one coalesced read; what earlier BLOCKS contribute is already in it) minus the columns of THIS block (LDS);
//   kp_pivchol_update_kernel  every CU: trailing matrix -= panel panel' (columns already chosen are skipped).
// The launches of all blocks are queued without a host round trip; once the remaining diagonal is rounding noise the panel
// kernel sets a flag and the launches behind it return at once.  The factor it leaves - rows in pivot order - IS the Cholesky
// factor of the selected r x r system, so the solve needs no second factorisation: gather, kp_factor_substitute_dev.
// Any width the library fits (the panel shrinks with W: 32 pivots per block up to W = 560, 6 at W = 2 940).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kp_internal.h"

namespace {

struct PivState {        // device words shared by the launches of one factorisation
  int rank;              // valid once done != 0
  int done;
  int pad[2];
  double d1;             // the first pivot
};

constexpr int PC_RPT_MAX = 4;   // rows per thread of the panel kernel: W <= 4096

}  // namespace

// A := G (working copy: the trailing matrix), dg := diag(G), ipos := -1 (row not chosen), state := 0
__global__ __launch_bounds__(256) void kp_pivchol_init_kernel(const double* __restrict__ G, int W, double* __restrict__ A, double* __restrict__ dg,
                                                              int* __restrict__ ipos, PivState* __restrict__ stt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (int64_t)W * W) A[e] = G[e];
  if (e < W) {
    dg[e] = G[e * (int64_t)W + e];
    ipos[e] = -1;
  }
  if (e == 0) {
    stt->rank = 0;
    stt->done = 0;
    stt->d1 = 0.0;
  }
}

// ---- wave-level arg max: the extreme VALUE by a DPP reduction inside the 16-lane rows (2 moves + 1 v_max per step) and four
// v_readlane across them, then the first lane that holds it (ballot + s_ff1) hands over its index - ~220 cycles where
// carrying (value, index) pairs through __shfl_xor (three ds_bpermute per step) took > 1000.  Ties go to the lowest lane.
template <int CTRL>
__device__ __forceinline__ double pc_dpp_mov(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pc_lane_get(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void pc_wave_argmax(double& v, int& idx) {
  double m = v;
  m = fmax(m, pc_dpp_mov<0xB1>(m));      // quad_perm [1,0,3,2]
  m = fmax(m, pc_dpp_mov<0x4E>(m));      // quad_perm [2,3,0,1]
  m = fmax(m, pc_dpp_mov<0x141>(m));     // row_half_mirror
  m = fmax(m, pc_dpp_mov<0x140>(m));     // row_mirror: every lane holds the maximum of its 16-lane row
  const double w = fmax(fmax(pc_lane_get(m, 0), pc_lane_get(m, 16)), fmax(pc_lane_get(m, 32), pc_lane_get(m, 48)));
  const unsigned long long mask = __ballot(v == w);
  idx = mask ? __builtin_amdgcn_readlane(idx, __ffsll((long long)mask) - 1) : 0x7fffffff;
  v = w;
}

// Steps k0 .. k0 + nb - 1.  L[k][i] (column k of the factor contiguous over the rows i), Pn[c][i] the same block in LDS.
// A step is one dependent chain - pivot search, pivot column, next search - so what it costs is latency: the column's global
// load is issued as soon as the pivot is known and the products with the block's earlier columns (LDS) run under it; 1 / sqrt
// is the hardware estimate + two Newton steps; ONE barrier per step.
template <int NT, int RPT>
__global__ __launch_bounds__(NT) void kp_pivchol_panel_kernel(const double* __restrict__ A, int W, int k0, int nb, double rel_tol, double* __restrict__ L,
                                                              double* __restrict__ dg, int* __restrict__ ipos, int* __restrict__ perm,
                                                              PivState* __restrict__ stt) {
  extern __shared__ double Pn[];         // [nb][W]
  __shared__ double red_v[2][NT / 64];
  __shared__ int red_i[2][NT / 64];
  if (stt->done) return;
  const int tid = threadIdx.x;
  double d[RPT];
  bool used[RPT], on[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int i = tid + r * NT;
    on[r] = i < W;
    d[r] = on[r] ? dg[i] : -1.0;
    used[r] = on[r] ? ipos[i] >= 0 : true;
  }
  double d1 = stt->d1;
  int k = k0, stop = 0;
  for (int j = 0; j < nb && k < W; ++j, ++k) {
    // arg max of the remaining diagonal
    double v = -1.0;
    int vi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int i = tid + r * NT;
      if (!used[r] && (d[r] > v || (d[r] == v && i < vi))) { v = d[r]; vi = i; }
    }
    pc_wave_argmax(v, vi);
    if ((tid & 63) == 0) { red_v[j & 1][tid >> 6] = v; red_i[j & 1][tid >> 6] = vi; }
    __syncthreads();                     // (also: the column written in step j - 1 is visible to everybody)
    double pv = red_v[j & 1][0];
    int p = red_i[j & 1][0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
      const double ov = red_v[j & 1][w];
      const int oi = red_i[j & 1][w];
      if (ov > pv || (ov == pv && oi < p)) { pv = ov; p = oi; }
    }
    if (k == 0) d1 = pv;
    if (!(pv > rel_tol * d1) || !(pv > 0.0)) { stop = 1; break; }   // what is left is rounding noise of the first pivots: rank = k
    double a[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) a[r] = on[r] ? A[tid + r * NT + (size_t)p * W] : 0.0;    // in flight across the products below
    double rinv = __builtin_amdgcn_rsq(pv);
    rinv = rinv * (1.5 - 0.5 * pv * rinv * rinv);
    rinv = rinv * (1.5 - 0.5 * pv * rinv * rinv);
    const double sq = pv * rinv;
    double acc0[RPT], acc1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc0[r] = acc1[r] = 0.0;
    int c = 0;
    for (; c + 1 < j; c += 2) {
      const double p0 = Pn[c * W + p], p1 = Pn[(c + 1) * W + p];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int i = on[r] ? tid + r * NT : 0;
        acc0[r] += Pn[c * W + i] * p0;
        acc1[r] += Pn[(c + 1) * W + i] * p1;
      }
    }
    if (c < j) {
      const double p0 = Pn[c * W + p];
#pragma unroll
      for (int r = 0; r < RPT; ++r) acc0[r] += Pn[c * W + (on[r] ? tid + r * NT : 0)] * p0;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int i = tid + r * NT;
      if (!on[r]) continue;
      const double s = a[r] - (acc0[r] + acc1[r]);
      double l = used[r] ? 0.0 : s * rinv;                  // rows already chosen have a zero below their own pivot (exactly)
      if (!used[r]) d[r] -= l * l;
      if (i == p) {
        l = sq;
        used[r] = true;
        perm[k] = p;
        ipos[i] = k;
      }
      Pn[j * W + i] = l;
      L[(size_t)k * W + i] = l;
    }
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    if (on[r]) dg[tid + r * NT] = d[r];
  if (tid == 0) {
    stt->d1 = d1;
    if (stop || k >= W) {
      stt->rank = k;
      stt->done = 1;
    }
  }
}

// The same block for the common widths (one row per thread, 32 pivots per block: W <= 560), built around the step's latency:
//  * a thread keeps its OWN row of the block in registers (lrow[c]; the steps are unrolled by hand so that the index is a
//    constant): the products with the block's earlier columns need one LDS broadcast read per column (the pivot row's
//    entry), all independent, and run under the pivot column's global load;
//  * the cross-wave stage of the pivot search is a second DPP reduction - every 16-lane row holds all NT / 64 wave
//    candidates (lane l reads candidate l mod NW) - instead of a serial scan over them;
//  * the pivot column comes from the trailing matrix in memory (one coalesced load, issued as soon as the pivot is known).
// What a step costs (0.93 us at W = 336: 34 us per launch of 32): NOT that load - a timing-only build without it
// (KP_PIV_ABL=1) takes the same time - but the step's own dependent instruction chain: two DPP reductions with their
// v_readlane / ballot tails, 1 / sqrt with two Newton steps, the products, at ~7 cycles per dependent vector instruction
// (tools/clock_probe).  (Fetching the columns of the runner-up candidates ahead was built on the assumption that the load
// bounds the step - the runner-up of step j is the pivot of step j + 1 in 192 of 251 steps on the arm data's W = 336 Gram -
// and taught two things about hipcc: it waits for EVERY outstanding load before the first use of any and before it reuses a
// destination register, and it copies the destination of a hand-issued load before the data has landed.)
template <int NT>
__global__ __launch_bounds__(NT) void kp_pivchol_panel32_kernel(const double* __restrict__ A, int W, int k0, int nb, double rel_tol, double* __restrict__ L,
                                                                double* __restrict__ dg, int* __restrict__ ipos, int* __restrict__ perm,
                                                                PivState* __restrict__ stt, int abl) {
  extern __shared__ double Pn[];         // [32][W]
  __shared__ double red_v[2][NT / 64];
  __shared__ int red_i[2][NT / 64];
  if (stt->done) return;
  const int tid = threadIdx.x;
  const bool on = tid < W;
  const int irow = on ? tid : 0;
  double d = on ? dg[tid] : -1.0;
  bool used = on ? ipos[tid] >= 0 : true;
  double d1 = stt->d1;
  double lrow[32];
  int k = k0, stop = 0;
  // one step, J a compile-time constant (lrow[] is indexed statically: registers); returns true when the block is finished
  auto step = [&](auto jc) -> bool {
    constexpr int j = decltype(jc)::value;
    if (j >= nb || k >= W) return true;
    double v = used ? -1.0 : d;
    int vi = tid;
    pc_wave_argmax(v, vi);
    if ((tid & 63) == 0) { red_v[j & 1][tid >> 6] = v; red_i[j & 1][tid >> 6] = vi; }
    // LDS-only barrier: __syncthreads() also drains the memory counter - every step would wait for the acknowledgement of the
    // previous step's stores of L (nobody in this kernel reads them back)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // best of the waves' candidates: a DPP reduction inside the 16-lane rows (each holds all of them), the first lane that
    // holds the maximum names the row (lowest wave = lowest index on ties)
    constexpr int NW = NT / 64;
    const double cv = red_v[j & 1][tid & (NW - 1)];
    const int ci = red_i[j & 1][tid & (NW - 1)];
    double m = cv;
    m = fmax(m, pc_dpp_mov<0xB1>(m)); m = fmax(m, pc_dpp_mov<0x4E>(m)); m = fmax(m, pc_dpp_mov<0x141>(m));
    if (NW > 8) m = fmax(m, pc_dpp_mov<0x140>(m));
    const double pv = pc_lane_get(m, 0);
    const unsigned long long mk = __ballot(cv == pv);
    const int p = __builtin_amdgcn_readlane(ci, __ffsll((long long)mk) - 1);
    if (k == 0) d1 = pv;
    if (!(pv > rel_tol * d1) || !(pv > 0.0)) { stop = 1; return true; }
    const double a = (abl & 1) ? (tid == p ? pv : 0.0) : A[irow + (size_t)p * W];   // in flight across the products below (abl: timing only)
    double rinv = __builtin_amdgcn_rsq(pv);
    rinv = rinv * (1.5 - 0.5 * pv * rinv * rinv);
    rinv = rinv * (1.5 - 0.5 * pv * rinv * rinv);
    const double sq = pv * rinv;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int c = 0; c < j; ++c) {
      if (c & 1) acc1 += lrow[c] * Pn[c * W + p];
      else acc0 += lrow[c] * Pn[c * W + p];
    }
    const double s = a - (acc0 + acc1);
    double l = used ? 0.0 : s * rinv;                       // rows already chosen have a zero below their own pivot (exactly)
    if (!used) d -= l * l;
    if (tid == p) {
      l = sq;
      used = true;
      perm[k] = p;
      ipos[tid] = k;
    }
    lrow[j] = l;
    if (on) {
      Pn[j * W + tid] = l;
      L[(size_t)k * W + tid] = l;
    }
    ++k;
    return false;
  };
  bool fin = false;
#define PC_STEP(J) if (!fin) fin = step(std::integral_constant<int, J>{});
  PC_STEP(0) PC_STEP(1) PC_STEP(2) PC_STEP(3) PC_STEP(4) PC_STEP(5) PC_STEP(6) PC_STEP(7)
  PC_STEP(8) PC_STEP(9) PC_STEP(10) PC_STEP(11) PC_STEP(12) PC_STEP(13) PC_STEP(14) PC_STEP(15)
  PC_STEP(16) PC_STEP(17) PC_STEP(18) PC_STEP(19) PC_STEP(20) PC_STEP(21) PC_STEP(22) PC_STEP(23)
  PC_STEP(24) PC_STEP(25) PC_STEP(26) PC_STEP(27) PC_STEP(28) PC_STEP(29) PC_STEP(30) PC_STEP(31)
#undef PC_STEP
  if (on) dg[tid] = d;
  if (tid == 0) {
    stt->d1 = d1;
    if (stop || k >= W) {
      stt->rank = k;
      stt->done = 1;
    }
  }
}

// Trailing matrix -= L_blk' L_blk over the steps [k0, k0 + nb) for the columns not chosen yet; tile 64 rows x 64 columns,
// thread = one row x 16 columns
__global__ __launch_bounds__(256) void kp_pivchol_update_kernel(double* __restrict__ A, int W, int k0, int nb, const double* __restrict__ L,
                                                                const int* __restrict__ ipos, const PivState* __restrict__ stt) {
  __shared__ double Li[32][64], Lc[32][64];
  if (stt->done) return;
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int e = tid; e < nb * 64; e += 256) {
    const int kk = e >> 6, x = e & 63;
    Li[kk][x] = i0 + x < W ? L[(size_t)(k0 + kk) * W + i0 + x] : 0.0;
    Lc[kk][x] = c0 + x < W ? L[(size_t)(k0 + kk) * W + c0 + x] : 0.0;
  }
  __syncthreads();
  const int x = tid & 63, q0 = (tid >> 6) * 16;
  const int i = i0 + x;
  if (i >= W) return;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  for (int kk = 0; kk < nb; ++kk) {
    const double li = Li[kk][x];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += li * Lc[kk][q0 + q];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = c0 + q0 + q;
    if (c < W && ipos[c] < 0) A[i + (size_t)c * W] -= acc[q];
  }
}

// Lp (n x n, n = W padded to 16): rows / columns < r = the factor's rows in pivot order (lower triangle), identity beyond;
// Cp (n x ncp): rows < r = C[perm, :], zero beyond.  r is read from the device state: no host round trip before the substitution.
__global__ void kp_pivchol_gather_kernel(const double* __restrict__ L, const double* __restrict__ C, int W, int ncols, const int* __restrict__ perm,
                                         const PivState* __restrict__ stt, int n, int ncp, double* __restrict__ Lp, double* __restrict__ Cp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nG = (int64_t)n * n;
  const int r = stt->rank;
  if (e < nG) {
    const int a = (int)(e % n), b = (int)(e / n);       // row a, column b
    double v = a == b ? 1.0 : 0.0;
    if (a < r && b < r) v = a >= b ? L[(size_t)b * W + perm[a]] : 0.0;
    Lp[e] = v;
  } else if (e < nG + (int64_t)n * ncp) {
    const int64_t f = e - nG;
    const int a = (int)(f % n), c = (int)(f / n);
    Cp[f] = (a < r && c < ncols) ? C[perm[a] + (size_t)c * W] : 0.0;
  }
}

__global__ __launch_bounds__(256) void kp_copy16_kernel(const double2* __restrict__ src, double2* __restrict__ dst, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) dst[e] = src[e];
}

// Device -> host copy of a result the caller waits for.  mapped (a page-locked, device-mapped destination such as
// kp_pinned_scratch's block) and a multiple of 16 bytes of at most 8 MB: plain stores over PCIe by a kernel - the runtime's copy
// engine moved the 0.9 MB of a W = 336 matrix at ~25 GB/s (36 us), the stores take ~20.  Otherwise hipMemcpyAsync.
int kp_copy_to_host_async(kp_ctx* ctx, const void* src_dev, void* dst_host, size_t bytes, int mapped, hipStream_t s) {
  if (mapped && bytes % 16 == 0 && bytes <= ((size_t)8 << 20) && !getenv("KP_NO_STORE_COPY")) {
    const int64_t nd = (int64_t)(bytes / 16);
    hipLaunchKernelGGL(kp_copy16_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, (const double2*)src_dev, (double2*)dst_host, nd);
    KP_HIP(ctx, hipGetLastError());
    return KP_OK;
  }
  KP_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
  return KP_OK;
}

// K (W x ncols, zeroed) [perm[a], c] = Ks[a, c] for a < r  (Ks with leading dimension n)
__global__ void kp_scatter_rows_kernel(const double* __restrict__ Ks, int W, int ncols, const int* __restrict__ perm, const PivState* __restrict__ stt,
                                       int n, double* __restrict__ K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)W * ncols) return;
  const int a = (int)(e % W), c = (int)(e / W);
  if (a < stt->rank) K[perm[a] + (size_t)c * W] = Ks[a + (size_t)c * n];
}

// Basic solution of G K = C over the column subset chosen by diagonal pivoting; *rank receives its size.  The stream is
// synchronised (the rank decides the size of the second stage).
// rank_hint > 0: the rank this matrix is expected to have (the previous fit of the same dictionary): only the panels that reach it
// are queued at first - each panel launched behind the one that finds the rank is an empty launch of ~4 us, and so is its update -
// and the rest follow, with the second stage repeated, if the factorisation turns out not to be finished.
// k_host (or nullptr): K_dev is also copied there (k_bytes), in front of the synchronisation; ev_solved (or nullptr): recorded behind
// the last kernel of the solve.
int kp_pivchol_solve_dev(kp_ctx* ctx, const double* G_dev, const double* C_dev, int W, int ncols, double* K_dev, int* rank, int rank_hint, void* k_host,
                         size_t k_bytes, hipEvent_t ev_solved, int k_host_mapped) {
  if (W > 1024 * PC_RPT_MAX) return ctx->fail(KP_ERR_ARG, "rank-revealing solve: W <= 4096");
  hipStream_t s = ctx->stream;
  const int n_max = (W + 15) / 16 * 16, ncp = (ncols + 15) / 16 * 16;
  const size_t bA = (size_t)W * W * 8, bLp = (size_t)n_max * n_max * 8, bCp = (size_t)n_max * ncp * 8, bD = (size_t)(n_max / 16) * 256 * 8;
  char* ws = (char*)ctx->workspace(9, 2 * bA + bLp + bCp + bD + (size_t)W * 16 + 256);
  if (!ws) return ctx->fail(KP_ERR_HIP, "rank-revealing solve: out of device memory");
  double* A = (double*)ws;
  double* L = (double*)(ws + bA);
  double* Lp = (double*)(ws + 2 * bA);
  double* Cp = (double*)(ws + 2 * bA + bLp);
  double* Dinv = (double*)(ws + 2 * bA + bLp + bCp);
  double* dg = (double*)(ws + 2 * bA + bLp + bCp + bD);
  int* perm = (int*)(dg + W);
  int* ipos = perm + W;
  PivState* stt = (PivState*)(((uintptr_t)(ipos + W) + 63) & ~(uintptr_t)63);
  // pivots below W * 64 eps of the first one are rounding noise of a Gram matrix (its entries carry eps * d_1)
  const double rel_tol = (double)W * 64.0 * 2.220446049250313e-16;
  hipLaunchKernelGGL(kp_pivchol_init_kernel, dim3((unsigned)(((int64_t)W * W + 255) / 256)), dim3(256), 0, s, G_dev, W, A, dg, ipos, stt);
  KP_HIP(ctx, hipGetLastError());
  const int nt = W <= 512 ? 512 : 1024, rpt = W <= 1024 ? 1 : PC_RPT_MAX;
  const int nb = std::max(1, std::min(32, (int)((size_t)140 * 1024 / ((size_t)8 * W))));
  const size_t lds = (size_t)nb * W * 8;
  const bool fast32 = nb == 32 && W <= 1024 && !getenv("KP_PIVCHOL_GENERIC");
  static KpLdsCache lds512, lds1024, lds4096, ldsf512, ldsf1024;
  if (fast32) KP_HIP(ctx, nt == 512 ? kp_ensure_lds(ldsf512, (const void*)kp_pivchol_panel32_kernel<512>, lds) : kp_ensure_lds(ldsf1024, (const void*)kp_pivchol_panel32_kernel<1024>, lds));
  KP_HIP(ctx, nt == 512 ? kp_ensure_lds(lds512, (const void*)kp_pivchol_panel_kernel<512, 1>, lds)
              : rpt == 1 ? kp_ensure_lds(lds1024, (const void*)kp_pivchol_panel_kernel<1024, 1>, lds)
                         : kp_ensure_lds(lds4096, (const void*)kp_pivchol_panel_kernel<1024, PC_RPT_MAX>, lds));
  const dim3 ugrid((W + 63) / 64, (W + 63) / 64);
  static const int piv_abl = kp_abl_int("KP_PIV_ABL");      // timing-only ablations (bit 0: no pivot-column load; -DKP_ABLATIONS builds)
  auto run_panels = [&](int k_from, int k_to) -> int {
    for (int k0 = k_from; k0 < k_to; k0 += nb) {
      const int nbk = std::min(nb, W - k0);
      if (fast32 && nt == 512)
        hipLaunchKernelGGL((kp_pivchol_panel32_kernel<512>), dim3(1), dim3(512), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt, piv_abl);
      else if (fast32)
        hipLaunchKernelGGL((kp_pivchol_panel32_kernel<1024>), dim3(1), dim3(1024), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt, piv_abl);
      else if (nt == 512)
        hipLaunchKernelGGL((kp_pivchol_panel_kernel<512, 1>), dim3(1), dim3(512), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt);
      else if (rpt == 1)
        hipLaunchKernelGGL((kp_pivchol_panel_kernel<1024, 1>), dim3(1), dim3(1024), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt);
      else
        hipLaunchKernelGGL((kp_pivchol_panel_kernel<1024, PC_RPT_MAX>), dim3(1), dim3(1024), lds, s, (const double*)A, W, k0, nbk, rel_tol, L, dg, ipos, perm, stt);
      if (k0 + nbk < W) hipLaunchKernelGGL(kp_pivchol_update_kernel, ugrid, dim3(256), 0, s, A, W, k0, nbk, (const double*)L, (const int*)ipos, (const PivState*)stt);
      KP_HIP(ctx, hipGetLastError());
    }
    return KP_OK;
  };
  // the panel in which pivot number rank_hint + 1 is tried is the one that declares the factorisation finished
  int k_done = (rank_hint > 0 && rank_hint < W) ? std::min(W, (rank_hint / nb + 1) * nb) : W;
  {
    int rcp = run_panels(0, k_done);
    if (rcp) return rcp;
  }
  PivState h{};
  PivState* hp = ctx->pin_small ? reinterpret_cast<PivState*>(ctx->pin_small + 4) : &h;     // (page-locked words of the context: direct DMA)
  auto second_stage = [&](int n) -> int {
    // everything behind the factorisation is queued without knowing the rank: the substitution runs at width n - the full
    // (padded) width, or the remembered rank rounded up to 16 (checked afterwards) - with an identity block beyond the rank:
    // ~30 % more substitution work at rank 252 of 336 without a hint, against two host round trips
    KP_HIP(ctx, hipMemsetAsync(K_dev, 0, (size_t)W * ncols * 8, s));
    const int64_t tot = (int64_t)n * n + (int64_t)n * ncp;
    hipLaunchKernelGGL(kp_pivchol_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const double*)L, C_dev, W, ncols, (const int*)perm,
                       (const PivState*)stt, n, ncp, Lp, Cp);
    KP_HIP(ctx, hipGetLastError());
    int rc = kp_factor_substitute_dev(ctx, Lp, n, Cp, ncp, Dinv, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kp_scatter_rows_kernel, dim3((unsigned)(((int64_t)W * ncols + 255) / 256)), dim3(256), 0, s, (const double*)Cp, W, ncols, (const int*)perm,
                       (const PivState*)stt, n, K_dev);
    KP_HIP(ctx, hipGetLastError());
    if (ev_solved) KP_HIP(ctx, hipEventRecord(ev_solved, s));
    if (k_host) {
      int rcc = kp_copy_to_host_async(ctx, K_dev, k_host, k_bytes, k_host_mapped, s);
      if (rcc) return rcc;
    }
    KP_HIP(ctx, hipMemcpyAsync(hp, stt, sizeof(PivState), hipMemcpyDeviceToHost, s));
    KP_HIP(ctx, hipStreamSynchronize(s));
    return KP_OK;
  };
  const int n_hint = (rank_hint > 0 && rank_hint < W) ? std::min(n_max, (rank_hint + 15) / 16 * 16) : n_max;
  {
    int rcs = second_stage(n_hint);
    if (rcs) return rcs;
  }
  if (!hp->done && k_done < W) {                  // the hint was too small: the remaining panels, and the second stage again
    int rcp = run_panels(k_done, W);
    if (!rcp) rcp = second_stage(n_max);
    if (rcp) return rcp;
  } else if (hp->done && hp->rank > n_hint) {     // finished in the hinted panels, but at a rank beyond the hinted width
    int rcs = second_stage(n_max);
    if (rcs) return rcs;
  }
  if (rank) *rank = hp->done ? hp->rank : W;
  return KP_OK;
}

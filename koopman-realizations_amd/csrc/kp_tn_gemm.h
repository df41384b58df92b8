// C (M x N) = beta C + alpha A' B,  A: K x M, B: K x N, all column-major f64 with leading dimensions lda / ldb / ldc - the
// "TN" product, BOTH operands contiguous along the contraction index.  It is the one product every stage of the WIDE
// path needs (dictionaries beyond the LDS-staged Gram kernels and the one-workgroup factorisation, W > 512: the reference's
// fourier dictionary on the arm's six states has 728 functions, Ksysid.m:694-731):
//   * Gram matrices of a lifted snapshot panel, G += Px' Px (upper tiles only, `tri`), C += Px' Py  (Ksysid.m:1114, 1125);
//   * trailing update of the blocked Cholesky factorisation, A22 -= U12' U12 (upper tiles only);
//   * block substitution, C_rest -= U12' Y_k  and  Y_rest -= L21' K_k  (kp_wide.hip).
// Same skeleton as kp_symm_gemm2_kernel (kp_symm_gemm.h): 4-wave workgroup, output tile 16 RA x 16 RB, wave w owns 4 RA rows as
// RA x RB accumulators of v_mfma_f64_4x4x4_4b, contraction in blocks of 16 double-buffered in LDS ([row][k], row stride TNG_RS
// doubles), one barrier per block, operand addresses base + immediate, result tile through LDS so that the stores run along
// the columns of C, XCD-aware workgroup order.  New here: general leading dimensions (64-bit tile bases + 32-bit offsets),
// a contraction RANGE per workgroup (split-K over blockIdx.y into partial buffers, summed in split order by
// kp_tn_gemm_reduce_kernel: bitwise reproducible), the alpha / beta epilogue, the triangular tile skip, and optional weights
// of the contraction index (the Kronecker form of bilinear Gram matrices, kp_wide.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#define TNG_KB 16
#ifndef TNG_RS
#define TNG_RS 22   // = 2 mod 4: the 16 rows x 2 k of a B-operand read (ds_read_b64: 32 lanes over 64 banks) hit 64 different banks; 20 (rounds
#endif              // 4-5) put rows r and r + 8 on the same ones (tools/lds_layout_sim.py model; W = 738 Gram pass 4.47 -> 4.26 ms per 1e5 pairs)

// NW waves per workgroup (4, two workgroups per CU; or 8, one): wave w owns the rows [4 RA w, 4 RA (w + 1)) of the tile
template <int RA, int RB, int NW = 4>
struct TngCfg {
  static constexpr int NT = 64 * NW;
  static constexpr int TM = 4 * RA * NW, TN = 16 * RB;
  static constexpr int PB = 4 * RB / NW;                           // staging loads of B per thread (A: RA)
  static_assert(4 * RB % NW == 0, "tile columns must divide over the staging threads");
  static constexpr int BUF = (TM + TN) * TNG_RS;
  static constexpr int OS = TM + 4;
  static constexpr int ECH = RB < 4 ? RB : 4;
  static constexpr int LDS_DOUBLES = (2 * BUF > 16 * ECH * OS) ? 2 * BUF : 16 * ECH * OS;
  static constexpr size_t LDS_BYTES = (size_t)LDS_DOUBLES * 8;
};

struct TngArgs {
  const double* A; const double* B; double* C;
  double* P;                 // split-K partials [split][N][M] (nsplit > 1)
  int64_t lda, ldb, ldc;
  int M, N, K;
  int kper;                  // contraction range of a split (multiple of TNG_KB)
  int nsplit, nrt, nct, tri;
  int per_xcd;               // tri: active tiles per XCD (launcher)
  double alpha, beta;
  const double *wa, *wb;     // optional row weights (length K each, nullptr = 1): C += alpha sum_k wa[k] wb[k] A[k,:]' B[k,:]
};

// VEC (round 6): the staging moves TWO consecutive contraction indices per thread - one 16-byte global load and one ds_write_b128 where
// the scalar form issues two 8-byte loads and two ds_write_b64 (14 + 14 instructions per thread and contraction block beside the
// 192 MFMAs of a wave; a store instruction costs a wave more than a vector add, measured on the Gram kernel).  Needs A, B 16-byte
// aligned and even leading dimensions (the launcher checks; the panels of the wide path and the blocks of the factorisation are).
template <int RA, int RB, int NW = 4, bool VEC = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void kp_tn_gemm_kernel(TngArgs g) {
  using Cfg = TngCfg<RA, RB, NW>;
  static_assert(!VEC || (RA % 2 == 0 && Cfg::PB % 2 == 0), "VEC: rows of a tile are staged 2 SR per pass");
  constexpr int TM = Cfg::TM, TN = Cfg::TN, BUF = Cfg::BUF, OS = Cfg::OS, RS = TNG_RS, KB = TNG_KB, ECH = Cfg::ECH;
  constexpr int NT = Cfg::NT, SR = NT / 16, PB = Cfg::PB;          // SR rows of a tile are staged per pass of the workgroup
  extern __shared__ __align__(16) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int ct, rt;
  if (!g.tri) {
    ct = (slot / g.nrt) * 8 + xcd;
    rt = slot % g.nrt;
    if (ct >= g.nct) return;
  } else {
    // upper tiles only (a tile strictly below the diagonal is read by nobody): the ACTIVE tiles, listed column by column, are
    // dealt to the XCDs in contiguous runs of g.per_xcd.  (Dealing whole tile columns ct = xcd mod 8 as above left XCD 0 with
    // one tile and XCD 7 with six of a 6 x 8 grid: two rounds of workgroups on one XCD while others idled - the weighted
    // 735-wide products of the Kronecker form took 3.2 ms with half the tiles of the 2.8 ms full products.)
    int t = xcd * g.per_xcd + slot;
    if (slot >= g.per_xcd) return;
    ct = 0;
    for (; ct < g.nct; ++ct) {
      const int cnt = min(g.nrt, (ct * TN + TN - 1) / TM + 1);
      if (t < cnt) break;
      t -= cnt;
    }
    if (ct >= g.nct) return;
    rt = t;
  }
  const int r0 = rt * TM, c0 = ct * TN;
  const int split = blockIdx.y;
  const int k_lo = split * g.kper, k_hi = min(g.K, k_lo + g.kper);
  const int klen = max(0, k_hi - k_lo);

  // staging geometry: thread = (row sr of a pass, contraction index sk); VEC: two indices sk, sk + 1 and twice the rows per pass
  constexpr int KPT = VEC ? 2 : 1, SRV = SR * KPT, RAV = RA / KPT, PBV = PB / KPT;
  const int sk = VEC ? 2 * (tid & 7) : tid & 15, sr = VEC ? tid >> 3 : tid >> 4;
  unsigned go[RAV], xo[PBV > 0 ? PBV : 1];
#pragma unroll
  for (int p = 0; p < RAV; ++p) {
    const int row = sr + SRV * p;
    go[p] = (unsigned)(((int64_t)(r0 + row < g.M ? row : 0) * g.lda + sk) * 8);
  }
#pragma unroll
  for (int p = 0; p < PBV; ++p) {
    const int col = sr + SRV * p;
    xo[p] = (unsigned)(((int64_t)(c0 + col < g.N ? col : 0) * g.ldb + sk) * 8);
  }
  const char* Ab = (const char*)(g.A + (int64_t)r0 * g.lda + k_lo);
  const char* Bb = (const char*)(g.B + (int64_t)c0 * g.ldb + k_lo);
  const int so = sr * RS + sk;
  double sg[RA], sx[PB > 0 ? PB : 1], swa = 1.0, swb = 1.0, swa1 = 1.0, swb1 = 1.0;
  const int nkb = (klen + KB - 1) / KB, nkb_full = klen / KB;
  const bool weighted = g.wa != nullptr || g.wb != nullptr;
  auto stage_load = [&](int kb) {
    if (kb < nkb_full) {
      const unsigned ko = (unsigned)kb * (KB * 8u);
      if constexpr (VEC) {
#pragma unroll
        for (int p = 0; p < RAV; ++p) { const double2 v = *(const double2*)(Ab + (go[p] + ko)); sg[2 * p] = v.x; sg[2 * p + 1] = v.y; }
#pragma unroll
        for (int p = 0; p < PBV; ++p) { const double2 v = *(const double2*)(Bb + (xo[p] + ko)); sx[2 * p] = v.x; sx[2 * p + 1] = v.y; }
      } else {
#pragma unroll
        for (int p = 0; p < RA; ++p) sg[p] = *(const double*)(Ab + (go[p] + ko));
#pragma unroll
        for (int p = 0; p < PB; ++p) sx[p] = *(const double*)(Bb + (xo[p] + ko));
      }
      if (weighted) {                     // (uniform) the weight of contraction index k rides on the A operand; it is applied
        const int k = k_lo + kb * KB + sk;     // in stage_store: a product here would wait for the loads before the MFMAs they hide behind
        swa = g.wa ? g.wa[k] : 1.0;
        swb = g.wb ? g.wb[k] : 1.0;
        if (VEC) { swa1 = g.wa ? g.wa[k + 1] : 1.0; swb1 = g.wb ? g.wb[k + 1] : 1.0; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < KPT; ++q) {     // the range's last, partial block: element by element
        const int k = kb * KB + sk + q;
        const bool kok = k < klen;
        const unsigned ko = (unsigned)((kok ? k : klen - 1) - sk) * 8u;
#pragma unroll
        for (int p = 0; p < RAV; ++p) { const double v = *(const double*)(Ab + (go[p] + ko)); sg[KPT * p + q] = kok ? v : 0.0; }
#pragma unroll
        for (int p = 0; p < PBV; ++p) { const double v = *(const double*)(Bb + (xo[p] + ko)); sx[KPT * p + q] = kok ? v : 0.0; }
        if (weighted) {
          const int kc = k_lo + (kok ? k : klen - 1);
          (q ? swa1 : swa) = g.wa ? g.wa[kc] : 1.0;
          (q ? swb1 : swb) = g.wb ? g.wb[kc] : 1.0;
        }
      }
    }
  };
  auto stage_store = [&](int buf) {
    double* d = sm + buf * BUF + so;
    if (weighted) {
      const double w = swa * swb, w1 = swa1 * swb1;
#pragma unroll
      for (int p = 0; p < RAV; ++p) {
        sg[KPT * p] *= w;
        if (VEC) sg[2 * p + 1] *= w1;
      }
    }
    if constexpr (VEC) {
#pragma unroll
      for (int p = 0; p < RAV; ++p) *reinterpret_cast<double2*>(__builtin_assume_aligned(d + SRV * p * RS, 16)) = make_double2(sg[2 * p], sg[2 * p + 1]);
#pragma unroll
      for (int p = 0; p < PBV; ++p) *reinterpret_cast<double2*>(__builtin_assume_aligned(d + (TM + SRV * p) * RS, 16)) = make_double2(sx[2 * p], sx[2 * p + 1]);
    } else {
#pragma unroll
      for (int p = 0; p < RA; ++p) d[SR * p * RS] = sg[p];
#pragma unroll
      for (int p = 0; p < PB; ++p) d[(TM + SR * p) * RS] = sx[p];
    }
  };

  const int lc = lane & 3, blk = (lane >> 2) & 3, lk = lane >> 4;
  const int ab = (wave * 4 * RA + lc) * RS + lk;
  const int bb = (TM + 4 * blk + lc) * RS + lk;
  double acc[RA][RB];
#pragma unroll
  for (int ra = 0; ra < RA; ++ra)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[ra][rb] = 0.0;

  auto compute = [&](const double* s) {
    double a[2][RA], b[2][RB];
#pragma unroll
    for (int ra = 0; ra < RA; ++ra) a[0][ra] = s[ab + 4 * ra * RS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) b[0][rb] = s[bb + 16 * rb * RS];
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < KB / 4) {
#pragma unroll
        for (int ra = 0; ra < RA; ++ra) a[nxt][ra] = s[ab + 4 * ra * RS + 4 * (kk + 1)];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) b[nxt][rb] = s[bb + 16 * rb * RS + 4 * (kk + 1)];
      }
#pragma unroll
      for (int ra = 0; ra < RA; ++ra)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[ra][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[cur][ra], b[cur][rb], acc[ra][rb], 0, 0, 0);
    }
  };

  if (nkb > 0) {
    stage_load(0);
    stage_store(0);
  }
  __syncthreads();
  for (int kb = 0; kb < nkb; kb += 2) {
    if (kb + 1 < nkb) stage_load(kb + 1);
    compute(sm);
    if (kb + 1 < nkb) stage_store(1);
    __syncthreads();
    if (kb + 1 >= nkb) break;
    if (kb + 2 < nkb) stage_load(kb + 2);
    compute(sm + BUF);
    if (kb + 2 < nkb) stage_store(0);
    __syncthreads();
  }
  // result tile through LDS, 16 ECH columns at a time
  double* Pp = g.nsplit > 1 ? g.P + (size_t)split * g.M * g.N : nullptr;
#pragma unroll
  for (int ch = 0; ch < RB; ch += ECH) {
    if (ch) __syncthreads();
#pragma unroll
    for (int ra = 0; ra < RA; ++ra)
#pragma unroll
      for (int rb = 0; rb < ECH; ++rb)
        if (ch + rb < RB) sm[(16 * rb + 4 * blk + lc) * OS + wave * 4 * RA + 4 * ra + lk] = acc[ra][ch + rb];
    __syncthreads();
    for (int e = tid; e < TM * 16 * ECH; e += NT) {
      const int j = e / TM, i = e - j * TM, jc = c0 + 16 * ch + j;
      if (r0 + i < g.M && jc < g.N && 16 * ch + j < TN) {
        const double v = sm[j * OS + i];
        if (Pp) Pp[(size_t)jc * g.M + r0 + i] = v;
        else {
          double* dst = g.C + (int64_t)jc * g.ldc + r0 + i;
          *dst = g.beta != 0.0 ? g.beta * *dst + g.alpha * v : g.alpha * v;
        }
      }
    }
  }
}

// C = beta C + alpha sum_s P[s], splits in order; the tiles the product skipped (`tri`) are skipped here too
static __global__ __launch_bounds__(256) void kp_tn_gemm_reduce_kernel(const double* __restrict__ P, int nsplit, int M, int N, double* __restrict__ C, int64_t ldc,
                                                                double alpha, double beta, int tri, int TM, int TN) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  const int i = (int)(e % M), j = (int)(e / M);
  if (tri && (i / TM) * TM > (j / TN) * TN + TN - 1) return;
  double s = 0.0;
  for (int p = 0; p < nsplit; ++p) s += P[(size_t)p * M * N + e];
  double* dst = C + (int64_t)j * ldc + i;
  *dst = beta != 0.0 ? beta * *dst + alpha * s : alpha * s;
}

// lower triangle := transpose of the upper one (n x n, leading dimension ld)
static __global__ __launch_bounds__(256) void kp_mirror_upper_kernel(double* __restrict__ A, int n, int64_t ld) {
  __shared__ double T[16][17];
  const int bi = blockIdx.x, bj = blockIdx.y;          // block (rows bi, columns bj) of the UPPER triangle: bi <= bj
  if (bi > bj) return;
  const int ti = threadIdx.x & 15, tj = threadIdx.x >> 4;
  const int i = 16 * bi + ti, j = 16 * bj + tj;
  T[tj][ti] = (i < n && j < n) ? A[(int64_t)j * ld + i] : 0.0;
  __syncthreads();
  // element (row 16 bj + ti, column 16 bi + tj) of the lower triangle = upper (16 bi + tj, 16 bj + ti)
  const int r = 16 * bj + ti, c = 16 * bi + tj;
  if (r < n && c < n && r > c) A[(int64_t)c * ld + r] = T[ti][tj];
}

// output tiles (tm x tn) that a product runs: all of them, or those that meet the upper triangle
static inline int64_t tng_count_tiles(int M, int N, int tm, int tn, int tri) {
  const int nrt = (M + tm - 1) / tm, nct = (N + tn - 1) / tn;
  if (!tri) return (int64_t)nrt * nct;
  int64_t cnt = 0;
  for (int r = 0; r < nrt; ++r)
    for (int c = 0; c < nct; ++c) cnt += r * tm <= c * tn + tn - 1 ? 1 : 0;
  return cnt;
}
template <int RA, int RB, int NW = 4, bool VEC = false>
static hipError_t tng_launch_cfg(hipStream_t st, TngArgs g) {
  using Cfg = TngCfg<RA, RB, NW>;
  if constexpr (!VEC && RA % 2 == 0 && Cfg::PB % 2 == 0) {
    static const bool no_vec = getenv("KP_TNG_NOVEC") != nullptr;       // (A/B measurements)
    if (!no_vec && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && g.lda % 2 == 0 && g.ldb % 2 == 0)
      return tng_launch_cfg<RA, RB, NW, true>(st, g);
  }
  static bool attr_set[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 32 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kp_tn_gemm_kernel<RA, RB, NW, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32) attr_set[dev] = true;
  }
  g.nrt = (g.M + Cfg::TM - 1) / Cfg::TM;
  g.nct = (g.N + Cfg::TN - 1) / Cfg::TN;
  int nblk = 8 * ((g.nct + 7) / 8) * g.nrt;
  g.per_xcd = 0;
  if (g.tri) {
    const int64_t act = tng_count_tiles(g.M, g.N, Cfg::TM, Cfg::TN, 1);
    g.per_xcd = (int)((act + 7) / 8);
    nblk = 8 * g.per_xcd;
  }
  hipLaunchKernelGGL((kp_tn_gemm_kernel<RA, RB, NW, VEC>), dim3(nblk, g.nsplit), dim3(Cfg::NT), Cfg::LDS_BYTES, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || g.nsplit <= 1) return e;
  hipLaunchKernelGGL(kp_tn_gemm_reduce_kernel, dim3((unsigned)(((int64_t)g.M * g.N + 255) / 256)), dim3(256), 0, st, g.P, g.nsplit, g.M, g.N, g.C, g.ldc,
                     g.alpha, g.beta, g.tri, Cfg::TM, Cfg::TN);
  return hipGetLastError();
}

// How many workgroups a product is cut into decides how well it fills the chip: its workgroups all take the same time, so they
// run in rounds of `slots` (2 per CU) and a product of 1.1 rounds takes as long as one of 2.  (The dense bilinear W = 2 940
// products ran 370 and 713 workgroups - 0.72 and 1.39 rounds - and the 735-wide weighted products of the Kronecker form 546
// and 576; measured: the weighted form, 62.5 % of the flops, took 96 % of the time.)  The planner counts the tiles that will
// actually run, for 96- and 64-column tiles, and picks the contraction split whose last round is fullest; wider tiles and
// fewer splits win ties (192 instead of 128 MFMAs per wave between two barriers; fewer partial sums to write and add).
// (M x N in tm x tn tiles, ns splits.)  The workgroups of a launch are dealt to the 8 XCDs round-robin and every XCD has its
// own slots / 8: what counts is the fullest XCD - full products give XCD x the tile columns x, x + 8, ...; triangular ones
// ceil(active / 8) tiles each (kp_tn_gemm_kernel).  30 active tiles x 17 splits are 510 workgroups on 512 slots and still
// TWO rounds: 4 x 17 = 68 on the 64 slots of seven XCDs (measured: as long as the full product's 1 008).
// The forms of the 128-row-and-up product: tile, workgroups per CU, relative speed of a FULL round (measured).  The two
// eight-wave forms (ONE workgroup per CU, 256 x 96 and 192 x 128: 1.3 - 1.4 x fewer operand bytes from L2 per flop) are built
// and OFF: all eight waves meet at one barrier per contraction block, and the bilinear W = 2 940 Gram pass took 59 - 60 ms
// with either against 42.6 ms with two independent 128 x 96 workgroups per CU (KP_TNG_FORCE=2 / 3; KP_TNG_BIG=w enables them
// with weight w) - the same lock-step loss as every one-workgroup-per-CU form measured in rounds 2 - 4.
struct TngShape { int tm, tn, wg_per_cu; double w; };
#define TNG_NSHAPES 4
static inline const TngShape* tng_shapes() {
  static const double big = [] { const char* e = getenv("KP_TNG_BIG"); return e ? atof(e) : 0.0; }();
  static const TngShape sh[TNG_NSHAPES] = {{128, 64, 2, 0.9}, {128, 96, 2, 1.0}, {256, 96, 1, big}, {192, 128, 1, big}};
  return sh;
}
static inline bool tng_shape_on(int i) {
  static const int wide_cols = [] { const char* e = getenv("KP_TNG_RB"); return e ? atoi(e) : 6; }();      // KP_TNG_RB=4: 64 columns only
  static const int force = [] { const char* e = getenv("KP_TNG_FORCE"); return e ? atoi(e) : -1; }();       // one form only (measurements)
  if (force >= 0) return i == force;
  return i == 0 || (wide_cols == 6 && (i == 1 || tng_shapes()[i].w > 0.0));
}
static inline double tng_score(int M, int N, const TngShape& sh, int tri, int ns, int slots) {
  const int64_t tiles = tng_count_tiles(M, N, sh.tm, sh.tn, tri);
  const int nrt = (M + sh.tm - 1) / sh.tm, nct = (N + sh.tn - 1) / sh.tn;
  const int64_t per_xcd = tri ? (tiles + 7) / 8 : (int64_t)nrt * ((nct + 7) / 8);
  const int64_t xs = std::max(1, slots * sh.wg_per_cu / 16), rounds = (per_xcd * ns + xs - 1) / xs;
  // work done per round-slot, in units of the 128 x 96 tile's
  const double fill = (double)(tiles * ns) / (double)(rounds * xs * 8);
  return fill * sh.w * (1.0 - 0.004 * (ns - 1));
}
// Form (index into tng_shapes) of a product that runs with `nsplit` contraction splits.
static inline int tng_pick_shape(int M, int N, int nsplit, int tri, int slots = 512) {
  const int ns = nsplit > 1 ? nsplit : 1;
  int best = 0;
  double bs = -1.0;
  for (int i = 0; i < TNG_NSHAPES; ++i) {
    if (!tng_shape_on(i)) continue;
    const double sc = tng_score(M, N, tng_shapes()[i], tri, ns, slots);
    if (sc > bs) { bs = sc; best = i; }
  }
  return best;
}
// Number of contraction splits: `slots` workgroup slots (2 per CU); at least 16 contraction blocks per split, at most 64 splits.
static inline int tng_pick_splits(int M, int N, int K, int tri, int slots) {
  int max_ns = K / (16 * TNG_KB);
  max_ns = max_ns < 1 ? 1 : max_ns > 64 ? 64 : max_ns;
  int best = 1;
  double bs = -1.0;
  if (M <= 64) {                                            // the 64 x 64 form
    const TngShape small{64, 64, 2, 1.0};
    for (int ns = 1; ns <= max_ns; ++ns) {
      const double sc = tng_score(M, N, small, tri, ns, slots);
      if (sc > bs) { bs = sc; best = ns; }
    }
    return best;
  }
  for (int ns = 1; ns <= max_ns; ++ns)
    for (int i = 0; i < TNG_NSHAPES; ++i) {
      if (!tng_shape_on(i)) continue;
      const double sc = tng_score(M, N, tng_shapes()[i], tri, ns, slots);
      if (sc > bs) { bs = sc; best = ns; }
    }
  return best;
}

// P (or nullptr): room for nsplit * M * N doubles when nsplit > 1.
// wa, wb (or nullptr): row weights, see TngArgs.
static inline hipError_t kp_tn_gemm(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb, int M, int N, int K, double* C, int64_t ldc,
                                    double alpha, double beta, int tri, int nsplit, double* P, const double* wa = nullptr, const double* wb = nullptr) {
  if (M <= 0 || N <= 0) return hipSuccess;
  // tile rows are addressed by 32-bit byte offsets from the tile's base
  if ((uint64_t)256 * (uint64_t)(lda > ldb ? lda : ldb) * 8u + (uint64_t)K * 8u >= (1ull << 32)) return hipErrorInvalidValue;
  TngArgs g;
  g.A = A; g.B = B; g.C = C; g.P = P;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.nsplit = (nsplit > 1 && P) ? nsplit : 1;
  g.kper = ((K + g.nsplit - 1) / g.nsplit + TNG_KB - 1) / TNG_KB * TNG_KB;
  if (g.kper < TNG_KB) g.kper = TNG_KB;
  g.nsplit = g.nsplit > 1 ? (K + g.kper - 1) / g.kper : 1;
  if (g.nsplit < 1) g.nsplit = 1;
  g.nrt = g.nct = 0;
  g.tri = tri;
  g.alpha = alpha; g.beta = beta;
  g.wa = wa; g.wb = wb;
  if (M <= 64) return tng_launch_cfg<4, 4>(st, g);
  switch (tng_pick_shape(M, N, g.nsplit, tri)) {
    case 3: return tng_launch_cfg<6, 8, 8>(st, g);
    case 2: return tng_launch_cfg<8, 6, 8>(st, g);
    case 1: return tng_launch_cfg<8, 6>(st, g);
    default: return tng_launch_cfg<8, 4>(st, g);
  }
}

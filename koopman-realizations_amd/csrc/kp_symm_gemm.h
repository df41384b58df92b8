// C (W x nc) = G (W x W, symmetric) * X (W x nc), all column-major f64: the product of every FISTA iteration of the lasso
// QP (Ksysid.solve_KoopmanQP, Ksysid.m:1095-1176; X = the K matrices of all running lasso values side by side) and of the
// optimality check of its active-set polish.
//
// G symmetric => C[i][j] = sum_k G[k + i W] X[k + j W]: BOTH operands are contiguous along the contraction index, so the
// tiles go from global memory to LDS without a transpose ([row][k], k fastest, row stride SG2_RS doubles).
//
// Workgroup = 4 waves, output tile (16 RA) rows x 64 columns; wave w owns the 4 RA rows [4 RA w, 4 RA (w + 1)) of the tile
// and all 64 columns as RA x 4 accumulators of v_mfma_f64_4x4x4_4b_f64 (A: one 4-row group, the same in the four blocks of
// the instruction; B: 16 columns, 4 per block): per 4 k-steps RA + 4 ds_read_b64 feed 4 RA MFMAs (W = 336: RA = 7, 11 reads
// per 28 MFMAs, 3 x 221 workgroups).  The contraction runs in blocks of 16, double-buffered in LDS: the global loads of block
// b + 1 are issued before the MFMAs of block b and stored after them, one barrier per block; inside a block every operand
// address is `base + immediate` (no VALU between the MFMAs: nothing on the vector ALU overlaps the f64 matrix pipe on gfx950,
// DESIGN 3.1).  The result tile leaves through LDS so that the stores run along the columns of C.  Workgroup order: the row
// tiles of one column tile sit on one XCD (blocks b, b + 8, ... share an L2), so X comes from HBM once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

// kernel choice by output size in (16 rows x 1 column) units, measured with tools/symm_gemm_probe at W = 336 / 136 / 60:
// up to SG2_SMALL_UNITS the stage-everything kernel wins (latency); above, the tiled kernel with 32 / 48 / 64 / 96 columns
// per workgroup, whichever fills the rounds of workgroups best (kp_symm_gemm2)
#define SG2_SMALL_UNITS 19000
#define SG2_KB 16        // contraction block
#define SG2_RS 22        // LDS row stride (doubles): = 2 mod 4 -> the 16 rows x 2 k of a B-operand read hit 64 different banks (20: rows r, r + 8 collide; kp_tn_gemm.h)

template <int RA, int RB>
struct Sg2Cfg {
  static constexpr int TM = 16 * RA;                            // output rows per workgroup
  static constexpr int TN = 16 * RB;                            // output columns per workgroup
  static constexpr int BUF = (TM + TN) * SG2_RS;                // doubles per stage buffer: G rows, then X rows
  static constexpr int OS = TM + 4;                             // column stride of the result tile in LDS
  static constexpr int ECH = RB < 4 ? RB : 4;                   // 16-column groups per chunk of the result tile
  static constexpr int LDS_DOUBLES = (2 * BUF > 16 * ECH * OS) ? 2 * BUF : 16 * ECH * OS;
  static constexpr size_t LDS_BYTES = (size_t)LDS_DOUBLES * 8;
};

#ifndef SG2_ABL
#define SG2_ABL 0        // timing-only ablations of tools/symm_gemm_probe (1: no global loads, 2: no stage stores, 3: no barriers)
#endif

template <int RA, int RB>
__global__ __launch_bounds__(256, 2) void kp_symm_gemm2_kernel(const double* __restrict__ G, const double* __restrict__ X, int W, int nc,
                                                               double* __restrict__ C, int nrt, int nct) {
  using Cfg = Sg2Cfg<RA, RB>;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, BUF = Cfg::BUF, OS = Cfg::OS, RS = SG2_RS, KB = SG2_KB, ECH = Cfg::ECH;
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: linear id -> (xcd, slot); the nrt row tiles of a column tile are consecutive slots of one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = (slot / nrt) * 8 + xcd, rt = slot % nrt;
  if (ct >= nct) return;
  const int r0 = rt * TM, c0 = ct * TN;

  // ---- staging: thread = (k = tid & 15, row = tid >> 4 (+16 per pass)); RA passes over the G tile, RB over the X tile.
  // Rows of G past W and columns of X past nc only reach output rows / columns that are never stored, so their loads are
  // merely redirected into valid memory; what must be exact is the contraction range: k >= W contributes zero (only in a
  // last, partial block: W = 336 has none).  Addresses = uniform base + 32-bit byte offset (one VGPR per row).
  const int sk = tid & 15, sr = tid >> 4;
  unsigned go[RA], xo[RB];
#pragma unroll
  for (int p = 0; p < RA; ++p) {
    const int row = r0 + sr + 16 * p;
    go[p] = ((unsigned)(row < W ? row : 0) * (unsigned)W + sk) * 8u;
  }
#pragma unroll
  for (int p = 0; p < RB; ++p) {
    const int col = c0 + sr + 16 * p;
    xo[p] = ((unsigned)(col < nc ? col : 0) * (unsigned)W + sk) * 8u;
  }
  const char* Gb = (const char*)G;
  const char* Xb = (const char*)X;
  const int so = sr * RS + sk;                         // + 16 p RS (G rows), + (TM + 16 p) RS (X rows)
  double sg[RA], sx[RB];
  const int nkb = (W + KB - 1) / KB, nkb_full = W / KB;
  auto stage_load = [&](int kb) {
    if (kb < nkb_full) {
      const unsigned ko = (unsigned)kb * (KB * 8u);
#pragma unroll
      for (int p = 0; p < RA; ++p) sg[p] = *(const double*)(Gb + (go[p] + ko));
#pragma unroll
      for (int p = 0; p < RB; ++p) sx[p] = *(const double*)(Xb + (xo[p] + ko));
    } else {
      const int k = kb * KB + sk;
      const bool kok = k < W;
      const unsigned ko = (unsigned)((kok ? k : W - 1) - sk) * 8u;
#pragma unroll
      for (int p = 0; p < RA; ++p) { const double v = *(const double*)(Gb + (go[p] + ko)); sg[p] = kok ? v : 0.0; }
#pragma unroll
      for (int p = 0; p < RB; ++p) { const double v = *(const double*)(Xb + (xo[p] + ko)); sx[p] = kok ? v : 0.0; }
    }
  };
  auto stage_store = [&](int buf) {
    double* d = sm + buf * BUF + so;
#pragma unroll
    for (int p = 0; p < RA; ++p) d[16 * p * RS] = sg[p];
#pragma unroll
    for (int p = 0; p < RB; ++p) d[(TM + 16 * p) * RS] = sx[p];
  };

  // ---- MFMA operands: A[i = lane & 3][k = lane >> 4] (same in all four blocks), B[k = lane >> 4][j = 4 blk + (lane & 3)]
  const int lc = lane & 3, blk = (lane >> 2) & 3, lk = lane >> 4;
  const int ab = (wave * 4 * RA + lc) * RS + lk;       // + 4 ra RS + 4 kk
  const int bb = (TM + 4 * blk + lc) * RS + lk;        // + 16 rb RS + 4 kk
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

  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (int kb = 0; kb < nkb; kb += 2) {                // unrolled by two: the buffer offsets are immediates
    if (SG2_ABL < 1 && kb + 1 < nkb) stage_load(kb + 1);
    compute(sm);
    if (SG2_ABL < 2 && kb + 1 < nkb) stage_store(1);
    if (SG2_ABL < 3) __syncthreads();
    if (kb + 1 >= nkb) break;
    if (SG2_ABL < 1 && kb + 2 < nkb) stage_load(kb + 2);
    compute(sm + BUF);
    if (SG2_ABL < 2 && kb + 2 < nkb) stage_store(0);
    if (SG2_ABL < 3) __syncthreads();
  }
  // ---- result tile through LDS, 64 columns at a time (every wave has passed the barrier behind its last operand read).
  // D lane: row = lane >> 4 of the 4-row group, column = 4 blk + (lane & 3) of the 16-column group
#pragma unroll
  for (int ch = 0; ch < RB; ch += ECH) {
    if (ch) __syncthreads();
#pragma unroll
    for (int ra = 0; ra < RA; ++ra)
#pragma unroll
      for (int rb = 0; rb < ECH; ++rb)
        if (ch + rb < RB) sm[(16 * rb + 4 * blk + lc) * OS + wave * 4 * RA + 4 * ra + lk] = acc[ra][ch + rb];
    __syncthreads();
    for (int e = tid; e < TM * 16 * ECH; e += 256) {
      const int j = e / TM, i = e - j * TM, jc = c0 + 16 * ch + j;
      if (r0 + i < W && jc < nc && 16 * ch + j < TN) C[(size_t)jc * W + r0 + i] = sm[j * OS + i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Few columns (a handful of lasso values still running): latency, not throughput.  Workgroup: 16 output rows x 32 output
// columns; the whole contraction range of both operands is staged in LDS at once ([r][16 G cols | 32 X cols], every load
// of the workgroup in flight together), then 8 waves x (one 4-row group, half of the k-steps) x 2 quads run without
// barriers.  (Round 2's only product kernel: 2.7 flop per byte pulled from L2.)
// ------------------------------------------------------------------------------------------------
#define SG_RS 49   // odd: the staging writes (lanes = consecutive rows) and the operand reads both spread over the banks
__global__ __launch_bounds__(512) void kp_symm_gemm_kernel(const double* __restrict__ G, const double* __restrict__ X, int W, int nc,
                                                           double* __restrict__ C) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0c = blockIdx.x * 16;       // output rows = columns of G'
  const int c0 = blockIdx.y * 32;        // output columns
  const int Wp = (W + 3) & ~3;
  // consecutive threads: consecutive rows of one column (coalesced); all 48 columns (loads) of a row in flight
  for (int r = tid; r < Wp; r += 512) {
    double v[48];
#pragma unroll
    for (int col = 0; col < 48; ++col) {
      const int gc = col < 16 ? r0c + col : c0 + col - 16;
      const bool ok = r < W && (col < 16 ? gc < W : gc < nc);
      const double* src = col < 16 ? G : X;
      v[col] = ok ? src[r + (size_t)gc * W] : 0.0;
    }
#pragma unroll
    for (int col = 0; col < 48; ++col) sm[r * SG_RS + col] = v[col];
  }
  __syncthreads();
  // 8 waves: A group = wave & 3 (4 output rows), half of the contraction range = wave >> 2
  const int lrow = (lane >> 4) * SG_RS, blk = (lane >> 2) & 3, lc = lane & 3;
  const int ao = lrow + 4 * (wave & 3) + lc;
  const int bo0 = lrow + 16 + 4 * blk + lc, bo1 = bo0 + 16;
  double acc0 = 0.0, acc1 = 0.0;
  const int nk = Wp / 4, kh = (nk + 1) / 2;
  const int k0 = (wave >> 2) * kh, k1 = min(nk, k0 + kh);
#pragma unroll 6
  for (int k = k0; k < k1; ++k) {
    const double a = sm[k * 4 * SG_RS + ao];
    const double b0 = sm[k * 4 * SG_RS + bo0], b1 = sm[k * 4 * SG_RS + bo1];
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc1, 0, 0, 0);
  }
  __syncthreads();                       // operands are consumed: reuse the LDS for the split-k partials
  if (wave >= 4) { sm[(wave - 4) * 128 + lane] = acc0; sm[(wave - 4) * 128 + 64 + lane] = acc1; }
  __syncthreads();
  if (wave < 4) {
    acc0 += sm[wave * 128 + lane];
    acc1 += sm[wave * 128 + 64 + lane];
    // D lane: column j = lane & 3 of block (lane >> 2) & 3, row i = lane >> 4
    const int io = r0c + 4 * wave + (lane >> 4);
    const int j0 = c0 + 4 * blk + lc, j1 = j0 + 16;
    if (io < W) {
      if (j0 < nc) C[io + (size_t)j0 * W] = acc0;
      if (j1 < nc) C[io + (size_t)j1 * W] = acc1;
    }
  }
}

// fallback for W too large for either LDS staging: one thread per output element
__global__ __launch_bounds__(256) void kp_symm_gemm_naive_kernel(const double* __restrict__ G, const double* __restrict__ X, int W, int nc,
                                                                 double* __restrict__ C) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)W * nc) return;
  const int i = (int)(e % W), j = (int)(e / W);
  double s = 0.0;
  for (int k = 0; k < W; ++k) s += G[i + (size_t)k * W] * X[k + (size_t)j * W];
  C[i + (size_t)j * W] = s;
}

// rows per workgroup = 16 RA: the RA in 4..8 with the least padded work (ties: the larger tile)
static inline int sg2_pick_ra(int W) {
  const int wg = (W + 3) / 4;
  int best = 8, best_pad = 1 << 30;
  for (int ra = 8; ra >= 4; --ra) {
    const int tiles = (wg + 4 * ra - 1) / (4 * ra), pad = tiles * 4 * ra;
    if (pad < best_pad) { best_pad = pad; best = ra; }
  }
  return best;
}

template <int RA, int RB>
static hipError_t sg2_launch(hipStream_t st, const double* G, const double* X, int W, int nc, double* C) {
  using Cfg = Sg2Cfg<RA, RB>;
  static bool attr_set[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 32 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kp_symm_gemm2_kernel<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32) attr_set[dev] = true;
  }
  const int nrt = (W + Cfg::TM - 1) / Cfg::TM, nct = (nc + Cfg::TN - 1) / Cfg::TN;
  const int nblk = 8 * ((nct + 7) / 8) * nrt;
  hipLaunchKernelGGL((kp_symm_gemm2_kernel<RA, RB>), dim3(nblk), dim3(256), Cfg::LDS_BYTES, st, G, X, W, nc, C, nrt, nct);
  return hipGetLastError();
}

template <int RB>
static hipError_t sg2_launch_rb(hipStream_t st, const double* G, const double* X, int W, int nc, double* C) {
  switch (sg2_pick_ra(W)) {
    case 4: return sg2_launch<4, RB>(st, G, X, W, nc, C);
    case 5: return sg2_launch<5, RB>(st, G, X, W, nc, C);
    case 6: return sg2_launch<6, RB>(st, G, X, W, nc, C);
    case 7: return sg2_launch<7, RB>(st, G, X, W, nc, C);
    default: return sg2_launch<8, RB>(st, G, X, W, nc, C);
  }
}

static inline hipError_t kp_symm_gemm_small(hipStream_t st, const double* G, const double* X, int W, int nc, double* C) {
  const size_t lds = (size_t)((W + 3) & ~3) * SG_RS * 8;
  static bool attr_set[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 156 * 1024) {
    hipLaunchKernelGGL(kp_symm_gemm_naive_kernel, dim3((unsigned)(((int64_t)W * nc + 255) / 256)), dim3(256), 0, st, G, X, W, nc, C);
    return hipGetLastError();
  }
  if (dev < 0 || dev >= 32 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kp_symm_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kp_symm_gemm_kernel, dim3((W + 15) / 16, (nc + 31) / 32), dim3(512), lds, st, G, X, W, nc, C);
  return hipGetLastError();
}

// variant: 0 = by shape; 1 = small-column kernel; 2 / 3 = tiled kernel with 32 / 64 columns per workgroup; 10 + RB = 16 RB columns
static inline hipError_t kp_symm_gemm2(hipStream_t st, const double* G, const double* X, int W, int nc, double* C, int variant = 0) {
  if (W <= 0 || nc <= 0) return hipSuccess;
  // the tiled kernel addresses X with 32-bit byte offsets from a uniform base (one VGPR per staged row)
  if ((uint64_t)nc * (uint64_t)W * 8u >= (1ull << 32)) variant = 1;
  if (variant == 0) {
    const int64_t cols_rows = (int64_t)nc * ((W + 15) / 16);      // 16-row x 1-column units of output
    if (W < 48 || cols_rows <= SG2_SMALL_UNITS) variant = 1;
    else {
      // Columns per workgroup (16 RB) by a two-line cost model fitted to tools/symm_gemm_probe at W = 336 (RA = 7, 21 blocks):
      // a workgroup takes  nkb (a + b RA RB)  and the launch  floor(r) + {0 | 0.55 | 1}  of those, r = workgroups / (2 per
      // CU) - a last round that fills at most half of the slots runs one workgroup per CU and is that much shorter.  (The
      // fixed 32 / 64 columns of before ran 663 workgroups on 512 slots at 14 112 columns, 552 at 11 760: 0.094 -> 0.090 ms
      // and 0.092 -> 0.071 ms with the widths chosen here.)
      static int slots = 0;
      if (!slots) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
      }
      const int ra = sg2_pick_ra(W), nrt = (W + 16 * ra - 1) / (16 * ra), nkb = (W + SG2_KB - 1) / SG2_KB;
      double best = 1e300;
      for (int rb : {2, 3, 4, 6}) {
        const double t_wg = nkb * (6.4e-4 + 8.6e-5 * ra * rb);
        const double r = (double)nrt * ((nc + 16 * rb - 1) / (16 * rb)) / slots;
        const double fl = floor(r), fr = r - fl;
        const double rounds = r <= 1.0 ? 1.0 : fl + (fr < 1e-9 ? 0.0 : fr <= 0.5 ? 0.55 : 1.0);
        if (t_wg * rounds < best) { best = t_wg * rounds; variant = 10 + rb; }
      }
    }
  }
  switch (variant) {
    case 1: return kp_symm_gemm_small(st, G, X, W, nc, C);
    case 2: case 12: return sg2_launch_rb<2>(st, G, X, W, nc, C);
    case 13: return sg2_launch_rb<3>(st, G, X, W, nc, C);
    case 15: return sg2_launch_rb<5>(st, G, X, W, nc, C);
    case 16: return sg2_launch_rb<6>(st, G, X, W, nc, C);
    default: return sg2_launch_rb<4>(st, G, X, W, nc, C);
  }
}

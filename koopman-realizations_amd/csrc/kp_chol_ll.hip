// Left-looking blocked Cholesky of one (padded) n x n Gram matrix by ONE workgroup, n <= 352 (the config shapes:
// W = 336, 200, 136), on the matrix pipe.  K = Px \ Py (Ksysid.m:1069) through the normal equations: this is the
// factorisation between the fused Gram kernel and the block substitution (kp_trsm2_kernel).
//
// Why left-looking.  One CU reads L2 at 25-35 B/clk (tools/cu_bw_probe.hip), a round trip is 200-500 cycles
// (tools/lat_probe.hip).  A right-looking sweep rewrites the whole trailing matrix once per 16-column panel: 12.6 MB
// through that path at n = 336, three times the time of its flops.  Here panel k gathers all earlier updates in one
// product,
//     S = A(k0:n, k0:k0+16) - L(k0:n, 0:k0) L(k0:k0+16, 0:k0)',
// reading every finished column of L once per panel (3.2 MB in total) and keeping S in accumulators (v_mfma_f64_4x4x4:
// the 4 blocks are the 4 row groups of a 16-row tile, B = 4 panel columns shared by the blocks).
//
// Roles and overlap - the factorisation of a 16 x 16 diagonal block is ~5000 serial cycles of one wave, a third of the
// kernel if nothing runs beside it:
//   wave 0      panel wave: factors the diagonal block of panel k in registers (lane r of every 16-lane row owns row r; pivots
//               and multipliers travel by ONE DP-ALU DPP move each, v_mov_b64_dpp row_newbcast) and inverts its four 4 x 4
//               diagonal blocks;
//   waves 1-7   MEANWHILE form the product of panel k+1 over the columns < k0 (everything that does not depend on panel
//               k), 1-3 row tiles each, L operands fetched 3-8 k-steps ahead, B operands one k-step ahead;
//   waves 1-7   then form L21' = L11^-1 S21' of their OWN tiles by 4 x 4 block forward substitution (the output layout of one MFMA
//               is the B-operand layout of the next: no transposes), store it, and - its registers being at the same time the A
//               operands of the 16 new columns - finish the product of panel k+1 with them, add the panel of A and leave S of
//               panel k+1 in LDS where S of panel k was.
// LDS: U[16][n] holds, for panel k, S(c, row) at rows >= k0 and - in the rows < k0 that S no longer needs - the B operands
// -L(k0 + 16 + c, j) of the next product; two buffers An[2][16][n] take the panels of A in turn.  The staging copies (rows of L
// that become B operands, the next-but-one panel of A; global -> LDS, 8 loads in flight per thread) are dealt to all waves,
// each doing its share after its own work of a phase.  Late panels (fewer tiles than product waves): see `late` below.
// L' (upper triangle) and the inverses of the 16 x 16 diagonal blocks, which only the TRSM kernel needs, are produced
// afterwards by kp_chol_finish_kernel, off the critical path.
#include "kp_internal.h"

#define LL_NT 512
#define LL_NPW 7                         // product waves: 1 .. 7
#define LL_TMAX 3                        // 16-row tiles per product wave: ceil((n / 16 - 1) / 7)  =>  n <= 352
#define LL_NMAX 352
#define LL_DS 17                         // row stride of the diagonal block in LDS

__device__ long long kp_chol_ll_prof[16];
__device__ int kp_chol_ll_pt[2][24][8];   // KP_CHOL_PROF=2: [phase 2 | phase 4][panel][wave] cycles   // KP_CHOL_PROF=1: cycles per phase of the last launch (system 0, wave 0)

namespace {

// value of lane N of every 16-lane row, in all lanes of the row: one DP-ALU DPP move (row_newbcast), no trip through SGPRs
template <int N>
__device__ __forceinline__ double ll_rowbc(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, true);
}

template <int C, int CC>
__device__ __forceinline__ void ll_diag_update(double (&row)[16], double l) {
  if constexpr (CC < 16) {
    row[CC] -= l * ll_rowbc<CC>(l);               // entries with CC <= r are used
    ll_diag_update<C, CC + 1>(row, l);
  }
}

// Pivot steps C .. 15 of a 16 x 16 block held as: lane r of every 16-lane row owns row r.  A pivot that is rounding noise
// of its original diagonal entry (n * 8 eps of it: odiag) is as singular as a non-positive one.
// (round 6: nothing in a step is predicated - the verdict on the pivots is collected in a register (every lane sees the same
// broadcast pivot) and 1 / diag in the lane that owns the row, both stored once behind the 16 steps; the masked stores and the
// `bad` branch of every step had put exec-mask saves, v_readlane restores of spilled masks and a scalar branch into the chain
// of dependent instructions that IS this routine's time: 163 000 -> 120 000 cycles for the 21 blocks of n = 336.)
template <int C>
__device__ __forceinline__ void ll_diag_step(double (&row)[16], int lane, double& dd_own, int& badv, const double* __restrict__ odiag) {
  if constexpr (C < 16) {
    double d = ll_rowbc<C>(row[C]);
    const bool ok = d > odiag[C];
    badv |= ok ? 0 : 1;
    d = ok ? d : 1.0;
    double id = __builtin_amdgcn_rsq(d);          // 1/sqrt(d): hardware estimate + two Newton steps (full f64 accuracy)
    id = id * (1.5 - 0.5 * d * id * id);
    id = id * (1.5 - 0.5 * d * id * id);
    dd_own = (lane & 15) == C ? id : dd_own;
    const double l = row[C] * id;                 // lane C: sqrt(d); lanes r > C: L_rC
    row[C] = l;
    ll_diag_update<C, C + 1>(row, l);
    ll_diag_step<C + 1>(row, lane, dd_own, badv, odiag);
  }
}

__device__ __forceinline__ void ll_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inverse of the lower triangular 4 x 4 block of Dm at (o, o), [i * 4 + k]; dd = 1 / its diagonal
__device__ __forceinline__ void ll_inv4(const double (*Dm)[LL_DS], const double* dd, int o, double* q) {
  const double x00 = dd[o], x11 = dd[o + 1], x22 = dd[o + 2], x33 = dd[o + 3];
  const double l10 = Dm[o + 1][o], l20 = Dm[o + 2][o], l21 = Dm[o + 2][o + 1];
  const double l30 = Dm[o + 3][o], l31 = Dm[o + 3][o + 1], l32 = Dm[o + 3][o + 2];
  const double x10 = -(l10 * x00) * x11;
  const double x21 = -(l21 * x11) * x22;
  const double x20 = -(l20 * x00 + l21 * x10) * x22;
  const double x32 = -(l32 * x22) * x33;
  const double x31 = -(l31 * x11 + l32 * x21) * x33;
  const double x30 = -(l30 * x00 + l31 * x10 + l32 * x20) * x33;
  q[0] = x00; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0;
  q[4] = x10; q[5] = x11; q[6] = 0.0; q[7] = 0.0;
  q[8] = x20; q[9] = x21; q[10] = x22; q[11] = 0.0;
  q[12] = x30; q[13] = x31; q[14] = x32; q[15] = x33;
}

// The 16 x 16 diagonal block of panel k by ONE wave: S is in U (U[c * ldu + k0 + r]); results: L11 in Dm (lower), 1 / diag in
// Dd, the inverses of the four 4 x 4 diagonal blocks in I4.
__device__ __forceinline__ void ll_diag16(const double* __restrict__ U, int ldu, int k0, double (*Dm)[LL_DS], double* Dd, double (*I4)[16], int* bad,
                                          const double* __restrict__ odiag) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15;
  double row[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) row[c] = U[c * ldu + k0 + r];
  double dd_own = 0.0;
  int badv = 0;
  ll_diag_step<0>(row, lane, dd_own, badv, odiag + k0);
  if (badv && lane == 0) *bad = 1;
  if (lane < 16) {
    Dd[r] = dd_own;
#pragma unroll
    for (int c = 0; c < 16; ++c) Dm[r][c] = c <= r ? row[c] : 0.0;
  }
  ll_wave_sync();
  if (lane < 4) ll_inv4(Dm, Dd, 4 * lane, I4[lane]);
}

// L11 (lower, and mirrored: the diagonal block of L') to memory - nobody in this kernel reads it back from there; one wave
__device__ __forceinline__ void ll_store_diag(const double (*Dm)[LL_DS], double* __restrict__ A, int n, int k0, int lane) {
  for (int e = lane; e < 256; e += 64) {
    const int rr = e & 15, c = e >> 4;
    if (c <= rr) {
      const double v = Dm[rr][c];
      A[(size_t)(k0 + c) * n + k0 + rr] = v;
      A[(size_t)(k0 + rr) * n + k0 + c] = v;
    }
  }
}

// tiles of the product waves for the panel that starts at column k1: wave pw of LL_NPW takes tiles [tbeg, tbeg + Tw) of the
// (n - k1) / 16 row tiles from row k1 on
__device__ __forceinline__ void ll_tiles(int n, int k1, int pw, int& tbeg, int& Tw) {
  const int ntq = k1 < n ? (n - k1) / 16 : 0;
  const int tq = ntq / LL_NPW, tr = ntq % LL_NPW;
  Tw = pw >= 0 ? tq + (pw < tr ? 1 : 0) : 0;
  tbeg = pw >= 0 ? pw * tq + min(pw, tr) : 0;
}

// Product of panel k+1 over the finished columns [4 ks_lo, 4 ks_hi): acc(t, m) = - sum_j L(rows of tile t, j) L(k1 + 4 m + jj, j), the L
// operands from memory (PF k-steps in flight), the B operands from U (rows < k0 hold -L(k1 + c, j) at U[c][j]), one k-step ahead.
// Straight-line code for a given number of tiles TW: the compiler's s_waitcnt placement then counts the loads exactly (a
// conditional load anywhere in the loop would make it drain the queue every k-step).  Steps past the end re-read the last
// one against a zero B operand.
template <int TW, int PF, bool EXACT>
__device__ __forceinline__ void ll_product_body(double (&acc)[LL_TMAX][4], const double* __restrict__ ap, int n, const double* __restrict__ U, int ldu,
                                                int ks_lo, int ks_hi, int kq, int jj) {
  const int last = ks_hi - 1, cmax = n / 4 - 1;          // last k-step of the range / of the matrix (prefetch clamp)
  double an[PF][TW];
  // (the scheduling barriers keep the load groups in program order, so that the wait before a group's first use is
  // "all but the PF - 1 younger groups" and not "all")
#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int t = 0; t < TW; ++t) an[p][t] = ap[(size_t)(4 * min(ks_lo + p, EXACT ? cmax : last)) * n + 16 * t];
    __builtin_amdgcn_sched_barrier(0);
  }
  double bn[4];
  const double* bp0 = U + jj * ldu + kq;
#pragma unroll
  for (int m = 0; m < 4; ++m) bn[m] = bp0[(4 * m) * ldu + 4 * ks_lo];
  for (int ks = ks_lo; ks < ks_hi; ks += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kc = ks + p;
      // the sign is on the B operand (U holds -L in its B-operand rows) and the L operands feed the MFMAs from the registers they
      // were loaded into; their refill for step kc + PF is issued after the MFMAs of step kc.  EXACT (the range is a whole
      // number of groups): no masking of the B operand, and its prefetch address is linear in kc (immediate offsets; the read
      // one step past the range stays inside U)
      double b[4];
      if (EXACT) {
#pragma unroll
        for (int m = 0; m < 4; ++m) b[m] = bn[m];
#pragma unroll
        for (int m = 0; m < 4; ++m) bn[m] = bp0[(4 * m) * ldu + 4 * (kc + 1)];
      } else {
        const bool on = kc <= last;
#pragma unroll
        for (int m = 0; m < 4; ++m) b[m] = on ? bn[m] : 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) bn[m] = bp0[(4 * m) * ldu + 4 * min(kc + 1, last)];
      }
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[p][t], b[m], acc[t][m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const size_t off = (size_t)(4 * min(kc + PF, EXACT ? cmax : last)) * n;
#pragma unroll
      for (int t = 0; t < TW; ++t) an[p][t] = ap[off + 16 * t];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int TW, int PF>
__device__ __forceinline__ void ll_product_mem(double (&acc)[LL_TMAX][4], const double* __restrict__ ap, int n, const double* __restrict__ U, int ldu,
                                               int ks_lo, int ks_hi, int kq, int jj) {
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[t][m] = 0.0;
  if (ks_lo >= ks_hi) return;
  if ((ks_hi - ks_lo) % PF == 0) ll_product_body<TW, PF, true>(acc, ap, n, U, ldu, ks_lo, ks_hi, kq, jj);
  else ll_product_body<TW, PF, false>(acc, ap, n, U, ldu, ks_lo, ks_hi, kq, jj);
}

// L21 of this wave's OWN tiles (the rows it will finish panel k+1 for): X' = L11^-1 S' by 4 x 4 block forward substitution.  A
// operand (i = lane & 3, k = lane >> 4): a 4 x 4 block of L11 / an inverse diagonal block, the same for all 4 MFMA blocks; B
// operand and output (k or i = lane >> 4, j = lane & 3): X'(4 J + k, 4 blk + j) - identical layouts, no transposes; and the
// output x[t][J] is at the same time the A operand (rows 4 blk + jj, column 4 J + kq) of the 16 new columns' product.  X goes
// to memory; the tile of rows k1 .. k1+15 also leaves -X in U as the B operands of the new columns.
template <int TW>
__device__ __forceinline__ void ll_l21_own(double (&x)[LL_TMAX][4], const double* __restrict__ U, double* __restrict__ Ub, double* __restrict__ A, int n,
                                           int ldu, int k0, int row0, bool first_tile, const double (*Dm)[LL_DS], const double (*I4)[16], int kq, int blk,
                                           int jj) {
  const int ai = jj, ak = kq;
  const double i0 = I4[0][ai * 4 + ak], i1 = I4[1][ai * 4 + ak], i2 = I4[2][ai * 4 + ak], i3 = I4[3][ai * 4 + ak];
  const double m10 = -Dm[4 + ai][ak], m20 = -Dm[8 + ai][ak], m21 = -Dm[8 + ai][4 + ak];
  const double m30 = -Dm[12 + ai][ak], m31 = -Dm[12 + ai][4 + ak], m32 = -Dm[12 + ai][8 + ak];
  double sv[TW][4];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int J = 0; J < 4; ++J) sv[t][J] = U[(4 * J + kq) * ldu + row0 + 16 * t + 4 * blk + jj];
#pragma unroll
  for (int t = 0; t < TW; ++t) x[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(i0, sv[t][0], 0.0, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    sv[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(m10, x[t][0], sv[t][1], 0, 0, 0);
    sv[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(m20, x[t][0], sv[t][2], 0, 0, 0);
    sv[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(m30, x[t][0], sv[t][3], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < TW; ++t) x[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(i1, sv[t][1], 0.0, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    sv[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(m21, x[t][1], sv[t][2], 0, 0, 0);
    sv[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(m31, x[t][1], sv[t][3], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < TW; ++t) x[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(i2, sv[t][2], 0.0, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t) sv[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(m32, x[t][2], sv[t][3], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t) x[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(i3, sv[t][3], 0.0, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    double* dst = A + (size_t)(k0 + kq) * n + row0 + 16 * t + 4 * blk + jj;   // L(r, k0 + 4 J + kq)
#pragma unroll
    for (int J = 0; J < 4; ++J) dst[(size_t)(4 * J) * n] = x[t][J];
  }
  if (first_tile) {                                  // rows k1 .. k1+15 (row0 = k1): U[c'][k0 + 4 J + kq] = -L(k1 + c', k0 + 4 J + kq)
    double* db = Ub + (4 * blk + jj) * ldu + k0 + kq;
#pragma unroll
    for (int J = 0; J < 4; ++J) db[4 * J] = -x[0][J];
  }
}

// The 16 new columns: acc(t, m) += X(rows of t, k0 + 4 J + k) * (-L(k1 + 4 m + j, k0 + 4 J + k)), the A operands straight from the
// L21 registers, the B operands from U; then the panel of A (An) is added and S of panel k+1 goes to U over this wave's rows.
template <int TW>
__device__ __forceinline__ void ll_finish_own(double (&acc)[LL_TMAX][4], const double (&x)[LL_TMAX][4], double* __restrict__ U,
                                              const double* __restrict__ An, int ldu, int k0, int row0, int kq, int blk, int jj) {
  double b[4][4];
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int m = 0; m < 4; ++m) b[J][m] = U[(4 * m + jj) * ldu + k0 + 4 * J + kq];
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[t][J], b[J][m], acc[t][m], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int o = (4 * m + jj) * ldu + row0 + 16 * t + 4 * blk + kq;
      U[o] = acc[t][m] + An[o];
    }
}

// columns [col0, col0 + 16) of A, rows >= row_lo, into a panel buffer P[c][row], by the whole workgroup (every wave at the end of
// its own work of the phase): 32 threads per column, 8 loads in flight per thread
__device__ __forceinline__ void ll_stage_panel(const double* __restrict__ A, int n, int col0, int row_lo, double* __restrict__ P, int ldu, int tid) {
  if (col0 >= n) return;
  const int c = tid >> 5, r0 = row_lo + (tid & 31);
  const double* src = A + (size_t)(col0 + c) * n;
  for (int rb = r0; rb < n; rb += 32 * 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[min(rb + 32 * q, n - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (rb + 32 * q < n) P[c * ldu + rb + 32 * q] = v[q];
  }
}

// rows [row0, row0 + 16) of -L, columns < ncol, into U[c][j] (the B operands of a later product), by the whole workgroup
__device__ __forceinline__ void ll_stage_rows(const double* __restrict__ A, int n, int row0, int ncol, double* __restrict__ U, int ldu, int tid) {
  for (int e0 = tid; e0 < ncol * 16; e0 += LL_NT * 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = min(e0 + LL_NT * q, ncol * 16 - 1);
      v[q] = A[(size_t)(e >> 4) * n + row0 + (e & 15)];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + LL_NT * q;
      if (e < ncol * 16) U[(e & 15) * ldu + (e >> 4)] = -v[q];
    }
  }
}

}  // namespace

// thr (or nullptr): pivot thresholds of the n rows supplied by the caller - the blocked factorisation of wide systems hands
// down the thresholds of the ORIGINAL diagonal, which a trailing-updated diagonal block no longer shows (single systems only)
__global__ __launch_bounds__(LL_NT) void kp_chol_ll_kernel(double* __restrict__ A, int n, int* __restrict__ info, int* __restrict__ sticky,
                                                           int prof, const double* __restrict__ thr) {
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;   // KP_CHOL_PROF=1: cycles per phase (wave 0)
#define LL_TICK(i) do { if (prof) { long long tnow = clock64(); tph[i] += tnow - tlast; tlast = tnow; } } while (0)
  extern __shared__ __align__(16) double sm[];
  __shared__ double Dm[16][LL_DS];       // L11 of the current panel (lower)
  __shared__ double Dd[16];              // 1 / diag(L11)
  __shared__ double I4[4][16];           // inverses of the four 4 x 4 diagonal blocks of L11, [J][i * 4 + k]
  __shared__ double odiag[LL_NMAX];      // pivot thresholds: n * 8 eps * original diagonal (0 for the identity padding)
  __shared__ int bad;
  const int ldu = n + 4;
  double* U = sm;                        // [16][ldu]
  double* An0 = U + 16 * ldu;            // panels of A, in turn: panel p lives in An[p & 1]
  double* An1 = An0 + 16 * ldu;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: loop bounds and tile counts live in SGPRs
  A += blockIdx.y * (size_t)n * n; info += blockIdx.y;   // system of a batch
  if (tid == 0) bad = 0;
  for (int i = tid; i < n; i += LL_NT) odiag[i] = thr ? thr[i] : fmax(A[(size_t)i * n + i], 0.0) * ((double)n * 8.0 * 2.220446049250313e-16);
  // panel 0: S = A(:, 0:16); panel 1 of A into its buffer
  for (int e = tid; e < 16 * n; e += LL_NT) {
    const int c = e / n, x = e - c * n;
    U[c * ldu + x] = A[(size_t)c * n + x];
    if (n > 16 && x >= 16) An1[c * ldu + x] = A[(size_t)(16 + c) * n + x];
  }
  __syncthreads();
  if (prof) tlast = clock64();
  const int nt = n / 16;
  double acc[LL_TMAX][4];
  for (int kb = 0; kb < nt; ++kb) {
    // a pivot below its threshold: the caller goes to the rank-revealing solve, nothing of this factor is used - stop here
    // (`bad` was last written before the barrier that ended the previous panel: uniform)
    if (bad) break;
    const int k0 = kb * 16, k1 = k0 + 16;
    // lane coordinates, opaque to the optimiser once per panel: otherwise every per-lane address of every phase and variant is
    // hoisted out of this loop and kept live across it (the kernel then spills; recomputing them is a few VALU ops per phase)
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    const int kq = lane_v >> 4, blk = (lane_v >> 2) & 3, jj = lane_v & 3;
    const int ntp = nt - kb;                       // 16-row tiles of panel k; tile 0 = the diagonal block
    const int ntq = ntp - 1;                       // tiles of panel k+1, dealt to the product waves
    const bool pwave = wave != 0;
    int tbeg, Tw;
    ll_tiles(n, k1, wave - 1, tbeg, Tw);
    const int row0 = k1 + 16 * tbeg;               // first row of this wave's tiles
    double* An = (kb + 1) & 1 ? An1 : An0;         // panel k+1 of A
    const bool late = ntq >= 1 && ntq < LL_NPW && k1 >= 256;     // (the partial sums need 256 free rows in the panel buffers)
    // ---- B operands of the next product, U[c][j] = L(k1 + c, j), in the rows of U that S no longer uses: the columns of the
    //      previous panel here, the older ones were staged by wave 0 during the previous panel's last phase ----
    if (ntq > 0 && k0 > 0 && tid < 256) {
      const int c = tid & 15, j = k0 - 16 + (tid >> 4);
      U[c * ldu + j] = -A[(size_t)j * n + k1 + c];
    }
    __syncthreads();
    LL_TICK(0);
    // ---- wave 0: the diagonal block of panel k  |  waves 1-7: product of panel k+1 over the columns < k0 ----
    if (wave == 0) {
      ll_diag16(U, ldu, k0, Dm, Dd, I4, &bad, odiag);
      if (prof) tph[4] += clock64() - tlast;       // the diagonal block alone

    } else if (late) {
      // Late panels: fewer tiles than product waves and 60-76 k-steps each, at ~220 cycles of fixed cost per k-step.  Tiles are
      // taken in PAIRS (8 MFMAs per k-step instead of 4); the (pair, k-step) items, pair-major, are cut into 7 equal ranges; a
      // range touches at most two pairs, whose partial sums go to LDS (the rows < k1 of the two panel buffers are unused by
      // now) and are added up in a fixed order by each tile's owner in the last phase.
      const int K = k0 / 4, npair = (ntq + 1) / 2, Ltot = npair * K, pw = wave - 1;
      const int lo = pw * Ltot / LL_NPW, hi = (pw + 1) * Ltot / LL_NPW;
      const int pa = lo / K;
#pragma unroll
      for (int piece = 0; piece < 2; ++piece) {
        const int pr = pa + piece;                                                   // tile pair 2 pr, 2 pr + 1
        const int ks_lo = piece == 0 ? lo - pa * K : 0, ks_hi = min(K, hi - pr * K);
        double accp[LL_TMAX][4];
        const int t0 = min(2 * pr, ntq - 1);
        const double* ap = A + (size_t)kq * n + k1 + 16 * t0 + 4 * blk + jj;
        const bool two = 2 * pr + 1 < ntq;
        if (two) ll_product_mem<2, 4>(accp, ap, n, U, ldu, ks_lo, ks_hi, kq, jj);   // (an empty range leaves zeros)
        else {
          ll_product_mem<1, 8>(accp, ap, n, U, ldu, ks_lo, ks_hi, kq, jj);
#pragma unroll
          for (int m = 0; m < 4; ++m) accp[1][m] = 0.0;
        }
        double* sc = piece == 0 ? An : ((kb + 2) & 1 ? An1 : An0);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int m = 0; m < 4; ++m) sc[(2 * pw + (m >> 1)) * ldu + t2 * 128 + (m & 1) * 64 + lane_v] = accp[t2][m];
      }
    } else {
      const double* ap = A + (size_t)kq * n + row0 + 4 * blk + jj;                 // + 4 ks n + 16 t : L(row0 + 16 t + 4 blk + i, 4 ks + kq)
      switch (Tw) {
        case 1: if (kb & 1) ll_product_mem<1, 4>(acc, ap, n, U, ldu, 0, k0 / 4, kq, jj); else ll_product_mem<1, 8>(acc, ap, n, U, ldu, 0, k0 / 4, kq, jj); break;
        case 2: ll_product_mem<2, 4>(acc, ap, n, U, ldu, 0, k0 / 4, kq, jj); break;
        case 3: ll_product_mem<3, 4>(acc, ap, n, U, ldu, 0, k0 / 4, kq, jj); break;
        default: break;
      }
    }
    // panel k+2 of A into the buffer nobody reads now: every wave its share, after its own work of this phase
    ll_stage_panel(A, n, k1 + 16, k1 + 16, (kb + 2) & 1 ? An1 : An0, ldu, wave * 64 + lane_v);
    if (prof == 2 && lane == 0 && blockIdx.y == 0) kp_chol_ll_pt[0][kb][wave] = (int)(clock64() - tlast);
    __syncthreads();
    LL_TICK(1);
    // ---- L21 of every product wave's own tiles (registers + memory; -X of the first tile into U)  |  wave 0: L11 to memory ----
    double xo[LL_TMAX][4];
    if (pwave) {
      switch (Tw) {
        case 1: ll_l21_own<1>(xo, U, U, A, n, ldu, k0, row0, tbeg == 0, Dm, I4, kq, blk, jj); break;
        case 2: ll_l21_own<2>(xo, U, U, A, n, ldu, k0, row0, tbeg == 0, Dm, I4, kq, blk, jj); break;
        case 3: ll_l21_own<3>(xo, U, U, A, n, ldu, k0, row0, tbeg == 0, Dm, I4, kq, blk, jj); break;
        default: break;
      }
    } else {
      ll_store_diag(Dm, A, n, k0, lane_v);
    }
    // (LDS-only: the stores of L stay in flight; the barrier at the end of the panel waits for them before anyone reads L)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    LL_TICK(2);
    // ---- waves 1-7: the 16 new columns, + the panel of A, S of panel k+1 back into U  |  wave 0: stores and staging ----
    if (late && Tw == 1) {                         // the owner of tile wave - 1 collects its partial sums, waves and pieces in order
      const int K = k0 / 4, npair = (ntq + 1) / 2, Ltot = npair * K, mine = wave - 1;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[0][m] = 0.0;
      for (int w = 0; w < LL_NPW; ++w) {
        const int lo = w * Ltot / LL_NPW, hi = (w + 1) * Ltot / LL_NPW;
        const int pa = lo / K;
        for (int piece = 0; piece < 2; ++piece) {
          const int pr = pa + piece;
          if (pr != (mine >> 1) || min(K, hi - pr * K) <= (piece == 0 ? lo - pa * K : 0)) continue;
          const double* sc = piece == 0 ? An : ((kb + 2) & 1 ? An1 : An0);
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[0][m] += sc[(2 * w + (m >> 1)) * ldu + (mine & 1) * 128 + (m & 1) * 64 + lane_v];
        }
      }
    }
    if (pwave) {
      switch (Tw) {
        case 1: ll_finish_own<1>(acc, xo, U, An, ldu, k0, row0, kq, blk, jj); break;
        case 2: ll_finish_own<2>(acc, xo, U, An, ldu, k0, row0, kq, blk, jj); break;
        case 3: ll_finish_own<3>(acc, xo, U, An, ldu, k0, row0, kq, blk, jj); break;
        default: break;
      }
    }
    // B operands of the product after the next one, columns < k0 (rows k1+16 .. k1+31 of L; the rows < k0 of U are free now -
    // the new-column products read rows k0 .. k1-1): every wave its share
    if (ntq > 1) ll_stage_rows(A, n, k1 + 16, k0, U, ldu, wave * 64 + lane_v);
    if (prof == 2 && lane == 0 && blockIdx.y == 0) kp_chol_ll_pt[1][kb][wave] = (int)(clock64() - tlast);
    __syncthreads();
    LL_TICK(3);
  }
  if (prof && tid == 0 && blockIdx.y == 0)
    for (int i = 0; i < 5; ++i) kp_chol_ll_prof[i] = tph[i];
  if (tid == 0) {
    *info = bad;
    if (bad && sticky) *sticky = 1;
  }
}

bool kp_chol_ll_applicable(int n) {
  static const bool off = getenv("KP_CHOL_OLD") != nullptr;
  return !off && n <= LL_NMAX;
}

static size_t kp_chol_ll_lds_bytes(int n) { return (size_t)3 * 16 * (n + 4) * sizeof(double); }

hipError_t kp_chol_ll_launch(double* Gp, int n, int nb, int* info, int* sticky, int prof, hipStream_t st, const double* thr) {
  static KpLdsCache lds_cache;
  const size_t lds = kp_chol_ll_lds_bytes(n);
  hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_chol_ll_kernel, kp_chol_ll_lds_bytes(LL_NMAX));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kp_chol_ll_kernel, dim3(1, nb), dim3(LL_NT), lds, st, Gp, n, info, sticky, prof, thr);
  e = hipGetLastError();
  if (prof && e == hipSuccess) {                  // diagnostic only: synchronous
    long long t[16];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpyFromSymbol(t, HIP_SYMBOL(kp_chol_ll_prof), sizeof(t)) == hipSuccess)
      fprintf(stderr, "chol-ll cycles (wave 0): new B columns %lld  diag | product %lld (diag alone %lld)  L21 %lld  new columns | staging %lld  (n = %d)\n",
              t[0], t[1], t[4], t[2], t[3], n);
    if (prof == 2) {
      static int pt[2][24][8];
      if (hipMemcpyFromSymbol(pt, HIP_SYMBOL(kp_chol_ll_pt), sizeof(pt)) == hipSuccess)
        for (int ph = 0; ph < 2; ++ph)
          for (int kb = 0; kb < n / 16; ++kb)
            fprintf(stderr, "  %s panel %2d: %6d | %6d %6d %6d %6d %6d %6d %6d\n", ph ? "new columns " : "diag|product", kb, pt[ph][kb][0], pt[ph][kb][1],
                    pt[ph][kb][2], pt[ph][kb][3], pt[ph][kb][4], pt[ph][kb][5], pt[ph][kb][6], pt[ph][kb][7]);
    }
  }
  return e;
}

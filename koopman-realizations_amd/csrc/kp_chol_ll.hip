// Left-looking blocked Cholesky of one (padded) n x n Gram matrix by ONE workgroup, n <= 352 (the config shapes:
// W = 336, 200, 136), on the matrix pipe.  K = Px \ Py (Ksysid.m:1069) through the normal equations: this is the
// factorisation between the fused Gram kernel and the block substitution (kp_trsm_kernel).
//
// Why left-looking: a right-looking sweep rewrites the whole trailing matrix once per 16-column panel - 12.6 MB through
// one CU's 64 B/clk path to L2 at n = 336, twice the time of its flops.  Here panel k gathers all earlier updates in
// one product,
//     S = A(k0:n, k0:k0+16) - L(k0:n, 0:k0) L(k0:k0+16, 0:k0)',
// reading L once (3.2 MB in total) and keeping S in accumulators: v_mfma_f64_4x4x4 (4 blocks = the 4 row groups of a
// 16-row tile; B = 4 panel columns, shared by the blocks), wave = (row quarter, half of the contraction range), the two
// half sums combined through LDS (fixed order: bitwise reproducible); the L operands are fetched three k-steps ahead.  Then
//     wave 0 factors the 16 x 16 diagonal block in registers (pivots travel by v_readlane) and inverts its four
//     4 x 4 diagonal blocks; all waves form L21' = L11^-1 S21' by 4 x 4 block forward substitution on the matrix pipe
//     (the output layout of one MFMA is the B-operand layout of the next: no transposes) and store L.
// L' (upper triangle) and the inverses of the 16 x 16 diagonal blocks, which only the TRSM kernel needs, are produced
// afterwards by kp_chol_finish_kernel, off the critical path.
#include "kp_internal.h"

#define LL_NT 512
#define LL_TMAX 6                        // 16-row tiles per wave: ceil((n / 16) / 4)  =>  n <= 352 (22 tiles: 6 + 6 + 5 + 5)
#define LL_NMAX 352
#define LL_PBUF (4 * LL_TMAX * 4 * 64)   // doubles of the partial buffer: [row quarter][tile][m][lane]
#define LL_PF 3                          // k-steps of L operands in flight
#define LL_LDS 17                        // row stride of the diagonal block in LDS

__device__ long long kp_chol_ll_prof[8];   // KP_CHOL_PROF=1: cycles per phase of the last launch (system 0)

namespace {

// value of lane N of every 16-lane row, in all lanes of the row: one DP-ALU DPP move (row_newbcast), no trip through SGPRs
template <int N>
__device__ __forceinline__ double ll_rowbc(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, false);
}

template <int C, int CC>
__device__ __forceinline__ void ll_diag_update(double (&row)[16], double l) {
  if constexpr (CC < 16) {
    row[CC] -= l * ll_rowbc<CC>(l);               // entries with CC <= r are used
    ll_diag_update<C, CC + 1>(row, l);
  }
}

template <int C>
__device__ __forceinline__ void ll_diag_step(double (&row)[16], int lane, int k0, double* Dd, int* bad, const double* __restrict__ odiag) {
  if constexpr (C < 16) {
    double d = ll_rowbc<C>(row[C]);
    if (!(d > odiag[k0 + C])) {
      if (lane == 0) *bad = 1;
      d = 1.0;
    }
    double id = __builtin_amdgcn_rsq(d);          // 1/sqrt(d): hardware estimate + two Newton steps (full f64 accuracy)
    id = id * (1.5 - 0.5 * d * id * id);
    id = id * (1.5 - 0.5 * d * id * id);
    if (lane == C) Dd[C] = id;
    const double l = row[C] * id;                 // lane C: sqrt(d); lanes r > C: L_rC
    row[C] = l;
    ll_diag_update<C, C + 1>(row, l);
    ll_diag_step<C + 1>(row, lane, k0, Dd, bad, odiag);
  }
}

// 16 x 16 diagonal block by one wave, entirely in registers: lane r of every 16-lane row owns row r of S (LDS, [c][r],
// leading dimension ld; the four rows of lanes hold the same data).  Writes L11 to D (LDS), 1 / diag to Dd and - after the
// arithmetic - L11 (lower) / L11' (upper) to A.  A pivot that is rounding noise of its original diagonal entry (n * 8 eps
// of it) is as singular as a non-positive one.
__device__ __forceinline__ void ll_diag_block(const double* __restrict__ S, int ld, double* __restrict__ A, int n, int k0,
                                              double (*D)[LL_LDS], double* Dd, int* bad, const double* __restrict__ odiag) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15;
  double row[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) row[c] = S[c * ld + r];
  ll_diag_step<0>(row, lane, k0, Dd, bad, odiag);
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) D[r][c] = c <= r ? row[c] : 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c <= r) {
        A[(size_t)(k0 + c) * n + k0 + r] = row[c];   // L (lower)
        A[(size_t)(k0 + r) * n + k0 + c] = row[c];   // L' (upper)
      }
    }
  }
}

// acc(tile t, column group m) = [FIRST: the panel itself] - sum over the k-steps [ks_lo, ks_hi) of L(rows of t, 4 ks + k) *
// L(k0 + 4 m + j, 4 ks + k).  Straight-line code for a given number of tiles TW (the loads are counted exactly by the
// compiler's s_waitcnt placement - a conditional load anywhere in the loop would force it to drain the queue every
// k-step): PF k-steps of L operands are in flight; steps past ks_hi re-read the last one against a zero B operand.
template <int TW, int PF, bool FIRST>
__device__ __forceinline__ void ll_product(double (&acc)[LL_TMAX][4], const double* __restrict__ ap, const double* __restrict__ pp, int n,
                                           const double* __restrict__ Bs, int ks_lo, int ks_hi, int kq, int jj) {
  double pa[TW][4];
  if (FIRST) {
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) pa[t][m] = pp[(size_t)(4 * m) * n + 16 * t];
  }
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[t][m] = 0.0;
  if (ks_lo < ks_hi) {
    const int last = ks_hi - 1;
    double an[PF][TW];
    // (the scheduling barriers keep the load groups in program order, so that the wait before a group's first use is
    // "all but the PF - 1 younger groups" and not "all")
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int t = 0; t < TW; ++t) an[p][t] = ap[(size_t)(4 * min(ks_lo + p, last)) * n + 16 * t];
      __builtin_amdgcn_sched_barrier(0);
    }
    // B operands one k-step ahead: their LDS round trip would otherwise sit in front of every k-step
    const double2* bp0 = reinterpret_cast<const double2*>(Bs + (4 * ks_lo + kq) * 16 + jj * 4);
    double2 n01 = bp0[0], n23 = bp0[1];
    for (int ks = ks_lo; ks < ks_hi; ks += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int kc = ks + p;
        // the sign goes on the B operand (4 values) and the L operands feed the MFMAs from the registers they were loaded
        // into; their refill for step kc + PF is issued after the MFMAs of step kc
        const bool on = kc <= last;
        const double b0 = on ? -n01.x : 0.0, b1 = on ? -n01.y : 0.0, b2 = on ? -n23.x : 0.0, b3 = on ? -n23.y : 0.0;
        const double2* bp = reinterpret_cast<const double2*>(Bs + (4 * min(kc + 1, last) + kq) * 16 + jj * 4);
        n01 = bp[0];
        n23 = bp[1];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[p][t], b0, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[p][t], b1, acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[p][t], b2, acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(an[p][t], b3, acc[t][3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const size_t off = (size_t)(4 * min(kc + PF, last)) * n;
#pragma unroll
        for (int t = 0; t < TW; ++t) an[p][t] = ap[off + 16 * t];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (FIRST) {
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] += pa[t][m];
  }
}

}  // namespace

__global__ __launch_bounds__(LL_NT) void kp_chol_ll_kernel(double* __restrict__ A, int n, int* __restrict__ info, int* __restrict__ sticky,
                                                           int prof) {
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;   // KP_CHOL_PROF=1: cycles per phase, printed by thread 0
#define LL_TICK(i) do { if (prof) { long long tnow = clock64(); tph[i] += tnow - tlast; tlast = tnow; } } while (0)
  extern __shared__ __align__(16) double sm[];
  __shared__ double D[16][LL_LDS];       // L11 of the current panel
  __shared__ double Dd[16];              // 1 / diag(L11)
  __shared__ double I4[4][16];           // inverses of the four 4 x 4 diagonal blocks of L11, [J][i * 4 + k]
  __shared__ double odiag[LL_NMAX]; // pivot thresholds: n * 8 eps * original diagonal (0 for the identity padding)
  __shared__ int bad;
  double* Bs = sm;                                 // [j][16]: L(k0 + c, j), columns permuted (c & 3) * 4 + (c >> 2)
  double* S = sm + (size_t)(n - 16) * 16;          // the finished panel, [c][row - k0]
  double* P = S + (size_t)16 * n;                  // partial sums of the second contraction half
  const int ldS = n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: loop bounds and tile counts live in SGPRs
  const int rq = wave & 3, s = wave >> 2;          // row quarter, contraction half
  const int kq = lane >> 4, blk = (lane >> 2) & 3, jj = lane & 3;
  A += blockIdx.y * (size_t)n * n; info += blockIdx.y;   // system of a batch
  if (tid == 0) bad = 0;
  for (int i = tid; i < n; i += LL_NT) odiag[i] = fmax(A[(size_t)i * n + i], 0.0) * ((double)n * 8.0 * 2.220446049250313e-16);
  __syncthreads();
  if (prof) tlast = clock64();
  const int nt = n / 16;
  for (int kb = 0; kb < nt; ++kb) {
    const int k0 = kb * 16;
    const int ntp = nt - kb;                       // 16-row tiles of the panel, tile 0 = the diagonal block
    const int tq = ntp / 4, tr = ntp % 4;          // quarters: tq + 1 tiles for the first tr, tq for the rest
    const int tbeg = rq * tq + min(rq, tr), Tw = tq + (rq < tr ? 1 : 0);
    // ---- rows k0 .. k0+15 of L (columns < k0) -> LDS: the B operands of this panel ----
    for (int e = tid; e < k0 * 16; e += LL_NT) {
      const int c = e & 15, j = e >> 4;
      Bs[j * 16 + (c & 3) * 4 + (c >> 2)] = A[(size_t)j * n + k0 + c];
    }
    __syncthreads();
    LL_TICK(0);
    // ---- S = A_panel - L(rows, 0:k0) L(k0:k0+16, 0:k0)' : wave (rq, s) = tiles of quarter rq, k-steps of half s ----
    double acc[LL_TMAX][4];
    {
      const int ks_all = k0 / 4;                   // k-steps of 4 columns; a multiple of 4
      const int ks_lo = s * (ks_all / 2), ks_hi = ks_lo + ks_all / 2;
      const double* ap = A + (size_t)kq * n + k0 + 16 * tbeg + 4 * blk + jj;    // + 4 ks n + 16 t : L(r0 + 4 blk + i, 4 ks + kq)
      const double* pp = A + (size_t)(k0 + jj) * n + k0 + 16 * tbeg + 4 * blk + kq;   // + 4 m n + 16 t : the panel, output layout
      // fewer tiles per wave = shorter k-steps: more of them in flight (the L2 round trip is ~1000 cycles)
      if (s == 0) {
        switch (Tw) {
          case 1: ll_product<1, 8, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 2: ll_product<2, 6, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 3: ll_product<3, 4, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 4: ll_product<4, 3, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 5: ll_product<5, 3, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 6: ll_product<6, 2, true>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          default: break;
        }
      } else {
        switch (Tw) {
          case 1: ll_product<1, 8, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 2: ll_product<2, 6, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 3: ll_product<3, 4, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 4: ll_product<4, 3, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 5: ll_product<5, 3, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          case 6: ll_product<6, 2, false>(acc, ap, pp, n, Bs, ks_lo, ks_hi, kq, jj); break;
          default: break;
        }
      }
    }
    LL_TICK(1);
    // ---- the two half sums, lane-wise through LDS in the accumulator layout; the first half's waves finish the panel ----
    {
      double* pw = P + (size_t)rq * LL_TMAX * 256 + lane;
      if (s == 1) {
#pragma unroll
        for (int t = 0; t < LL_TMAX; ++t)
          if (t < Tw) {
#pragma unroll
            for (int m = 0; m < 4; ++m) pw[(t * 4 + m) * 64] = acc[t][m];
          }
      }
      __syncthreads();
      if (s == 0) {
#pragma unroll
        for (int t = 0; t < LL_TMAX; ++t)
          if (t < Tw) {
            const int r = 16 * (tbeg + t) + 4 * blk + kq;       // row - k0
#pragma unroll
            for (int m = 0; m < 4; ++m) S[(4 * m + jj) * ldS + r] = acc[t][m] + pw[(t * 4 + m) * 64];
          }
      }
      __syncthreads();
    }
    LL_TICK(2);
    // ---- diagonal block (wave 0), then the inverses of its 4 x 4 diagonal blocks ----
    if (wave == 0) {
      ll_diag_block(S, ldS, A, n, k0, D, Dd, &bad, odiag);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < 4) {
        const int o = 4 * lane;
        const double x00 = Dd[o], x11 = Dd[o + 1], x22 = Dd[o + 2], x33 = Dd[o + 3];
        const double l10 = D[o + 1][o], l20 = D[o + 2][o], l21 = D[o + 2][o + 1];
        const double l30 = D[o + 3][o], l31 = D[o + 3][o + 1], l32 = D[o + 3][o + 2];
        const double x10 = -(l10 * x00) * x11;
        const double x21 = -(l21 * x11) * x22;
        const double x20 = -(l20 * x00 + l21 * x10) * x22;
        const double x32 = -(l32 * x22) * x33;
        const double x31 = -(l31 * x11 + l32 * x21) * x33;
        const double x30 = -(l30 * x00 + l31 * x10 + l32 * x20) * x33;
        double* q = I4[lane];
        q[0] = x00; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0;
        q[4] = x10; q[5] = x11; q[6] = 0.0; q[7] = 0.0;
        q[8] = x20; q[9] = x21; q[10] = x22; q[11] = 0.0;
        q[12] = x30; q[13] = x31; q[14] = x32; q[15] = x33;
      }
    }
    __syncthreads();
    LL_TICK(3);
    if (ntp > 1) {
      // ---- L21' = L11^-1 S21' by 4 x 4 block forward substitution.  MFMA blocks = the 4 row groups of a 16-row tile;
      //      A operand (i = lane & 3, k = lane >> 4): a 4 x 4 block of L11 / an inverse diagonal block, the same for all
      //      blocks; B operand and output (k or i = lane >> 4, j = lane & 3): X'(4J + k, 4 blk + j) - identical layouts ----
      const int ai = jj, ak = kq;
      const double i0 = I4[0][ai * 4 + ak], i1 = I4[1][ai * 4 + ak], i2 = I4[2][ai * 4 + ak], i3 = I4[3][ai * 4 + ak];
      const double m10 = -D[4 + ai][ak], m20 = -D[8 + ai][ak], m21 = -D[8 + ai][4 + ak];
      const double m30 = -D[12 + ai][ak], m31 = -D[12 + ai][4 + ak], m32 = -D[12 + ai][8 + ak];
      for (int t = 1 + wave; t < ntp; t += 8) {
        const int r = 16 * t + 4 * blk + jj;       // row - k0
        const double s0 = S[(0 + kq) * ldS + r], s1 = S[(4 + kq) * ldS + r], s2 = S[(8 + kq) * ldS + r], s3 = S[(12 + kq) * ldS + r];
        const double x0 = __builtin_amdgcn_mfma_f64_4x4x4f64(i0, s0, 0.0, 0, 0, 0);
        double t1 = __builtin_amdgcn_mfma_f64_4x4x4f64(m10, x0, s1, 0, 0, 0);
        double t2 = __builtin_amdgcn_mfma_f64_4x4x4f64(m20, x0, s2, 0, 0, 0);
        double t3 = __builtin_amdgcn_mfma_f64_4x4x4f64(m30, x0, s3, 0, 0, 0);
        const double x1 = __builtin_amdgcn_mfma_f64_4x4x4f64(i1, t1, 0.0, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_4x4x4f64(m21, x1, t2, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f64_4x4x4f64(m31, x1, t3, 0, 0, 0);
        const double x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(i2, t2, 0.0, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f64_4x4x4f64(m32, x2, t3, 0, 0, 0);
        const double x3 = __builtin_amdgcn_mfma_f64_4x4x4f64(i3, t3, 0.0, 0, 0, 0);
        double* dst = A + (size_t)(k0 + kq) * n + k0 + r;          // L(k0 + r, k0 + 4J + kq)
        dst[0] = x0;
        dst[(size_t)4 * n] = x1;
        dst[(size_t)8 * n] = x2;
        dst[(size_t)12 * n] = x3;
      }
    }
    __syncthreads();                                // L of this panel is in memory before the next panel reads it
    LL_TICK(4);
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

size_t kp_chol_ll_lds_bytes(int n) { return ((size_t)(n - 16) * 16 + (size_t)16 * n + (size_t)LL_PBUF) * sizeof(double); }

hipError_t kp_chol_ll_launch(double* Gp, int n, int nb, int* info, int* sticky, int prof, hipStream_t st) {
  static KpLdsCache lds_cache;
  const size_t lds = kp_chol_ll_lds_bytes(n);
  hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_chol_ll_kernel, kp_chol_ll_lds_bytes(LL_NMAX));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kp_chol_ll_kernel, dim3(1, nb), dim3(LL_NT), lds, st, Gp, n, info, sticky, prof);
  e = hipGetLastError();
  if (prof && e == hipSuccess) {                  // diagnostic only: synchronous
    long long t[8];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpyFromSymbol(t, HIP_SYMBOL(kp_chol_ll_prof), sizeof(t)) == hipSuccess)
      fprintf(stderr, "chol-ll cycles: B rows %lld  product %lld  reduce %lld  diag %lld  L21 %lld  (n = %d)\n", t[0], t[1], t[2], t[3], t[4], n);
  }
  return e;
}

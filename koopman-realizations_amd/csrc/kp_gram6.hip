// Fused lift + Gram kernel for bilinear monomial dictionaries with three inputs, second form (round 4): ONE workgroup of EIGHT
// waves per CU and the WEIGHTED A operands in LDS.
//
// kp_gram3_kernel (kp_gram3.hip) gives every wave the raw A fragment psi_x and has it form the ten weighted copies
// w_ab psi_x itself: 9 v_mul_f64 per (A group, k-step), 18-36 per wave and 8-snapshot tile beside 120 MFMAs - and on gfx950
// nothing on the VALU overlaps the f64 MFMA stream of its SIMD (profiles/r01_coissue.txt).  A timing-only build without those
// multiplies (KP_ABL3=7) runs the 1e5-pair, W = 336 launch in 0.360 instead of 0.395 ms.  Here the LIFT writes the weighted
// columns: a Psi row is [ten weighted copies of psi_x, column by column | psi_y | zero group], 11 x 84 + 4 doubles, so that every
// MFMA operand - A as well as B - is one ds_read_b64 with an immediate offset and the MFMA loop holds no multiply at all.  That
// row is 7.4 KB; two buffers of eight rows are 121 KB, which leaves room for ONE workgroup per CU - so it has eight waves
// (two per SIMD, as before), and three workgroups instead of seven serve a snapshot split: the tile is lifted 3 times instead
// of 7, by 512 threads instead of 256, and a wave holds NQ = 7 quads (140 MFMAs per tile; 24 jobs of 7 = the 168 quads of
// N = 84 exactly).  The weights w_ab = ut_a ut_b themselves are power-table entries: the linear ones are the inputs' own
// entries, the six quadratic ones are written by the raw-loader threads of the input rows (which fetch the other inputs of
// their snapshot as well).  Per wave and tile that leaves ~20 multiplies of the lift, the power table and a handful of
// address updates against 140 MFMAs.
//
// Status: OPT-IN (KP_GRAM6=1), exact, 0.565 ms against kp_gram3's 0.395 ms for the 1e5-pair launch: the compiler spills 66 dwords of
// the 7-quad waves at 256 registers and surrounds the reads with ~100 v_mov_b64 / ~125 v_add_u32 per three tiles.
//
// Same partial layout, plan format and reduction as kp_gram3 ([split][job][quad][weight][lane]; kp_gram3_reduce_kernel), same
// tail mask (the table's entries are 0 past Ns), bitwise reproducible.  Replaces the per-row lift loop of
// Ksysid.get_Koopman (Ksysid.m:1030-1065) and Px'Px / Px'Py (:1114, :1125) for model_type 'bilinear', m = 3.
#include <type_traits>

#include "kp_gram3_args.h"

#define KT6 8                       // snapshots per tile (two k-steps)
#define NF6 3                       // table entries per monomial
#define NWT6 10                     // weights of three inputs: (m + 1)(m + 2) / 2
#define NID6 128                    // power-table entries per snapshot
#define PST6 10                     // doubles per entry: KT6 snapshots + 2 (bank spread of the 16-byte reads, as kp_gram3)
#define POWBUF6 (PST6 * NID6)

template <int G4C>
struct G6Layout {
  static constexpr int CS = 4 * G4C;                 // columns per side
  static constexpr int WS = NWT6;                    // psi_x: [column][weight] - the ten weighted copies of a column are contiguous (80 B:
                                                     // an A set is five 16-byte reads with immediate offsets; a stride of whole blocks made
                                                     // the compiler merge pairs into ds_read2_b64 behind a VALU add for every new base)
  static constexpr int YO = NWT6 * CS;               // psi_y [column]
  static constexpr int ZO = YO + CS;                 // zero group
  static constexpr int RS = ((ZO + 4 - 16 + 31) / 32) * 32 + 16;     // row stride = 16 mod 32: the four k-rows of an operand read hit disjoint banks
  static constexpr int PSIBUF = KT6 * RS;
  static constexpr int PSI0 = 2 * POWBUF6;
  static constexpr int LDS_DOUBLES = PSI0 + 2 * PSIBUF;
};

template <int NQ, int G4C>
__global__ __launch_bounds__(512) void kp_gram6_kernel(Gram3Args a) {
  using L = G6Layout<G4C>;
  constexpr int CS = L::CS, WS = L::WS, YO = L::YO, ZO = L::ZO, RS = L::RS, PSIBUF = L::PSIBUF, PSI0 = L::PSI0;
  (void)CS;
  extern __shared__ __align__(16) double sm[];
  const BasisDev& b = a.b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order (as kp_gram3): the workgroups of one snapshot split share an XCD's L2
  const int per_xcd = gridDim.x / 8;
  const int logical = (int)blockIdx.x < per_xcd * 8 ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int super = logical % a.nsuper, split = logical / a.nsuper;
  const int job = super * 8 + wave;
  const int nzm = b.nzeta + b.m, nrawrows = 2 * nzm, D = a.D;
  const int CID = nrawrows * D;                       // the constant 1 (0 past Ns); the quadratic weights follow it
  const int N = b.nfull;

  // ---- MFMA operand offsets (doubles, Psi buffer 0, k-step 0) ----
  const uint32_t* jd = a.desc + (size_t)job * (1 + NQ);
  const uint32_t jh = jd[0];
  const int lrow = (lane >> 4) * RS, blk = (lane >> 2) & 3, lc = lane & 3;
  const int ao0 = PSI0 + lrow + (4 * (int)(jh & 255u) + lc) * WS;
  const int ao1 = PSI0 + lrow + (4 * (int)((jh >> 8) & 255u) + lc) * WS;
  const int qs = __builtin_amdgcn_readfirstlane((int)((jh >> 16) & 255u));
  int bo[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int g = (int)((jd[1 + q] >> (8 * blk)) & 255u);
    bo[q] = PSI0 + lrow + (g < a.G4 ? (4 * g + lc) * WS : g < 2 * a.G4 ? YO + 4 * (g - a.G4) + lc : ZO + lc);
  }
  double acc[NQ][NWT6];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NWT6; ++w) acc[q][w] = 0.0;

  for (int e = tid; e < L::LDS_DOUBLES; e += 512) sm[e] = 0.0;        // padding columns and the zero group stay zero

  // ---- lift items: EVERY thread lifts one x item (column, snapshot pair: psi and its nine weighted copies) and one y item,
  // item = tid mod 4 N - the threads beyond 4 N repeat the first items (identical values to identical addresses), so that the
  // MFMA loop holds no exec-mask branch (divergent branches cut it into small scheduling regions and cost register copies) ----
  const int nxi = 4 * N;
  const int item = tid % nxi;
  const int it_ch = item / N, it_col = item - it_ch * N;
  int fa[2][NF6];                                     // [side][factor]
  {
    const uint32_t r = a.recipes[it_col];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
      for (int f = 0; f < NF6; ++f) {
        const int id = (int)((r >> (8 * f)) & 255u);
        fa[sd][f] = (id == 255 ? CID : sd * nzm * D + id) * PST6 + 2 * it_ch;
      }
  }
  // destination: rows 2 ch and 2 ch + 1; x -> column it_col (weight w adds w CS), y -> YO + it_col
  const int wdst = PSI0 + (2 * it_ch) * RS + it_col * WS;          // x: [column][weight]; y: YO + column (formed below)
  const int ydst = PSI0 + (2 * it_ch) * RS + YO + it_col;
  // weight w >= 1 of the item's snapshot pair: table entries CID + 1 + k (one base address, immediate offsets; 16-byte reads
  // of the two snapshots).  k = 0..5: ut_a ut_b, 1 <= a <= b <= 3 in that order; k = 6..8: copies of ut_1..ut_3
  const int wbase = (CID + 1) * PST6 + 2 * it_ch;
  auto wk = [](int w) constexpr { return w <= 3 ? 5 + w : w - 4; };     // weight index (x <= y order: 1, u1, u2, u3, u1u1, ...) -> k

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT6 - 1) / KT6;
  const int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // ---- raw loader: thread t < 2 (nzeta + m) KT6 = (row, snapshot); rows [alpha (nzeta) u (m) | beta (nzeta) u (m)] ----
  struct RawRegs { double v, e0, e1; bool ok; };
  const bool ld_on = tid < nrawrows * KT6;            // (the others load what thread tid mod (rows KT6) loads and store nothing)
  const int ld_s = tid & (KT6 - 1);
  const int ld_r = (tid % (nrawrows * KT6)) / KT6, ld_rr = ld_r % nzm;
  const int ld_ua = (ld_r < nzm && ld_rr >= b.nzeta) ? ld_rr - b.nzeta : -1;      // input index of an alpha-side input row
  const double* ld_ptr = (ld_rr < b.nzeta ? ((ld_r < nzm ? a.alpha : a.beta) + (int64_t)ld_rr * a.Ns) : (a.u + (int64_t)(ld_rr - b.nzeta) * a.Ns)) +
                         kt0 * KT6 + ld_s;
  const int ld_e0 = ld_ua >= 0 && ld_ua + 1 < 3 ? (int)a.Ns : 0, ld_e1 = ld_ua >= 0 && ld_ua + 2 < 3 ? 2 * (int)a.Ns : 0;   // the other inputs' columns
  const int ld_dst = ld_r * D * PST6 + ld_s;
  const int ld_wk = ld_ua == 0 ? 0 : ld_ua == 1 ? 3 : 5;                                   // first quadratic weight this thread writes
  int ld_rem = (int)max((int64_t)-1000000, min((int64_t)1 << 30, a.Ns - (kt0 * KT6 + ld_s)));
  auto load_raw = [&]() __attribute__((always_inline)) -> RawRegs {
    RawRegs x;
    x.ok = ld_rem > 0;
    x.v = *ld_ptr;                                     // branch-free: every thread loads (offsets 0 where there is nothing else to fetch)
    x.e0 = ld_ptr[ld_e0];
    x.e1 = ld_ptr[ld_e1];
    ld_ptr += KT6;
    ld_rem -= KT6;
    return x;
  };
  auto store_raw = [&](auto buf_c, const RawRegs& x) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    if (ld_on) {
      double* dst = sm + BUF * POWBUF6 + ld_dst;
      const double xv = x.ok ? x.v : 0.0;             // snapshots past Ns: every power is 0 (the tail mask)
      double p = xv;
      for (int e = 0; e < D; ++e) {
        dst[e * PST6] = p;
        p *= xv;
      }
      if (ld_ua >= 0) {                               // ut_a ut_b, b >= a >= 1 (Ksysid.m:510-511: the Kronecker weights), and ut_a again
        double* wq = sm + BUF * POWBUF6 + (CID + 1) * PST6 + ld_s;
        wq[ld_wk * PST6] = xv * xv;
        if (ld_ua < 2) wq[(ld_wk + 1) * PST6] = xv * x.e0;
        if (ld_ua < 1) wq[(ld_wk + 2) * PST6] = xv * x.e1;
        wq[(6 + ld_ua) * PST6] = xv;
      }
    }
    if (tid < KT6) sm[BUF * POWBUF6 + CID * PST6 + tid] = x.ok ? 1.0 : 0.0;
  };

  // ---- the lift in sub-steps, issued between the MFMAs: 0 reads two factors of each side, 1 multiplies them and reads the
  // third, 2 writes psi_x, psi_y and reads weight 1, w + 2 writes weighted copy w of psi_x and reads weight w + 1.  `psb` =
  // this thread's destination in the Psi buffer being filled (formed once per tile: the two Psi buffers span 121 KB, beyond
  // the 64 KB an immediate DS offset reaches) ----
  double2 lxa, lxb, lya, lyb, psi, psy, wv, wlo;
  auto lift_sub = [&](int sub, auto buf_c, int psb, int ysb) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const double* pw = sm + BUF * POWBUF6;
    if (sub == 0) {
      lxa = *reinterpret_cast<const double2*>(&pw[fa[0][0]]);
      lxb = *reinterpret_cast<const double2*>(&pw[fa[0][1]]);
      lya = *reinterpret_cast<const double2*>(&pw[fa[1][0]]);
      lyb = *reinterpret_cast<const double2*>(&pw[fa[1][1]]);
    } else if (sub == 1) {
      psi.x = lxa.x * lxb.x; psi.y = lxa.y * lxb.y;
      psy.x = lya.x * lyb.x; psy.y = lya.y * lyb.y;
      lxa = *reinterpret_cast<const double2*>(&pw[fa[0][2]]);
      lya = *reinterpret_cast<const double2*>(&pw[fa[1][2]]);
    } else if (sub == 2) {
      psi.x *= lxa.x; psi.y *= lxa.y;
      sm[ysb] = psy.x * lya.x;
      sm[ysb + RS] = psy.y * lya.y;
      wv = *reinterpret_cast<const double2*>(&pw[wbase + wk(1) * PST6]);
      wlo = psi;                                        // weight 0 rides with weight 1 (one 16-byte write per snapshot)
    } else if (sub <= NWT6 + 1) {
      const int w = sub - 2;                          // 1 .. 9
      const double2 cur = {psi.x * wv.x, psi.y * wv.y};
      if (w & 1) {                                      // (w - 1, w): one 16-byte write per snapshot
        *reinterpret_cast<double2*>(&sm[psb + (w - 1)]) = double2{wlo.x, cur.x};
        *reinterpret_cast<double2*>(&sm[psb + RS + (w - 1)]) = double2{wlo.y, cur.y};
      } else {
        wlo = cur;
      }
      if (w + 1 < NWT6) wv = *reinterpret_cast<const double2*>(&pw[wbase + wk(w + 1 < NWT6 ? w + 1 : w) * PST6]);
    }
  };
  constexpr int NSUB = NWT6 + 2;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  __syncthreads();
  store_raw(B0{}, load_raw());
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NSUB; ++i) lift_sub(i, B0{}, wdst, ydst);
  store_raw(B1{}, load_raw());
  __syncthreads();

  constexpr int NSTEP = (KT6 / 4) * NQ;               // quad steps (NWT6 MFMAs each) per tile
  constexpr int PF = 3;
  static_assert(NSTEP >= NSUB, "the lift's sub-steps ride on the quad steps");

  auto tile = [&](auto cur_c, auto qs_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int QS = decltype(qs_c)::value;
    using NXT = std::integral_constant<int, 1 - CUR>;
    // per-tile base addresses (one VALU add each): the compile-time part of every operand offset below - k-step, weight block -
    // then stays inside the 64 KB an immediate DS offset reaches
    int pcur = CUR * PSIBUF, pnxt = (1 - CUR) * PSIBUF;
    asm volatile("" : "+v"(pcur), "+v"(pnxt));
    const int aA = ao0 + pcur, aB = ao1 + pcur;
    int bq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bq[q] = bo[q] + pcur;
    const int psb = wdst + pnxt, ysb = ydst + pnxt;
    RawRegs rawreg;
    double bvs[NSTEP];
    double2 aw2[NWT6 / 2];                             // the A set: weights (2 j, 2 j + 1) of this lane's column, 16-byte reads
#pragma unroll
    for (int j = 0; j < NWT6 / 2; ++j) aw2[j] = *reinterpret_cast<const double2*>(&sm[aA + 2 * j]);
#pragma unroll
    for (int i = 0; i < PF; ++i) bvs[i] = sm[(i / NQ) * 4 * RS + bq[i % NQ]];
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int kk = step / NQ, q = step % NQ;
      if (step == 1) rawreg = load_raw();
      if (step + PF < NSTEP) bvs[step + PF] = sm[((step + PF) / NQ) * 4 * RS + bq[(step + PF) % NQ]];
      const double bv = bvs[step];
      // the A set the NEXT quad step needs, when it differs: fetched weight by weight right behind the MFMA that last
      // reads the register (nine MFMAs pass before the first of them is used)
      const bool sw1 = QS < NQ && q == QS - 1;                                   // next: group a1 of this k-step
      const bool sw0 = q == NQ - 1 && kk + 1 < KT6 / 4;                          // next: group a0 of the next k-step
#pragma unroll
      for (int j = 0; j < NWT6 / 2; ++j) {
        acc[q][2 * j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw2[j].x, bv, acc[q][2 * j], 0, 0, 0);
        acc[q][2 * j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw2[j].y, bv, acc[q][2 * j + 1], 0, 0, 0);
        if (sw1) aw2[j] = *reinterpret_cast<const double2*>(&sm[kk * 4 * RS + aB + 2 * j]);
        else if (sw0) aw2[j] = *reinterpret_cast<const double2*>(&sm[(kk + 1) * 4 * RS + aA + 2 * j]);
      }
      if (step < NSUB) lift_sub(step, NXT{}, psb, ysb);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_raw(cur_c, rawreg);
    __syncthreads();
  };
  auto run_tiles = [&](auto qs_c) __attribute__((always_inline)) {
    int t = 0;
    for (; t + 1 < nkt; t += 2) {
      tile(B0{}, qs_c);
      tile(B1{}, qs_c);
    }
    if (t < nkt) tile(B0{}, qs_c);
  };
  switch (qs) {
    case 1: run_tiles(std::integral_constant<int, 1>{}); break;
    case 2: run_tiles(std::integral_constant<int, (NQ >= 2 ? 2 : NQ)>{}); break;
    case 3: run_tiles(std::integral_constant<int, (NQ >= 3 ? 3 : NQ)>{}); break;
    case 4: run_tiles(std::integral_constant<int, (NQ >= 4 ? 4 : NQ)>{}); break;
    case 5: run_tiles(std::integral_constant<int, (NQ >= 5 ? 5 : NQ)>{}); break;
    case 6: run_tiles(std::integral_constant<int, (NQ >= 6 ? 6 : NQ)>{}); break;
    default: run_tiles(std::integral_constant<int, NQ>{}); break;
  }

  double* dst = a.part + (((size_t)split * a.njobs + job) * NQ) * NWT6 * 64 + lane;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NWT6; ++w) dst[(q * NWT6 + w) * 64] = acc[q][w];
}

template <int NQ, int G4C>
static hipError_t launch6(const Gram3Args& a, int grid, hipStream_t st) {
  static KpLdsCache lds_cache;
  const size_t lds = (size_t)G6Layout<G4C>::LDS_DOUBLES * sizeof(double);
  hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_gram6_kernel<NQ, G4C>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((kp_gram6_kernel<NQ, G4C>), dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

bool kp_gram6_serves(int nq, int G4) { return nq == 7 && G4 == 21; }

hipError_t kp_gram6_launch_kernel(const Gram3Args& a, int nq, int grid, hipStream_t st) {
  if (nq == 7 && a.G4 == 21) return launch6<7, 21>(a, grid, st);
  return hipErrorInvalidValue;
}

// Fused lift + Gram kernel for BILINEAR monomial dictionaries, exploiting the Kronecker
// structure of the bilinear rows  Psi = psi (x) [1; u]  (Ksysid.m:510-511, 1594-1604):
//
//   Px'Px = sum_k (ut ut') (x) (psi_x psi_x')     Px'Py = sum_k (ut ut') (x) (psi_x psi_y')
//
// with ut = [1; u_k].  Only the (m+1)(m+2)/2 distinct weights w_ab = ut_a ut_b are needed, and
// only psi (N columns per side, not N(m+1)) is lifted into LDS.  For every weight the kernel
// accumulates  S_w = sum_k w psi_x psi_x'  (symmetric: circulant half) and T_w = sum_k w psi_x psi_y'
// in 4x4 blocks with v_mfma_f64_4x4x4_4b_f64:  A = one 4-column group of psi_x scaled by the
// weight (one v_mul per group and weight), B = four 4-column groups of [psi_x | psi_y]
// (one LDS read feeds all weights).  Executed flops are 62.5 % of the dense W x W products the
// reference forms (W = N(m+1)); the result is the same matrices G = Px'Px, C = Px'Py.
//
// Design rule this kernel is built around (tools/mfma_coissue_bench.hip, profiles/r01_coissue.txt):
// on gfx950 NO VALU instruction (integer or floating point) overlaps with the FP64 MFMA stream of
// its SIMD -- each costs ~5.5 cycles of MFMA time even with two waves per SIMD -- while LDS reads
// are free up to about one per MFMA.  So the LDS layout is a compile-time constant (every operand
// read is `ds_read_b64 v, vaddr offset:imm`, the tile loop is unrolled by two so that the buffer
// index is an immediate too), the weights ut_a ut_b come from LDS (written once per snapshot by the
// lift) and the tail mask lives in the power table's constant entry.
//
// Round 6 (0.396 -> 0.358 ms per 1e5 pairs at W = 336; what each step gave is in DESIGN.md 3.1 "Round 6", measured with the
// timing-only builds of tools/build_abl3.sh: MFMAs and their operand reads alone take 0.305 ms):
//  * m = 3: TWO weights per weighted A operand (TUP below) - 5 weight multiplies per A group and k-step instead of 9;
//  * the power table of the tile after next is built INSIDE the MFMA loop by every thread, branch-free (x .. x^4 at a
//    compile-time entry stride, raw value loaded a tile ahead; INL), not behind it with a run-time loop and exec masks;
//  * the Kronecker weights are lifted like columns (no separate step at the head of the tile);
//  * LDS traffic is NOT free at this instruction mix: table rows at a stride of 62 doubles (bank conflicts of the lift's
//    16-byte reads 77 -> 57 cycles per chunk), a lane's five tuple weights contiguous (two 16-byte reads and one 8-byte read
//    instead of the ds_read2_b64 pairs the compiler formed at half the LDS rate);
//  * the lift's (item, snapshot pair) jobs dealt over 3 x 256 slots: three lift steps per tile instead of four;
//  * the first three raw tiles requested in front of the one-time LDS setup.
// Measured and NOT kept: lift stores as two ds_write_b64 at immediate offsets instead of ds_write2_b64 + v_add (slower), the
// k-step's weights kept in registers across the wave's two A groups (slower), 7 quads per wave (147 spilled registers), the
// items with three real factors gathered in the last wave so that the others skip a read and a multiply (branches and the
// scattered stores cost more), B operands 2 / 4 steps ahead and the raw load behind other steps (no difference), cbsz / abid
// (they do not broadcast blocks on v_mfma_f64_4x4x4_4b: tools/mfma_bcast_probe.hip).
//
// Replaces the per-row lift loop of Ksysid.get_Koopman (Ksysid.m:1030-1065) and the products
// PxTPx, PxTPy (Ksysid.m:1114,1125) for model_type 'bilinear'.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kp_internal.h"

#ifndef KP_ABL3
#define KP_ABL3 0
#endif
#ifndef KP_G3_ASMW
#define KP_G3_ASMW 0    // 1: lift stores as 8-byte stores at immediate offsets (inline asm) instead of the compiler's ds_write2_b64 + v_add.
                        // Measured SLOWER (0.3718 against 0.3659 ms): two store instructions cost the wave more than a v_add and one
#endif
#ifndef KP_G3_NLS
#define KP_G3_NLS 3     // lift steps per tile: 3 = (item, snapshot pair) jobs dealt over 3 x 256 slots; 4 = item per thread, one step per pair
#endif
#ifndef KP_G3_KEEPWT
#define KP_G3_KEEPWT 0   // 1: the k-step's weights stay in registers for the wave's second A group instead of being read again
#endif
#ifndef KP_G3_TBASM
#define KP_G3_TBASM 1   // ... the table stores that way are faster (0.3640 against 0.3659 ms; four entries 80 bytes apart: no ds_write2 pairs them without an add)
#endif
#ifndef KP_RAW_STEP
#define KP_RAW_STEP 1   // MFMA step of a tile behind which the raw loads of the tile after next are issued
#endif
#define KT3 8     // snapshots per LDS tile (two k-steps)
#define NF3 3     // single-variable powers per column (recipes with 4 factors use the general monomial kernel)
// LDS row (doubles): psi_x [0,96) | psi_y [96,192) | zero group [192,196) | weights [196,208) | scratch [208,240)
#define RS3 240   // = 16 mod 32: the four k-rows of an MFMA operand read hit disjoint banks
#define YOFF3 96
#define ZOFF3 192
#define WOFF3 196
#define SOFF3 208
#define NIDMAX3 144                 // power-table entries per snapshot (2 x 16 rows x 4 powers + the constant row's 4 fit)
#define PST3 10                     // doubles per power-table entry: KT3 snapshots + 2 of padding, so that the 16-byte reads of
                                    // 64 lanes with arbitrary ids spread over all banks (a stride of 8 puts them on 4 groups)
#define POWBUF3 (PST3 * NIDMAX3)    // doubles per power-table buffer
#define PSIBUF3 (KT3 * RS3)         // doubles per Psi buffer
#define PSI03 (2 * POWBUF3)         // LDS: pow[2] | psi[2]
#define LDS3_DOUBLES (PSI03 + 2 * PSIBUF3)
// dim_red dictionaries: the projection matrix lives behind the Psi buffers as [column tile of 16 PCs][full column][16]
// (row stride 16 doubles: the two full columns a 32-lane group reads sit on disjoint banks), at most 32 PCs
#define PCS03 LDS3_DOUBLES
#define PCSMAX3 32
#define LDS3_PCS_DOUBLES (2 * YOFF3 * 16)
// fourier / gaussian dictionaries (EXT): the per-variable table also holds cos / sin(2 pi j x) behind the powers, and one
// entry per (side, gaussian centre) behind the constant; the centres live in LDS where the projection matrix would
#define GC03 LDS3_DOUBLES
#define GAUSSMAX3 32                // centres (2 sides x 32 x KT3 values per tile = two items per thread)
#ifndef KP_PCS_REGS
#define KP_PCS_REGS 24                 // k-steps of the pcs projection whose matrix operand stays in registers (<= 96 full columns)
#endif
#define GNZMAX3 8                   // state variables of a gaussian dictionary
#define LDS3_GAUSS_DOUBLES (GAUSSMAX3 * GNZMAX3)

#include "kp_gram3_args.h"

// the constant row of the in-loop power table "loads" its 1.0 like the raw rows load their values (kp_gram3_kernel, INL); not
// `const`: a pointer that may be this or a kernel argument must stay a GLOBAL pointer (constant address space: flat loads)
__device__ double kp_gram3_ones[KT3] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};

// PCS: econ lift through a projection matrix (dim_red dictionaries)
// EXT: fourier (def_fourierLift, Ksysid.m:694-731) and gaussian (def_gaussianLift, :790-817) blocks through the same table
// TUP (round 6; m = 3 and m = 2: an even number of weights): the four blocks of a weighted A operand carry TWO weights - tuple p = (w_2p, w_2p, w_2p+1, w_2p+1) -
// against B operands (g0, g1, g0, g1) and (g2, g3, g2, g3) of the quad: the same 10 MFMAs per quad and k-step produce the same
// 40 (weight, group) blocks, bit for bit, with 5 weight multiplies per A group and k-step instead of 9 (nothing on the vector
// pipe overlaps the f64 MFMA stream: each costs ~5.5 cycles of it) and 10 operand registers less; the price is a second
// operand read per quad.  The weights sit in the 12 weight entries of a Psi row in tuple order (wslot), w_0 = 1 stored.
template <int NQ, int BM, bool PCS, bool EXT = false, bool PRE = false, bool TUP = false>
__global__ __launch_bounds__(256, 2) void kp_gram3_kernel(Gram3Args a) {
  constexpr int NWT = (BM + 1) * (BM + 2) / 2;
  static_assert(!TUP || NWT % 2 == 0, "paired weights: an even number of weights (m = 3: 10 = 5 tuples, m = 2: 6 = 3 tuples)");
  // TUP: weight w sits in slot 6 (w & 1) + (w >> 1) of the 12 weight entries of a Psi row (w_0 = 1 stored in slot 0): the five
  // weights a lane multiplies in - w_{2p+h}, p = 0..4, h = blk >> 1 - are contiguous and 16-byte aligned (two ds_read_b128 and a
  // ds_read_b64; as 8-byte reads at stride 2 the compiler paired them into ds_read2_b64, half the LDS rate)
  auto wslot = [](int w) { return TUP ? 6 * (w & 1) + (w >> 1) : w - 1; };
  extern __shared__ __align__(16) double sm[];
  const BasisDev& b = a.b;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // XCD-aware order: workgroups b and b + 8 share an XCD (round-robin dispatch) and its L2, so consecutive LOGICAL ids -
  // the nsuper workgroups of one snapshot split, which read the same raw tiles - are dealt to one XCD: the tile is
  // fetched from HBM once per split instead of once per workgroup (a speed matter only; any mapping is correct)
  const int per_xcd = gridDim.x / 8;
  const int logical = (int)blockIdx.x < per_xcd * 8 ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int super = logical % a.nsuper;
  const int split = logical / a.nsuper;
  const int job = super * 4 + wave;
  const int nzm = b.nzeta + b.m;
  const int nrawrows = 2 * nzm;
  const int D = a.D;
  // INL (monomial dictionaries lifted in the kernel): the power table of the tile after next is built INSIDE the MFMA loop -
  // branch-free, four powers per raw value at a compile-time entry stride - instead of behind it (round 6: the end-of-tile
  // build with its run-time power loop, its exec-mask branches and its wait for the raw load cost 29 of 389 us, timing-only
  // build KP_ABL3=8).  kp_gram3_applicable admits monomial dictionaries with powers <= 4 only (others: kp_gram2 / general kernel).
  constexpr bool INL = !EXT && !PRE;
  // Table layout (doubles): row r (raw rows, then the constant's row) at r * RB, its entry e at + e * PST3, KT3 snapshots each.
  // EXT keeps the dense id order of its recipes (RB = D entries).  INL: 4 entries per row and, when the table has the room, a row
  // stride of 62 doubles - the lift's 16-byte reads of 64 columns with unrelated ids then spread over the banks as well as the
  // 3-entry rows of rounds 1-5 did (57 LDS cycles per chunk of a workgroup for the poly-3 dictionary on 6 states against 77 at
  // the dense stride 40; tools/lds_layout_sim.py).
  const int RB = !INL ? D * PST3 : ((nrawrows + 1) * 62 + 4 * PST3 <= POWBUF3 ? 62 : 4 * PST3);
  const int CA = nrawrows * RB;                    // address of the constant 1 (0 for snapshots past Ns: the tail mask)

  // ---- MFMA operand offsets (doubles, Psi buffer 0): row (lane>>4) of k-step 0 ----
  const uint32_t* jd = a.desc + (size_t)job * (1 + NQ);
  const uint32_t jh = jd[0];
  const int lrow = (lane >> 4) * RS3, blk = (lane >> 2) & 3, lc = lane & 3;
  const int ao0 = PSI03 + lrow + 4 * (int)(jh & 255u) + lc;      // A group replicated over the 4 blocks
  const int ao1 = PSI03 + lrow + 4 * (int)((jh >> 8) & 255u) + lc;
  const int qs = __builtin_amdgcn_readfirstlane((int)((jh >> 16) & 255u));   // wave-uniform
  int bo[NQ];
  int bo2[TUP ? NQ : 1];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    // TUP: block b of the first B operand reads group (b & 1) of the quad, of the second one group 2 + (b & 1)
    const int g = (int)((jd[1 + q] >> (8 * (TUP ? (blk & 1) : blk))) & 255u);
    bo[q] = PSI03 + lrow + (g < a.G4 ? 4 * g : g < 2 * a.G4 ? YOFF3 + 4 * (g - a.G4) : ZOFF3) + lc;
    if (TUP) {
      const int g2 = (int)((jd[1 + q] >> (8 * (2 + (blk & 1)))) & 255u);
      bo2[q] = PSI03 + lrow + (g2 < a.G4 ? 4 * g2 : g2 < 2 * a.G4 ? YOFF3 + 4 * (g2 - a.G4) : ZOFF3) + lc;
    }
  }
  // TUP: this lane's weight of tuple p is entry 2 p + (blk >> 1)
  const int wo = PSI03 + lrow + WOFF3 + (TUP ? 6 * (blk >> 1) : 0);

  double acc[NQ][NWT];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NWT; ++w) acc[q][w] = 0.0;

  // ---- lifting thread constants: thread = one (side, column), all KT3 snapshots of the tile ----
  // power table layout [id][snapshot]: the KT3 values of one entry are contiguous (16-byte reads of snapshot pairs)
  // The NWT - 1 Kronecker weights ut_x ut_y (pair index w + 1 in (x <= y) order) are lifted like columns: thread 2 nfull + w
  // multiplies the table entries of u_x and u_y (and the constant) for all KT3 snapshots and writes entry WOFF3 + WSH + w of the
  // Psi rows.  (Until round 6 a separate step at the head of every tile formed them: an LDS round trip and a branch in front
  // of the tile's first operand reads.)
  // item -> its three table addresses and its destination column in a Psi row
  auto item_setup = [&](int item, int (&fa_)[NF3], int& woff_) __attribute__((always_inline)) {
    const bool lvalid = item >= 0 && item < 2 * b.nfull;
    const int lside = (lvalid && item >= b.nfull) ? 1 : 0;
    const int lcol = lvalid ? item - lside * b.nfull : 0;
    const int lw = item - 2 * b.nfull;             // weight item: 0 <= lw < NWT - 1
    const bool lwt = lw >= 0 && lw < NWT - 1;
    const uint32_t r = lvalid ? a.recipes[lcol] : 0xffffffffu;
#pragma unroll
    for (int f = 0; f < NF3; ++f) {
      const int id = (int)((r >> (8 * f)) & 255u);
      fa_[f] = id == 255 ? CA : (EXT && id >= 128) ? CA + (1 + lside * a.ng + (id - 128)) * PST3 : (lside * nzm + id / D) * RB + (id % D) * PST3;
    }
    if (lwt) {
      int cnt = 0;
      for (int x = 0; x <= BM; ++x)
        for (int y = x; y <= BM; ++y) {
          if (cnt == lw + 1) {
            if (x > 0) fa_[0] = (b.nzeta + x - 1) * RB;
            if (y > 0) fa_[1] = (b.nzeta + y - 1) * RB;
          }
          ++cnt;
        }
    }
    woff_ = PSI03 + (lvalid ? lside * YOFF3 + lcol : lwt ? WOFF3 + wslot(lw + 1) : SOFF3 + (tid & 31));
  };
#if KP_G3_NLS == 3
  // The 4 (2 nfull + NWT - 1) (item, snapshot pair) jobs of a tile dealt over 3 x 256 slots: three lift steps per tile instead of
  // four (item = thread, one step per snapshot pair: 256 lanes for 177 items at N = 84).  A slot's snapshot pair is part of its
  // addresses, not an immediate: 12 address registers per thread instead of 4.  kp_gram3_applicable: 2 nfull + NWT - 1 <= 192.
  constexpr int NLS = 3;
  int fs[NLS][NF3], ws[NLS];
  {
    const int nitems = 2 * b.nfull + NWT - 1;
#pragma unroll
    for (int j = 0; j < NLS; ++j) {
      const int e = tid + 256 * j;
      const bool on = e < 4 * nitems;
      const int ch = on ? e / nitems : 0;
      item_setup(on ? e - ch * nitems : -1, fs[j], ws[j]);
#pragma unroll
      for (int f = 0; f < NF3; ++f) fs[j][f] += 2 * ch;
      ws[j] += 2 * ch * RS3;
    }
  }
#else
  int fa[NF3];
  int woff;
  item_setup(tid, fa, woff);
#endif

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT3 - 1) / KT3;
  const int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // ---- raw loader; rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)] ----
  // value e = tid + j*256 of a tile -> (row e / KT3, snapshot e % KT3); tiles are loaded in order, so every
  // thread keeps a running pointer and a running count of the snapshots left in its row
  constexpr int LR = 1;                             // 2 (nzeta + m) KT3 <= 256 raw values per tile (kp_gram3_applicable): one per thread
  const int nld = (nrawrows * KT3 + 255) / 256;     // wave-uniform number of active j
  constexpr int LG = EXT ? 2 : 0;                   // gaussian items (side, centre, snapshot) per thread
  struct RawRegs { double v[LR]; bool ok; };
  bool ld_on[LR];
  bool ld_isz[LR];                                  // row of a state variable (EXT: gets the trigonometric entries)
  const double* ld_ptr[LR];
  const int ld_s = tid & (KT3 - 1);                 // the same snapshot for every j (256 is a multiple of KT3)
  const int ld_dst0 = (tid / KT3) * RB + ld_s;
  int ld_rem = (int)max((int64_t)-1000000, min((int64_t)1 << 30, a.Ns - (kt0 * KT3 + ld_s)));
#pragma unroll
  for (int j = 0; j < LR; ++j) {
    const int e = tid + j * 256;
    ld_on[j] = e < nrawrows * KT3;
    const int r = ld_on[j] ? e / KT3 : 0;
    const int rr = r % nzm;
    const double* src = rr < b.nzeta ? ((r < nzm ? a.alpha : a.beta) + (int64_t)rr * a.Ns) : (a.u + (int64_t)(rr - b.nzeta) * a.Ns);
    ld_ptr[j] = src + kt0 * KT3 + ld_s;
    ld_isz[j] = ld_on[j] && rr < b.nzeta;
  }
  // EXT: gaussian item q of this thread = (side, centre) for snapshot ld_s; it reads the nzeta raw values of its side itself
  // (the same cache lines the row loaders fetch) and writes exp(-|zeta - c|^2) into the table entry of its centre
  bool g_on[LG > 0 ? LG : 1];
  const double* g_ptr[LG > 0 ? LG : 1];
  int g_dst[LG > 0 ? LG : 1], g_cen[LG > 0 ? LG : 1];
  if (EXT) {
#pragma unroll
    for (int q = 0; q < LG; ++q) {
      const int gi = tid + q * 256;
      g_on[q] = gi < 2 * a.ng * KT3;
      const int gc = g_on[q] ? (gi / KT3) % a.ng : 0, gside = g_on[q] ? gi / (KT3 * a.ng) : 0;
      g_ptr[q] = (gside ? a.beta : a.alpha) + kt0 * KT3 + ld_s;
      g_dst[q] = CA + (1 + gside * a.ng + gc) * PST3 + ld_s;
      g_cen[q] = GC03 + gc * b.nzeta;
    }
  }
  auto load_raw = [&]() __attribute__((always_inline)) -> RawRegs {                // next tile of this workgroup's range
    RawRegs x;
    x.ok = ld_rem > 0;
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      x.v[j] = 0.0;
      if (j < nld) {
        x.v[j] = *ld_ptr[j];        // no arithmetic on the loaded value here: its first use (store_raw, at the END of a tile) is
                                    // where the wave waits for the load - the tail mask is applied there
        ld_ptr[j] += KT3;
      }
    }
    ld_rem -= KT3;
    return x;
  };
  auto store_raw = [&](auto buf_c, const RawRegs& x) __attribute__((always_inline)) {   // powers x^1..x^D, and the constant / tail-mask entry
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      if (j < nld && ld_on[j]) {
        double* dst = sm + BUF * POWBUF3 + ld_dst0 + j * 32 * RB;
        const double xv = x.ok ? x.v[j] : 0.0;        // snapshots past Ns: every power is 0 (the tail mask)
        double p = xv;
        const int Dp = EXT ? a.Dp : D;
        for (int e = 0; e < Dp; ++e) {
          dst[e * PST3] = p;
          p *= xv;
        }
        if (EXT && a.df > 0 && ld_isz[j]) {
          // cos / sin(2 pi j x), j = 1..df, by the angle-addition recurrence; 0 past Ns (the tail mask: cos 0 = 1 would count)
          double s1, c1;
          sincospi(2.0 * xv, &s1, &c1);               // exact range reduction (the argument is 2 x, not 2 pi x)
          double cj = x.ok ? c1 : 0.0, sj = x.ok ? s1 : 0.0, cm = x.ok ? 1.0 : 0.0, sm1 = 0.0;
          for (int h = 0; h < a.df; ++h) {
            dst[(Dp + 2 * h) * PST3] = cj;
            dst[(Dp + 2 * h + 1) * PST3] = sj;
            const double cn = 2.0 * c1 * cj - cm, sn = 2.0 * c1 * sj - sm1;
            cm = cj; sm1 = sj; cj = cn; sj = sn;
          }
        }
      }
    }
    if (EXT) {
      // the raw values of the item's snapshot are fetched HERE, not with the row loads at the start of the tile: the row
      // loaders have pulled the same cache lines a tile ago, and 16 doubles held across the tile's MFMA loop would spill
#pragma unroll
      for (int q = 0; q < LG; ++q)
        if (g_on[q]) {
          double r2 = 0.0;
          for (int i = 0; i < b.nzeta; ++i) {
            const double dlt = g_ptr[q][(int64_t)i * a.Ns] - sm[g_cen[q] + i];
            r2 += dlt * dlt;
          }
          g_ptr[q] += KT3;
          sm[BUF * POWBUF3 + g_dst[q]] = x.ok ? exp(-r2) : 0.0;
        }
    }
    if (tid < KT3) sm[BUF * POWBUF3 + CA + tid] = x.ok ? 1.0 : 0.0;
  };

  // ---- INL: raw value -> register a tile ahead, table entries x, x^2, x^3, x^4 inside the next MFMA loop ----
  // EVERY thread fills a table row - no exec masking, no branch in the steady state: thread tid / KT3 = row; raw rows, then ONE
  // constant row (it "loads" 1.0 from a constant buffer with a pointer that does not advance; its first entry is the constant /
  // tail-mask address CA); the threads behind it repeat rows 0, 1, .. (the same values to the same addresses; within a wave the
  // addresses stay distinct).  Whether a raw tile lies wholly inside [0, Ns) is wave-uniform (scalar counter): only the range's
  // last tile can be partial, and only then does a thread look at its own snapshot index.
  const int tb_r = (tid / KT3) % (nrawrows + 1);
  const double* tb_ptr;
  int tb_inc = KT3;
  {
    const int rr = tb_r % nzm;
    tb_ptr = (rr < b.nzeta ? ((tb_r < nzm ? a.alpha : a.beta) + (int64_t)rr * a.Ns) : (a.u + (int64_t)(rr - b.nzeta) * a.Ns)) + kt0 * KT3 + ld_s;
    if (tb_r == nrawrows) { tb_ptr = kp_gram3_ones; tb_inc = 0; }
  }
  const uint32_t tb_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)(sm + tb_r * RB + ld_s);
  double tb_v = 0.0;                                               // the tile in flight
  int tb_tile = 0;                                                 // local index of the next tile to load (scalar)
  const int tb_full = (int)max((int64_t)0, min((int64_t)1 << 30, a.Ns / KT3 - kt0));   // local tiles wholly inside [0, Ns)
  auto load_tb = [&]() __attribute__((always_inline)) {
    tb_v = *tb_ptr;
    tb_ptr += tb_inc;
    ++tb_tile;
  };
  auto store_tb_v = [&](auto buf_c, double xv, int tile) __attribute__((always_inline)) {   // tile: local index of xv's tile (scalar)
    constexpr int BUF = decltype(buf_c)::value;
    if (__builtin_expect(tile >= tb_full, 0)) {                    // (scalar branch; at most once per workgroup)
      const int64_t s_glob = (kt0 + tile) * KT3 + ld_s;
      if (s_glob >= a.Ns) xv = 0.0;
    }
    const double x2 = xv * xv, x3 = x2 * xv, x4 = x2 * x2;
    const uint32_t tb_dst_b = tb_dst;                              // (an asm operand alone does not capture in a generic lambda)
#if !KP_G3_TBASM
    double* dstp = sm + BUF * POWBUF3 + tb_r * RB + ld_s;
    dstp[0] = xv; dstp[PST3] = x2; dstp[2 * PST3] = x3; dstp[3 * PST3] = x4;
    (void)tb_dst_b;
#else
    // (8-byte stores at immediate offsets from one address register: four entries 80 bytes apart, no ds_write2 pairs them without an add)
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(tb_dst_b), "v"(xv), "n"((BUF * POWBUF3) * 8) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(tb_dst_b), "v"(x2), "n"((BUF * POWBUF3 + PST3) * 8) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(tb_dst_b), "v"(x3), "n"((BUF * POWBUF3 + 2 * PST3) * 8) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(tb_dst_b), "v"(x4), "n"((BUF * POWBUF3 + 3 * PST3) * 8) : "memory");
#endif
  };
  auto store_tb = [&](auto buf_c) __attribute__((always_inline)) { store_tb_v(buf_c, tb_v, tb_tile - 1); };
  // the first three tiles' raw values are requested HERE, in front of the one-time LDS setup and its barriers: the setup runs in
  // the shadow of their latency (until round 6: setup, load tile 0, wait, table, barrier, lift, load tile 1, wait, ... - three
  // memory latencies in a row at the head of every workgroup, ~4 us of a 62 us launch at the arm data's 11 999 pairs)
  double tb_v0 = 0.0, tb_v1 = 0.0;
  if constexpr (INL) {
    load_tb(); tb_v0 = tb_v;
    load_tb(); tb_v1 = tb_v;
    load_tb();
  }

  // ---- one-time LDS setup: everything zero (padding columns and the zero group stay zero) ----
  for (int e = tid; e < LDS3_DOUBLES; e += 256) sm[e] = 0.0;
  if (TUP) {   // w_0 = 1 in every row of both Psi buffers (rows past Ns have psi = 0: the tail mask needs no weight)
    __syncthreads();
    if (tid < 2 * KT3) sm[PSI03 + (tid / KT3) * PSIBUF3 + (tid & (KT3 - 1)) * RS3 + WOFF3] = 1.0;
  }
  if (EXT) {   // gaussian centres (centre-major, nzeta coordinates each)
    for (int e = tid; e < a.ng * b.nzeta; e += 256) sm[GC03 + e] = a.centres[e];
  }
  if (PCS) {   // projection matrix, zero padded to (nfull4 x 32)
    const int nf4 = a.nfull4;
    for (int e = tid; e < 2 * nf4 * 16; e += 256) {
      const int ct = e / (nf4 * 16), r = e - ct * nf4 * 16, c = r >> 4, p = 16 * ct + (r & 15);
      sm[PCS03 + e] = (c < b.nfull && p < b.k_pcs) ? a.pcs[c + (size_t)p * b.nfull] : 0.0;
    }
  }

  // ---- PRE: the tile comes lifted from memory (kp_gram3_prelift_kernel): entries [psi_x | psi_y | weights] x KT3 snapshots, two 16-byte pieces per thread ----
  struct PreRegs { double2 v[2]; };
  const int pre_n2 = PRE ? KT3 * a.pre_rl / 2 : 0;   // 16-byte pieces per tile (<= 512)
  const double2* pre_ptr = PRE ? reinterpret_cast<const double2*>(a.pre + kt0 * KT3 * a.pre_rl) + tid : nullptr;
  int pre_left = nkt;
  bool pre_on[2] = {false, false};
  int pre_dst[2] = {0, 0};
  if (PRE) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * 256;
      pre_on[j] = e < pre_n2;
      // tile layout [snapshot pair][row entry c][2]: a piece = snapshots (s, s + 1) of entry c; consecutive lanes = consecutive
      // entries of one pair, so the 16 lanes of a store group write 16 different columns of two Psi rows (the round-4 layout
      // [entry][8 snapshots] put the four pairs of one entry on neighbouring lanes: rows 240 doubles apart, the same bank - 32 %
      // of this kernel's LDS cycles were conflicts)
      const int e1 = pre_on[j] ? e : 0;
      const int sp = e1 / a.pre_rl, c = e1 - sp * a.pre_rl, srow = 2 * sp;
      // (the lifted rows carry w_1 .. w_{NWT-1} and zeros up to 12 entries: the zeros go to the unused weight slots and the scratch entry behind them)
      const int cw = c - 8 * a.G4;
      const int off = c < 4 * a.G4 ? c : c < 8 * a.G4 ? YOFF3 + (c - 4 * a.G4) : WOFF3 + (!TUP ? cw : cw < NWT - 1 ? wslot(cw + 1) : (cw - (NWT - 1)) < 2 * (6 - NWT / 2) ? ((cw - (NWT - 1)) / (6 - NWT / 2)) * 6 + NWT / 2 + (cw - (NWT - 1)) % (6 - NWT / 2) : 12);
      pre_dst[j] = PSI03 + srow * RS3 + off;
    }
  }
  auto load_pre = [&]() __attribute__((always_inline)) -> PreRegs {
    PreRegs x;
    x.v[0] = make_double2(0.0, 0.0);
    x.v[1] = make_double2(0.0, 0.0);
    if (pre_left > 0) {
      if (pre_on[0]) x.v[0] = pre_ptr[0];
      if (pre_on[1]) x.v[1] = pre_ptr[256];
      pre_ptr += pre_n2;
    }
    --pre_left;
    return x;
  };
  auto store_pre = [&](auto buf_c, const PreRegs& x) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    if (pre_on[0]) { sm[BUF * PSIBUF3 + pre_dst[0]] = x.v[0].x; sm[BUF * PSIBUF3 + pre_dst[0] + RS3] = x.v[0].y; }
    if (pre_on[1]) { sm[BUF * PSIBUF3 + pre_dst[1]] = x.v[1].x; sm[BUF * PSIBUF3 + pre_dst[1] + RS3] = x.v[1].y; }
  };

  // ---- lift of a snapshot tile: power-table buffer B -> Psi buffer B, in pipelined chunks ----
#if KP_G3_NLS == 3
  constexpr int NCH = NLS;                           // lift steps per tile
  double2 lf[NF3];
  auto lift_read = [&](int j, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int f = 0; f < NF3; ++f)      // (RB, PST3, CA and the pair offset are even: 16-byte aligned, one ds_read_b128)
      lf[f] = *reinterpret_cast<const double2*>(__builtin_assume_aligned(&sm[BUF * POWBUF3 + fs[j][f]], 16));
  };
  auto lift_write = [&](int j, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    sm[BUF * PSIBUF3 + ws[j]] = (lf[0].x * lf[1].x) * lf[2].x;
    sm[BUF * PSIBUF3 + ws[j] + RS3] = (lf[0].y * lf[1].y) * lf[2].y;
  };
#else
  constexpr int NCH = KT3 / 2;                       // chunks: snapshot pairs
  double2 lf[NF3];
  auto lift_read = [&](int ch, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int f = 0; f < NF3; ++f)      // (RB, PST3 and CA are even: 16-byte aligned, one ds_read_b128)
      lf[f] = *reinterpret_cast<const double2*>(__builtin_assume_aligned(&sm[BUF * POWBUF3 + fa[f] + 2 * ch], 16));
  };
  // The two stores of a chunk go out as ds_write_b64 with 16-bit immediate offsets from ONE address register (every (buffer, row)
  // of the Psi region lies within 64 KB of it).  Left to the compiler they become a ds_write2_b64, whose 8-bit offsets reach 2 KB:
  // a v_add_u32 per chunk for the base - a vector instruction in the MFMA stream (~5.5 cycles of it) - for the same LDS time
  // (13 cycles against 2 x 6).  The compiler does not count these stores: the tile's barrier is preceded by an explicit
  // s_waitcnt lgkmcnt(0) (wait_lds_stores), and its own counted waits only become more conservative.
  uint32_t woff_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)(sm + woff);
  auto lift_write = [&](int ch, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const double v0 = (lf[0].x * lf[1].x) * lf[2].x, v1 = (lf[0].y * lf[1].y) * lf[2].y;
#define KP_LIFT_STORE(CH)                                                                                                        \
  case CH:                                                                                                                       \
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(woff_b), "v"(v0), "n"((BUF * PSIBUF3 + (2 * CH) * RS3) * 8) : "memory");     \
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(woff_b), "v"(v1), "n"((BUF * PSIBUF3 + (2 * CH + 1) * RS3) * 8) : "memory"); \
    break;
#if KP_G3_ASMW
    switch (ch) {                                       // (ch is a constant once the tile loop is unrolled; the offset must be an immediate)
      KP_LIFT_STORE(0) KP_LIFT_STORE(1) KP_LIFT_STORE(2) KP_LIFT_STORE(3)
    }
#else
    sm[BUF * PSIBUF3 + woff + (2 * ch) * RS3] = v0;
    sm[BUF * PSIBUF3 + woff + (2 * ch + 1) * RS3] = v1;
#endif
#undef KP_LIFT_STORE
    static_assert(KT3 / 2 == 4, "one case per snapshot pair of a tile");
  };
#endif
  auto wait_lds_stores = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xc07f); };   // lgkmcnt(0)
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  // ---- dim_red: econ lift of a lifted tile in place (Ksysid.m:1594-1618: [zeta ; pcs' psi_full ; 1]) ----
  // The lift above has left the FULL dictionary in the Psi buffer (columns [0, nfull) per side).  pcs' psi is a small
  // product on the matrix pipe: wave = (side, 4-snapshot group), two 4 x 16 output tiles (32 PCs), contraction over the
  // full columns.  Reads first, barrier, then the PCs overwrite columns nzeta.., the constant moves to column N - 1 and
  // the padding of the last 4-column group is cleared; the Gram MFMAs only touch columns < 4 G4 afterwards.
  // The projection's matrix operand is the same in every tile: this lane's entries of the FIRST 16 components (one per
  // k-step, KP_PCS_REGS of them at most) stay in registers for the whole kernel - the shape is bound by the LDS pipe
  // (DESIGN 6), and of the 3 operand reads per 2 MFMAs of this loop one goes away.  (Both component tiles would be 84
  // registers: measured, 75 spilled dwords in the <3,3,true> instantiation and 0.21 -> 0.25 ms; half of them: no gain.)
  constexpr int PREG = PCS ? KP_PCS_REGS : 1;
  double pcs_r[PREG];
  if (PCS) {
    __syncthreads();                                    // the staged matrix is visible
    const int bbase0 = PCS03 + (lane >> 4) * 16 + 4 * blk + lc;
#pragma unroll
    for (int kk = 0; kk < PREG; ++kk) pcs_r[kk] = kk < a.nfull4 / 4 ? sm[bbase0 + kk * 64] : 0.0;
  }
  auto project = [&](auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const int side = wave >> 1, rg = wave & 1;
    const int abase = BUF * PSIBUF3 + PSI03 + (4 * rg + lc) * RS3 + side * YOFF3 + (lane >> 4);
    const int bbase = PCS03 + (lane >> 4) * 16 + 4 * blk + lc;
    const int nf4 = a.nfull4;
    double p0 = 0.0, p1 = 0.0;
    const int bb1 = bbase + nf4 * 16;
    const int nk = nf4 / 4;
#pragma unroll
    for (int kk = 0; kk < PREG; ++kk) {
      if (kk < nk) {                                    // (uniform)
        const double av = sm[abase + 4 * kk];
        const double b1 = sm[bb1 + kk * 64];
        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, pcs_r[kk], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b1, p1, 0, 0, 0);
      }
    }
#pragma unroll 3
    for (int kk = PREG; kk < nk; ++kk) {                // dictionaries with more than 4 KP_PCS_REGS full columns
      const double av = sm[abase + 4 * kk];
      const double b0 = sm[bbase + kk * 64], b1 = sm[bb1 + kk * 64];
      p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, p0, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b1, p1, 0, 0, 0);
    }
    // the full dictionary's constant (0 past Ns: the tail mask) is read BEFORE the barrier: with nzeta + k_pcs >= nfull
    // the components written below land on its column
    const int crow = BUF * PSIBUF3 + PSI03 + (tid & (KT3 - 1)) * RS3 + ((tid / KT3) & 1) * YOFF3;
    const double one = sm[crow + b.nfull - 1];
    __syncthreads();
    {
      const int srow = BUF * PSIBUF3 + PSI03 + (4 * rg + (lane >> 4)) * RS3 + side * YOFF3;
      const int pc = 4 * blk + lc;
      if (pc < b.k_pcs) sm[srow + b.nzeta + pc] = p0;
      if (pc + 16 < b.k_pcs) sm[srow + b.nzeta + pc + 16] = p1;
    }
    if (tid < 2 * KT3) {
      sm[crow + b.N - 1] = one;
      for (int c = b.N; c < 4 * a.G4; ++c) sm[crow + c] = 0.0;
    }
    __syncthreads();
  };

  __syncthreads();
  if constexpr (PRE) {
    store_pre(B0{}, load_pre());
    __syncthreads();
  } else {
    if constexpr (INL) store_tb_v(B0{}, tb_v0, 0);
    else store_raw(B0{}, load_raw());
    wait_lds_stores();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      lift_read(i, B0{});
      lift_write(i, B0{});
    }
    if constexpr (INL) store_tb_v(B1{}, tb_v1, 1);
    else store_raw(B1{}, load_raw());
    wait_lds_stores();
    __syncthreads();
    if (PCS) project(B0{});
  }

  constexpr int NSTEP = (KT3 / 4) * NQ;                 // quad steps (NWT MFMAs each) per snapshot tile
  constexpr int SP = NSTEP / NCH > 0 ? NSTEP / NCH : 1;
  constexpr int LAG = SP / 2 > 0 ? SP / 2 : 1;
#ifndef KP_G3_PF
#define KP_G3_PF 3
#endif
  constexpr int PF = NSTEP < KP_G3_PF ? NSTEP : KP_G3_PF;   // B operands requested this many quad steps ahead

  // one snapshot tile: MFMAs on Psi buffer CUR, lift of the next tile into buffer 1-CUR, raw prefetch two ahead.
  // QS = number of leading quads that use A group a0 (wave-uniform, selected once outside the loop).
  auto tile = [&](auto cur_c, auto qs_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int QS = decltype(qs_c)::value;
    using NXT = std::integral_constant<int, 1 - CUR>;
    constexpr int PB = CUR * PSIBUF3;
    // The raw loads of the tile after next are issued INSIDE the MFMA loop (behind step KP_RAW_STEP), not here: registers the
    // compiler has spilled are reloaded at the top of the tile, and a scratch reload behind a global load waits for that
    // load too (vmcnt counts in order) - with the load up here every wave sat out its HBM latency at the start of every tile.
    RawRegs rawreg;
    PreRegs prereg;
    constexpr int NAW = TUP ? NWT / 2 : NWT;           // weighted A operands per A group (TUP: tuples of two weights)
    double bvs[NSTEP];
    double bvs2[TUP ? NSTEP : 1];                      // TUP: the quad's second B operand (groups 2, 3)
    double aw[NAW];
    double wt[NAW];
    double av1 = 0.0;
    double avn0, avn1 = 0.0;                           // raw A fragments of the NEXT k-step (prefetched)
    auto weigh = [&](double av) __attribute__((always_inline)) {
#pragma unroll
      for (int w = 0; w < NAW; ++w) {
#if KP_ABL3 == 7 || KP_ABL3 == 11
        aw[w] = av;
#else
        aw[w] = (!TUP && w == 0) ? av : av * wt[w];
#endif
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      bvs[i] = sm[PB + (i / NQ) * 4 * RS3 + bo[i % NQ]];
      if constexpr (TUP) bvs2[i] = sm[PB + (i / NQ) * 4 * RS3 + bo2[i % NQ]];
    }
    avn0 = sm[PB + ao0];
    if (QS < NQ) avn1 = sm[PB + ao1];
    // the NWT-1 weights of k-step kk, as 16-byte LDS reads (a pair each; immediate offsets, no address arithmetic);
    // TUP: this lane's weight of each of the 5 tuples, 8-byte reads at immediate offsets from its own base
    auto load_wt = [&](int kk) __attribute__((always_inline)) {
      if constexpr (TUP) {
#pragma unroll
        for (int w = 0; w + 1 < NAW; w += 2) {
          const double2 v = *reinterpret_cast<const double2*>(&sm[PB + kk * 4 * RS3 + wo + w]);
          wt[w] = v.x;
          wt[w + 1] = v.y;
        }
        wt[NAW - 1] = sm[PB + kk * 4 * RS3 + wo + NAW - 1];
      } else {
#pragma unroll
        for (int w = 0; w < NWT - 1; w += 2) {
          const double2 v = *reinterpret_cast<const double2*>(&sm[PB + kk * 4 * RS3 + wo + w]);
          wt[w + 1] = v.x;
          if (w + 2 < NWT) wt[w + 2] = v.y;
        }
      }
    };
    load_wt(0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int kk = step / NQ, q = step % NQ;
      if (step == (NSTEP > KP_RAW_STEP ? KP_RAW_STEP : 0)) {
        if constexpr (PRE) prereg = load_pre();        // the NEXT tile, lifted; stored behind the MFMA loop
        else if constexpr (INL) {                      // table of the tile after next from the value loaded a tile ago, then
          store_tb(cur_c);                             // the load for the tile behind it
          load_tb();
        } else rawreg = load_raw();
      }
      if (q == 0) {
        weigh(avn0);
        if (QS < NQ) av1 = avn1;
        if (kk + 1 < KT3 / 4) {
          avn0 = sm[PB + (kk + 1) * 4 * RS3 + ao0];
          if (QS < NQ) avn1 = sm[PB + (kk + 1) * 4 * RS3 + ao1];
        }
      }
      if (QS < NQ && q == QS) weigh(av1);
      // fetch the weights one step before they are multiplied in (LDS reads are free, registers are not)
#if !KP_G3_KEEPWT
      if (QS < NQ && QS > 1 && q == QS - 1) load_wt(kk);
#endif
      if (q == NQ - 1 && kk + 1 < KT3 / 4) load_wt(kk + 1);
#if KP_ABL3 != 6
      if (step + PF < NSTEP) {
        bvs[step + PF] = sm[PB + ((step + PF) / NQ) * 4 * RS3 + bo[(step + PF) % NQ]];
        if constexpr (TUP) bvs2[step + PF] = sm[PB + ((step + PF) / NQ) * 4 * RS3 + bo2[(step + PF) % NQ]];
      }
      const double bv = bvs[step];
      const double bv2 = TUP ? bvs2[step] : 0.0;
#else
      const double bv = bvs[step % PF];
      const double bv2 = TUP ? bvs2[step % PF] : 0.0;
#endif
#if KP_ABL3 == 10
      if constexpr (true) {
#pragma unroll
        for (int w = 0; w < NAW; ++w) asm volatile("" :: "v"(aw[w]), "v"(bv), "v"(bv2));
      } else
#endif
      if constexpr (TUP) {
        // accumulator 2 p + h: blocks b = (weight 2 p + (b >> 1), group 2 h + (b & 1) of the quad)
#pragma unroll
        for (int w = 0; w < NAW; ++w) {
          acc[q][2 * w] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw[w], bv, acc[q][2 * w], 0, 0, 0);
          acc[q][2 * w + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw[w], bv2, acc[q][2 * w + 1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int w = 0; w < NWT; ++w) acc[q][w] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw[w], bv, acc[q][w], 0, 0, 0);
      }
#if KP_ABL3 != 1 && KP_ABL3 != 6 && KP_ABL3 != 11
      // one register set for the chunk in flight: the write of chunk i precedes the read of chunk i+1
      if constexpr (!PRE) {
        if (step >= LAG && (step - LAG) % SP == 0 && (step - LAG) / SP < NCH) lift_write((step - LAG) / SP, NXT{});
        if (step % SP == 0 && step / SP < NCH) lift_read(step / SP, NXT{});
      }
#endif
      __builtin_amdgcn_sched_barrier(0);   // keep the hand-made software pipeline: no hoisting of later steps' LDS reads
    }
#if KP_ABL3 != 1 && KP_ABL3 != 6 && KP_ABL3 != 11
    if constexpr (!PRE) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (i * SP >= NSTEP) lift_read(i, NXT{});
        if (i * SP + LAG >= NSTEP) lift_write(i, NXT{});
      }
    }
#endif
#if KP_ABL3 != 8 && KP_ABL3 != 11
    if constexpr (PRE) store_pre(NXT{}, prereg);
    else if constexpr (!INL) store_raw(cur_c, rawreg);
#endif
#if KP_ABL3 != 9 && KP_ABL3 != 11
    if constexpr (!PRE) wait_lds_stores();
    __syncthreads();
#endif
    if (PCS) project(NXT{});
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
    default: run_tiles(std::integral_constant<int, NQ>{}); break;
  }

  // epilogue: [split][job][q][w][lane]
  double* dst = a.part + (((size_t)split * a.njobs + job) * NQ) * NWT * 64 + lane;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NWT; ++w) dst[(q * NWT + w) * 64] = acc[q][w];
}

// Sums the partials of one (job, quad, weight) block vector in split order and scatters the
// 4 blocks (4x4 each) into G and C.  lane l: block = (l>>2)&3, row r = l>>4, col c = l&3.
// tup (kp_gram3_kernel<.., TUP = true>): accumulator w' = 2 p + h of a quad holds, in block b, weight 2 p + (b >> 1) against
// group 2 h + (b & 1) of the quad.
// blockDim.x / 64 = 4, 8 or 16 waves share the split sum (the launcher picks by the split count: a dim_red fit at the arm data's
// size has 167 splits of 184 KB - four waves walked 42 dependent-latency loads each, 14 us for a 23 us Gram kernel), each wave
// with four loads in flight; every order is fixed: bitwise reproducible.
__global__ __launch_bounds__(1024) void kp_gram3_reduce_kernel(const double* __restrict__ part, int nsplit, int njobs, int NQ, int NWT,
                                                              int BM, const uint32_t* __restrict__ desc, int G4, int N, int W,
                                                              double* __restrict__ G, double* __restrict__ C, int tup) {
  const int idx = blockIdx.x;                 // (job*NQ + q)*NWT + w
  int w = idx % NWT;
  const int jq = idx / NWT, q = jq % NQ, job = jq / NQ;
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  __shared__ double red[16][64];
  const size_t per_split = (size_t)njobs * NQ * NWT * 64;
  const double* src = part + (size_t)idx * 64 + l;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int p = wv;
  for (; p + 3 * nwv < nsplit; p += 4 * nwv) {
    const double v0 = src[(size_t)p * per_split], v1 = src[(size_t)(p + nwv) * per_split];
    const double v2 = src[(size_t)(p + 2 * nwv) * per_split], v3 = src[(size_t)(p + 3 * nwv) * per_split];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; p < nsplit; p += nwv) s0 += src[(size_t)p * per_split];
  red[wv][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wv) return;
  double s = red[0][l];
  for (int v = 1; v < nwv; ++v) s += red[v][l];
  const uint32_t* jd = desc + (size_t)job * (1 + NQ);
  const int ga = q < (int)((jd[0] >> 16) & 255u) ? (int)(jd[0] & 255u) : (int)((jd[0] >> 8) & 255u);
  int gsel = (l >> 2) & 3;
  if (tup) {
    const int b = gsel;
    gsel = 2 * (w & 1) + (b & 1);
    w = (w & ~1) + (b >> 1);
  }
  const int gb = (int)((jd[1 + q] >> (8 * gsel)) & 255u);
  if (gb >= 2 * G4) return;                   // zero group: padding of the last quad / idle job
  // weight index -> (x, y), x <= y
  int wa = 0, wb = 0, cnt = 0;
  for (int x = 0; x <= BM; ++x)
    for (int y = x; y <= BM; ++y) {
      if (cnt == w) { wa = x; wb = y; }
      ++cnt;
    }
  const int ia = 4 * ga + (l >> 4);
  if (ia >= N) return;
  if (gb < G4) {                              // S block: psi_x' psi_x  ->  G (symmetric)
    const int jb = 4 * gb + (l & 3);
    if (jb >= N) return;
    if (ga == gb && ia > jb) return;          // diagonal block: keep the upper half, mirror below (exact symmetry)
    const size_t r1 = (size_t)wa * N + ia, c1 = (size_t)wb * N + jb;
    const size_t r2 = (size_t)wb * N + ia, c2 = (size_t)wa * N + jb;
    G[c1 * W + r1] = s; G[r1 * W + c1] = s;
    G[c2 * W + r2] = s; G[r2 * W + c2] = s;
  } else {                                    // T block: psi_x' psi_y  ->  C
    const int jb = 4 * (gb - G4) + (l & 3);
    if (jb >= N) return;
    C[((size_t)wb * N + jb) * W + (size_t)wa * N + ia] = s;
    C[((size_t)wa * N + jb) * W + (size_t)wb * N + ia] = s;
  }
}

struct kp_gram3_plan {
  int G4 = 0, nq = 0, njobs = 0, nsuper = 0;
  int wpw = 4;               // waves (jobs) per workgroup: 4 (kp_gram3_kernel) or 8 (kp_gram6_kernel, one workgroup per CU)
  uint32_t* desc = nullptr;  // device
};

void kp_gram3_plan_free(kp_gram3_plan* p) {
  if (!p) return;
  if (p->desc) (void)hipFree(p->desc);
  delete p;
}

// Row g of psi_x (one 4-column group) is paired with: its circulant half of the psi_x groups
// (g, g+1, ..., g+floor(G4/2) mod G4; antipodal pairs once) and all G4 groups of psi_y.
// All quads (A group, 4 B groups) of all rows form one list; a job (one wave) is nq CONSECUTIVE quads,
// so it spans at most two A groups when nq <= quads per row, and whole workgroups (4 jobs) fill evenly.
static int make_plan3(kp_ctx* ctx, int N, int nwt, int nq_cap, kp_gram3_plan** out, int wpw = 4, int nq_force = 0) {
  kp_gram3_plan* p = new kp_gram3_plan();
  p->wpw = wpw;
  const int G4 = (N + 3) / 4;
  p->G4 = G4;
  const int ZG = 2 * G4;
  std::vector<std::vector<int>> rows(G4);
  for (int g = 0; g < G4; ++g) {
    for (int d = 0; d <= G4 / 2; ++d) {
      if (d > 0 && 2 * d == G4 && g >= G4 / 2) continue;
      rows[g].push_back((g + d) % G4);
    }
    for (int h = 0; h < G4; ++h) rows[g].push_back(G4 + h);
  }
  size_t maxq = 0;
  for (auto& r : rows) maxq = std::max(maxq, (r.size() + 3) / 4);
  std::vector<std::pair<int, uint32_t>> quads;
  for (int g = 0; g < G4; ++g) {
    const size_t nquads = (rows[g].size() + 3) / 4;
    for (size_t q = 0; q < nquads; ++q) {
      uint32_t packed = 0;
      for (int k = 0; k < 4; ++k) {
        size_t idx = q * 4 + k;
        int gb = idx < rows[g].size() ? rows[g][idx] : ZG;
        packed |= (uint32_t)gb << (8 * k);
      }
      quads.push_back({g, packed});
    }
  }
  const int TQ = (int)quads.size();
  // cost ~ waves x (MFMA cycles of nq quads over the two k-steps of a tile + the per-tile VALU share)
  int nq = 1;
  double best = 1e300;
  constexpr int NQMAX = 6;   // 7 or 8 quads (140/160 accumulator registers) spill with the 256-register budget of 2 waves per SIMD
  for (int c = 1; c <= std::min(NQMAX, nq_cap) && (size_t)c <= maxq; ++c) {
    int waves = ((TQ + c - 1) / c + 3) / 4 * 4;
    double cost = (double)waves * (c * nwt * 33.0 + 400.0);
    if (cost < best) { best = cost; nq = c; }
  }
  if (nq_force > 0) nq = nq_force;
  else if (const char* ov = getenv("KP_GRAM3_NQ")) {   // tuning override
    int v = atoi(ov);
    if (v >= 1 && v <= std::min(NQMAX, nq_cap) && (size_t)v <= maxq) nq = v;
  }
  p->nq = nq;
  std::vector<uint32_t> desc;
  int njobs = 0;
  const uint32_t zq = (uint32_t)ZG * 0x01010101u;
  for (int q0 = 0; q0 < TQ || njobs % wpw; q0 += nq) {
    int a0 = q0 < TQ ? quads[q0].first : 0, a1 = a0, qs = nq;
    for (int q = 0; q < nq; ++q)
      if (q0 + q < TQ && quads[q0 + q].first != a0) { a1 = quads[q0 + q].first; qs = q; break; }
    desc.push_back((uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)qs << 16));
    for (int q = 0; q < nq; ++q) desc.push_back(q0 + q < TQ ? quads[q0 + q].second : zq);
    ++njobs;
  }
  p->njobs = njobs;
  p->nsuper = njobs / wpw;
  hipError_t e = hipMalloc((void**)&p->desc, desc.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(p->desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    kp_gram3_plan_free(p);
    return ctx->fail(KP_ERR_HIP, std::string("kp_fit_gram: plan upload: ") + hipGetErrorString(e));
  }
  *out = p;
  return KP_OK;
}

// m = 3 and m = 2 run the paired-weight form (TUP) of the kernel; KP_GRAM3_NOTUP=1 (read once) keeps the one-weight-per-operand form for
// A/B measurements - the two write different partial layouts, the reduction is told which
static bool gram3_tup(int bm) {
  static const bool off = getenv("KP_GRAM3_NOTUP") != nullptr;
  return bm >= 2 && !off;              // (an even number of weights: m = 2 -> 6, m = 3 -> 10)
}

template <int NQ, int BM, bool PCS, bool EXT = false, bool PRE = false, bool TUP = false>
static hipError_t launch3c(const Gram3Args& a, int grid, size_t lds, hipStream_t st) {
  static KpLdsCache lds_cache;
  {
    const size_t lds_max = (size_t)(LDS3_DOUBLES + (PCS ? LDS3_PCS_DOUBLES : 0) + (EXT ? LDS3_GAUSS_DOUBLES : 0)) * sizeof(double);
    hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_gram3_kernel<NQ, BM, PCS, EXT, PRE, TUP>, lds_max);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((kp_gram3_kernel<NQ, BM, PCS, EXT, PRE, TUP>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int NQ, int BM, bool PCS, bool EXT = false, bool PRE = false>
static hipError_t launch3b(const Gram3Args& a, int grid, size_t lds, hipStream_t st) {
  if constexpr (BM >= 2) {
    if (gram3_tup(BM)) return launch3c<NQ, BM, PCS, EXT, PRE, true>(a, grid, lds, st);
  }
  return launch3c<NQ, BM, PCS, EXT, PRE, false>(a, grid, lds, st);
}

template <int NQ>
static hipError_t launch3(const Gram3Args& a, int bm, int grid, size_t lds, hipStream_t st) {
  if (a.pre) {                           // dim_red dictionary, econ lift done by kp_gram3_prelift_kernel
    switch (bm) {
      case 1: return launch3b<NQ, 1, false, false, true>(a, grid, lds, st);
      case 2: return launch3b<NQ, 2, false, false, true>(a, grid, lds, st);
      default: return launch3b<NQ, 3, false, false, true>(a, grid, lds, st);
    }
  }
  if (a.pcs) {
    if constexpr (NQ <= 4) {             // more quads per wave spill once the projection is inlined (plans of dim_red dictionaries stay below)
      switch (bm) {
        case 1: return launch3b<NQ, 1, true>(a, grid, lds, st);
        case 2: return launch3b<NQ, 2, true>(a, grid, lds, st);
        default: return launch3b<NQ, 3, true>(a, grid, lds, st);
      }
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.df > 0 || a.ng > 0) {            // fourier / gaussian entries in the table
    switch (bm) {
      case 1: return launch3b<NQ, 1, false, true>(a, grid, lds, st);
      case 2: return launch3b<NQ, 2, false, true>(a, grid, lds, st);
      default: return launch3b<NQ, 3, false, true>(a, grid, lds, st);
    }
  }
  switch (bm) {
    case 1: return launch3b<NQ, 1, false>(a, grid, lds, st);
    case 2: return launch3b<NQ, 2, false>(a, grid, lds, st);
    default: return launch3b<NQ, 3, false>(a, grid, lds, st);
  }
}

// fourier / gaussian blocks as table entries (EXT kernel): not with a projection, <= GAUSSMAX3 centres, <= GNZMAX3 variables
static bool gram3_ext(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  return basis->fast_ext && (basis->ext_df > 0 || basis->ext_ng > 0) && b.k_pcs == 0 && basis->ext_ng <= GAUSSMAX3 &&
         (basis->ext_ng == 0 || b.nzeta <= GNZMAX3) && !getenv("KP_NO_GRAM3_EXT");
}

// dim_red dictionaries: econ lift by kp_gram3_prelift_kernel (its power table: 2 x nzeta * depth entries x 256 threads of LDS);
// fourier / gaussian dictionaries: their table entries once per snapshot by kp_gram3_prelift_ext_kernel (kp_gram3_prelift.hip)
static bool gram3_prelift(const kp_basis* basis) {
  static const bool off = getenv("KP_GRAM3_NO_PRELIFT") != nullptr;
  const BasisDev& b = basis->dev;
  if (off) return false;
  if ((b.N + 3) / 4 > 14) return false;               // the tile loader of the PRE kernel moves two 16-byte pieces per thread: 4 (8 G4 + 12) <= 512
  if (b.k_pcs > 0) return b.k_pcs <= 32 && basis->fast && b.nzeta * basis->pow_depth <= 24;      // (2 x 24 entries x 256 threads: 96 KB of LDS)
  return gram3_ext(basis) && b.nzeta <= 16 && b.nzeta * (basis->ext_Dp + 2 * basis->ext_df) + basis->ext_ng + 1 <= 64;
}

bool kp_gram3_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  if (getenv("KP_NO_GRAM3")) return false;
  // the lift deals 4 (2 nfull + weights) (item, snapshot pair) jobs over 3 x 256 slots (KP_G3_NLS = 3)
  if (KP_G3_NLS == 3 && 2 * b.nfull + (b.m + 1) * (b.m + 2) / 2 - 1 > 192) return false;
  if (gram3_ext(basis))
    return b.model_type == KP_MODEL_BILINEAR && basis->ext_max_factors <= NF3 && b.nfull <= YOFF3 && b.m >= 1 && b.m <= 3 &&
           2 * (b.nzeta + b.m) * KT3 <= 256 &&
           2 * (b.nzeta + b.m) * (basis->ext_Dp + 2 * basis->ext_df) + 1 + 2 * basis->ext_ng <= NIDMAX3;
  // dim_red dictionaries: econ layout [zeta | k_pcs principal components | 1], at most 32 components
  if (b.k_pcs > 0 && (b.k_pcs > PCSMAX3 || b.N != b.nzeta + b.k_pcs + 1 || getenv("KP_NO_GRAM3_PCS"))) return false;
  // (powers <= 4: the in-loop table build writes x .. x^4 at an entry stride of 4; the constant row's 4 entries follow the raw rows)
  return b.model_type == KP_MODEL_BILINEAR && basis->fast && basis->max_factors <= NF3 && b.nfull <= YOFF3 && basis->pow_depth <= 4 &&
         b.m >= 1 && b.m <= 3 && (2 * (b.nzeta + b.m) + 1) * KT3 <= 256 && 2 * (b.nzeta + b.m) * 4 + 4 <= NIDMAX3;   // (+ 1: the constant's table row has a thread)
}

int kp_gram3_launch(kp_ctx* ctx, const kp_basis* basis_c, const kp_snapshots* s, double* GC_dev) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W, N = b.N;
  const int BM = b.m, NWT = (BM + 1) * (BM + 2) / 2;
  if (!basis->plan3) {
    // fourier / gaussian tables: the transcendental code's constants and temporaries cost ~40 registers, so fewer quads
    // (accumulators) per wave or the kernel spills (measured: 4 quads per wave is the fastest cap, tools/gram_shapes_probe.py)
    static const int ext_cap = [] { const char* e = getenv("KP_GRAM3_EXT_NQ"); return e ? atoi(e) : 4; }();
    // round 4 experiment (KP_GRAM6=1): eight-wave workgroups, the weighted A operands in LDS (kp_gram6.hip)
    static const bool g6_on = getenv("KP_GRAM6") != nullptr;
    const bool g6 = g6_on && BM == 3 && b.k_pcs == 0 && !gram3_ext(basis) && kp_gram6_serves(7, (N + 3) / 4);
    int rc = g6 ? make_plan3(ctx, N, NWT, 7, &basis->plan3, 8, 7)
                : make_plan3(ctx, N, NWT, b.k_pcs > 0 ? 4 : gram3_ext(basis) ? ext_cap : 6, &basis->plan3);     // (one plan serves both forms of a dim_red / fourier / gaussian fit)
    if (rc) return rc;
  }
  kp_gram3_plan& plan = *basis->plan3;
  const int nfull4 = (b.nfull + 3) / 4 * 4;
  const bool ext = gram3_ext(basis);
  
  int64_t ktiles = (s->Ns + KT3 - 1) / KT3;
  int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int wg_per_cu = plan.wpw == 8 ? 1 : 2;              // __launch_bounds__(256, 2): two workgroups share a CU (kp_gram6: one of eight waves)
  if (const char* ov = getenv("KP_GRAM3_WGPCU")) wg_per_cu = std::max(1, atoi(ov));
  int64_t slots = (int64_t)std::max(8, ncu - ctx->reserve_cus) * wg_per_cu;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, slots / plan.nsuper > 0 ? slots / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);
  size_t per_split = (size_t)plan.njobs * plan.nq * NWT * 64;
  // asynchronous fits: two partial buffers, so that the next Gram kernel may run while the solve stream reduces this one
  const size_t part_bytes = ((size_t)nsplit * per_split * 8 + 255) & ~(size_t)255;
  // sized for the largest split count of this dictionary at once: a workspace that grows with the snapshot count would put
  // a hipFree + hipMalloc of ~80 MB (17 ms, and a device synchronisation) into the first large fit of a running pipeline
  const size_t part_max = ((size_t)std::max<int64_t>(nsplit, (int64_t)ncu * wg_per_cu / plan.nsuper) * per_split * 8 + 255) & ~(size_t)255;
  char* part_base = (char*)ctx->workspace(4, part_max * (ctx->reduce_stream ? 2 : 1));
  if (!part_base) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");
  double* part = (double*)(part_base + (ctx->reduce_stream ? (size_t)ctx->part_flip * part_bytes : 0));

  // dim_red dictionaries: the econ lift once per snapshot into a row buffer (kp_gram3_prelift.hip), the Gram kernel loads tiles of it.
  // Round 4's prelift kernel walked its 84 columns in ~40 us however few threads ran, so the in-kernel projection won below
  // ~45 000 pairs; round 5's works tile by tile: at the arm data's 11 999 pairs 0.040 against 0.053 ms of kernels
  // (tools/arm_shape_latency.py with KP_GRAM3_PRELIFT_MIN_NS=0).  Below a few thousand pairs the second launch is not worth it.
  static const int64_t pre_min_ns = [] { const char* e = getenv("KP_GRAM3_PRELIFT_MIN_NS"); return e ? (int64_t)atoll(e) : (int64_t)6000; }();
  bool pre = gram3_prelift(basis) && s->Ns >= pre_min_ns;
  const int pre_rl = 8 * plan.G4 + 12;
  double* pre_buf = nullptr;
  if (pre) {
    // the row buffer is ~0.7 - 1 KB per snapshot (7 - 10 GB at 1e7 pairs), in one piece: when it is more than a quarter of the
    // device's memory, or cannot be had (a busy device, several contexts of a kp_multi on one GPU), the fit runs with the
    // in-kernel lift instead of failing - same Grams, no buffer
    const size_t pre_bytes = (size_t)ktiles * KT3 * pre_rl * 8;
    if (ctx->hbm_bytes > 0 && pre_bytes > (size_t)ctx->hbm_bytes / 4) pre = false;
    if (pre) {
      pre_buf = (double*)ctx->workspace(14, pre_bytes);
      if (!pre_buf) {
        (void)hipGetLastError();
        pre = false;
      }
    }
  }
  if (pre && !ext && !basis->d_pcsT) {
    if (hipMalloc(&basis->d_pcsT, (size_t)b.nfull * 32 * 8) != hipSuccess) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");
    KP_HIP(ctx, kp_gram3_pcs_transpose_launch(b.pcs, b.nfull, b.k_pcs, (double*)basis->d_pcsT, ctx->stream));
  }
  const size_t lds = (size_t)(LDS3_DOUBLES + (b.k_pcs > 0 && !pre ? 2 * nfull4 * 16 : 0) + (ext && !pre ? LDS3_GAUSS_DOUBLES : 0)) * sizeof(double);

  Gram3Args a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.G4 = plan.G4;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.D = ext ? basis->ext_Dp + 2 * basis->ext_df : basis->pow_depth;
  a.recipes = (const uint32_t*)(ext ? basis->d_recipes_ext : basis->d_recipes);
  a.Dp = ext ? basis->ext_Dp : basis->pow_depth;
  a.df = ext ? basis->ext_df : 0;
  a.ng = ext ? basis->ext_ng : 0;
  a.centres = b.centres;
  a.desc = plan.desc;
  a.part = part;
  a.njobs = plan.njobs;
  a.pcs = b.k_pcs > 0 && !pre ? b.pcs : nullptr;
  a.nfull4 = nfull4;
  a.pre = pre_buf;
  a.pre_rl = pre_rl;
  const int grid = plan.nsuper * nsplit;
  // every event record is a barrier packet the command processor works through between two Gram kernels: the
  // pipelined path keeps two (kernel start / end; the end also releases the reduction on the solve stream)
  const bool pipelined = ctx->reduce_stream || ctx->ring_timing;
  if (!pipelined) KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipEvent_t ev_start = ctx->evp[0], ev_end = ctx->evp[1];
  // deferred-solve pipeline: every event record is a barrier packet between two Gram kernels, so only every 4th launch
  // is timed (the mean over the ring is what kp_synchronize reports)
  const bool timed = !ctx->ring_timing || (ctx->ring_skip++ & 3) == 0;
  if (pipelined && timed) {                       // pipelined fits: a ring of event pairs, averaged at kp_synchronize
    ev_start = ctx->ring[2 * ctx->ring_pos];
    ev_end = ctx->ring[2 * ctx->ring_pos + 1];
    ctx->ring_pos = (ctx->ring_pos + 1) % 64;
    if (ctx->ring_n < 64) ++ctx->ring_n;
  }
  if (timed) KP_HIP(ctx, hipEventRecord(ev_start, ctx->stream));
  hipError_t e;
  if (pre && ext)
    KP_HIP(ctx, kp_gram3_prelift_ext_launch(BM, s->alpha, s->beta, s->u, s->Ns, ktiles * KT3, b.nzeta, basis->ext_Dp, basis->ext_df, basis->ext_ng, b.nfull, plan.G4,
                                            (const uint32_t*)basis->d_recipes_ext, b.centres, pre_buf, pre_rl, ctx->stream));
  else if (pre)
    KP_HIP(ctx, kp_gram3_prelift_launch(BM, s->alpha, s->beta, s->u, s->Ns, ktiles * KT3, b.nzeta, basis->pow_depth, b.nfull, b.k_pcs, N, plan.G4,
                                        (const uint32_t*)basis->d_recipes, (const double*)basis->d_pcsT, pre_buf, pre_rl, ctx->stream));
  if (plan.wpw == 8) e = kp_gram6_launch_kernel(a, plan.nq, grid, ctx->stream);
  else switch (plan.nq) {
#ifndef KP_G3_DEV            // (development builds instantiate the headline shape only: the file takes minutes otherwise)
    case 1: e = launch3<1>(a, BM, grid, lds, ctx->stream); break;
    case 2: e = launch3<2>(a, BM, grid, lds, ctx->stream); break;
    case 3: e = launch3<3>(a, BM, grid, lds, ctx->stream); break;
    case 4: e = launch3<4>(a, BM, grid, lds, ctx->stream); break;
    case 5: e = launch3<5>(a, BM, grid, lds, ctx->stream); break;
    default: e = launch3<6>(a, BM, grid, lds, ctx->stream); break;
#else
    default: e = launch3b<6, 3, false>(a, grid, lds, ctx->stream); break;
#endif
  }
  KP_HIP(ctx, e);
  if (timed) KP_HIP(ctx, hipEventRecord(ev_end, ctx->stream));
  hipStream_t rs = ctx->stream;
  if (ctx->reduce_stream) {                       // reduction (and everything after it) belongs to the solve stream
    rs = ctx->reduce_stream;
    KP_HIP(ctx, hipStreamWaitEvent(rs, ev_end, 0));
    KP_HIP(ctx, hipEventRecord(ctx->evp[4], rs));   // start of the reduction on its own stream (timer 6)
    ctx->solve_chained = true;                      // the solve stream already waits for this Gram kernel
  }
  ctx->reduce_timed_from = ctx->reduce_stream ? 4 : 1;
  hipLaunchKernelGGL(kp_gram3_reduce_kernel, dim3(plan.njobs * plan.nq * NWT), dim3(nsplit >= 128 ? 1024 : nsplit >= 32 ? 512 : 256), 0, rs, part, nsplit, plan.njobs,
                     plan.nq, NWT, BM, plan.desc, plan.G4, N, W, GC_dev, GC_dev + (size_t)W * W, plan.wpw != 8 && gram3_tup(BM) ? 1 : 0);
  KP_HIP(ctx, hipGetLastError());
  if (!pipelined) KP_HIP(ctx, hipEventRecord(ctx->ev1, rs));
  if (!ctx->ring_timing) KP_HIP(ctx, hipEventRecord(ctx->evp[2], rs));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  // executed on the matrix pipe per pair (timer 10): jobs (padding included) x quads x weights MFMAs per 4 snapshots, 512 flop
  // each; dim_red: + the projection pcs' psi of every workgroup of a split (2 x nfull4 / 4 MFMAs per wave and tile)
  ctx->timers[10] = (double)plan.njobs * plan.nq * NWT * 128.0 + (b.k_pcs > 0 && !pre ? (double)plan.nsuper * nfull4 * 128.0 : 0.0);
  return KP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// LINEAR models with dim_red (the first model of the reference's example_sysid.m: poly-3, econ lift [zeta; pcs' psi; 1]).
// Their row [psi(x), u] is a column subset of the BILINEAR row psi (x) [1; u] of the same dictionary - the last dictionary
// column is the constant, so u_i = psi_N u_i is column (i + 1) N + N - 1 of it - hence G and C of the linear fit are
// sub-blocks of the bilinear Grams.  The only kernel that served these dictionaries was the general one (two-stage lift through
// LDS, v_mfma_f64_16x16x4: 0.55 ms per 1e5 pairs at N = 34); the Kronecker kernel with the in-kernel projection forms the
// bilinear Grams of the same dictionary in 0.21 ms, and a gather of (N + m)^2 entries makes the linear ones of them.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kp_gram3_linear_gather_kernel(const double* __restrict__ GCb, int N, int m, int Wb, double* __restrict__ GC) {
  const int W = N + m;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * W * W) return;
  const int which = e / (W * W), r = e - which * W * W, i = r % W, j = r / W;
  const int bi = i < N ? i : (i - N + 1) * N + N - 1, bj = j < N ? j : (j - N + 1) * N + N - 1;
  GC[e] = GCb[(size_t)which * Wb * Wb + (size_t)bj * Wb + bi];
}

static kp_basis* gram3_shadow(const kp_basis* basis_c) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  if (!basis->shadow_bil) {
    kp_basis* sh = new kp_basis(*basis);               // shares every device array (never freed through the shadow)
    sh->dev.model_type = KP_MODEL_BILINEAR;
    sh->dev.W = sh->dev.N * (sh->dev.m + 1);
    sh->plan = nullptr; sh->plan2 = nullptr; sh->plan3 = nullptr; sh->plan5 = nullptr; sh->shadow_bil = nullptr;
    sh->d_pcsT = nullptr;                              // (its own transposed projection matrix, built on first use)
    basis->shadow_bil = sh;
  }
  return basis->shadow_bil;
}

void kp_gram3_shadow_free(kp_basis* basis) {
  if (!basis) return;
  if (basis->shadow_bil) {
    kp_gram3_plan_free(basis->shadow_bil->plan3);
    if (basis->shadow_bil->d_pcsT) (void)hipFree(basis->shadow_bil->d_pcsT);
    delete basis->shadow_bil;
    basis->shadow_bil = nullptr;
  }
  if (basis->shadow_full) {
    kp_gram_plan_free(basis->shadow_full->plan);
    kp_gram2_plan_free(basis->shadow_full->plan2);
    kp_gram5_plan_free(basis->shadow_full->plan5);
    delete basis->shadow_full;
    basis->shadow_full = nullptr;
  }
}

bool kp_gram3_linear_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  if (b.model_type != KP_MODEL_LINEAR || b.k_pcs <= 0 || b.m < 1 || getenv("KP_NO_GRAM3_LINEAR")) return false;
  return kp_gram3_applicable(gram3_shadow(basis));
}

int kp_gram3_linear_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev) {
  kp_basis* sh = gram3_shadow(basis);
  const int N = basis->dev.N, m = basis->dev.m, W = basis->dev.W, Wb = sh->dev.W;
  double* tmp = (double*)ctx->workspace(10, (size_t)2 * Wb * Wb * 8);
  if (!tmp) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");
  // the reduction (and with it the gather) may belong to the solve stream of the asynchronous pipeline
  hipStream_t rs = ctx->reduce_stream ? ctx->reduce_stream : ctx->stream;
  int rc = kp_gram3_launch(ctx, sh, s, tmp);
  if (rc) return rc;
  hipLaunchKernelGGL(kp_gram3_linear_gather_kernel, dim3((2 * W * W + 255) / 256), dim3(256), 0, rs, tmp, N, m, Wb, GC_dev);
  KP_HIP(ctx, hipGetLastError());
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  // timer 10 (flop EXECUTED on the matrix pipe per pair) stays what the inner launch set: the work added here - the gather, the
  // congruence products 2 (Wf^2 W + Wf W^2) flop - is per FIT and runs on the vector pipe, so it is in the kernel time of the
  // bench's roofline block but, rightly, not in its matrix-pipe flop
  return KP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// dim_red dictionaries of LINEAR and NONLINEAR models (the first and third model of the reference's example_sysid.m).  The
// econ lift [v; pcs' psi_full(v); 1] (Ksysid.m:1594-1618) is a LINEAR map of the full lift - the variables are the first
// columns of the full dictionary, the constant its last -  psi_econ = T' psi_full,  T = [E_vars | pcs | e_const]  (and the
// identity on the input columns of a linear row), so
//     G_econ = T' (Psi_x' Psi_x) T,      C_econ = T' (Psi_x' Psi_y) T :
// the full dictionary's Grams by the monomial kernels (kp_gram5 / kp_gram2: 0.115 ms linear, 0.34 ms nonlinear poly-3 per 1e5
// pairs) and two small congruence products, instead of the general kernel's per-pair projection through LDS (0.55 ms and
// 4.2 ms).  Bilinear dim_red dictionaries keep the Kronecker kernel with the in-kernel projection (0.21 ms against 0.40 ms).
// ---------------------------------------------------------------------------------------------------------------------
struct CongT { int nvars, k, nfull, N, W, Wf; const double* pcs; };
// entry of T: row c of the full row layout (dictionary columns, then the inputs of a linear row), econ column j
__device__ __forceinline__ int cong_unit(const CongT& t, int j) {        // >= 0: T(:, j) = e_idx; -1: a pcs column
  if (j < t.nvars) return j;
  if (j < t.nvars + t.k) return -1;
  if (j == t.N - 1) return t.nfull - 1;
  return t.nfull + (j - t.N);                                            // input columns of a linear row
}
// R (Wf x W) = M (Wf x Wf) T, for M = G_full and C_full (blockIdx.y)
__global__ __launch_bounds__(256) void kp_cong_right_kernel(const double* __restrict__ GCf, CongT t, double* __restrict__ R) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= t.Wf * t.W) return;
  const int i = e % t.Wf, j = e / t.Wf;
  const double* M = GCf + (size_t)blockIdx.y * t.Wf * t.Wf;
  const int u = cong_unit(t, j);
  double v;
  if (u >= 0) v = M[i + (size_t)u * t.Wf];
  else {
    const double* pc = t.pcs + (size_t)(j - t.nvars) * t.nfull;
    double s0 = 0.0, s1 = 0.0;
    int c = 0;
    for (; c + 1 < t.nfull; c += 2) { s0 += M[i + (size_t)c * t.Wf] * pc[c]; s1 += M[i + (size_t)(c + 1) * t.Wf] * pc[c + 1]; }
    if (c < t.nfull) s0 += M[i + (size_t)c * t.Wf] * pc[c];
    v = s0 + s1;
  }
  R[(size_t)blockIdx.y * t.Wf * t.W + e] = v;
}
// out (W x W) = T' R
__global__ __launch_bounds__(256) void kp_cong_left_kernel(const double* __restrict__ R, CongT t, double* __restrict__ GC) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= t.W * t.W) return;
  const int i = e % t.W, j = e / t.W;
  const bool sym = blockIdx.y == 0;                    // G: one triangle is computed and mirrored (exactly symmetric, like the kernels' own G)
  if (sym && i > j) return;
  const double* Rj = R + (size_t)blockIdx.y * t.Wf * t.W + (size_t)j * t.Wf;
  const int u = cong_unit(t, i);
  double v;
  if (u >= 0) v = Rj[u];
  else {
    const double* pc = t.pcs + (size_t)(i - t.nvars) * t.nfull;
    double s0 = 0.0, s1 = 0.0;
    int c = 0;
    for (; c + 1 < t.nfull; c += 2) { s0 += pc[c] * Rj[c]; s1 += pc[c + 1] * Rj[c + 1]; }
    if (c < t.nfull) s0 += pc[c] * Rj[c];
    v = s0 + s1;
  }
  GC[(size_t)blockIdx.y * t.W * t.W + e] = v;
  if (sym && i < j) GC[(size_t)j + (size_t)i * t.W] = v;
}

static kp_basis* gram_shadow_full(const kp_basis* basis_c) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  if (!basis->shadow_full) {
    kp_basis* sh = new kp_basis(*basis);               // shares every device array (never freed through the shadow)
    sh->dev.k_pcs = 0;
    sh->dev.pcs = nullptr;
    sh->d_pcs = nullptr;
    sh->dev.N = sh->dev.nfull;
    sh->dev.W = sh->dev.model_type == KP_MODEL_LINEAR ? sh->dev.nfull + sh->dev.m : sh->dev.nfull;
    sh->plan = nullptr; sh->plan2 = nullptr; sh->plan3 = nullptr; sh->plan5 = nullptr; sh->shadow_bil = nullptr; sh->shadow_full = nullptr;
    sh->d_pcsT = nullptr;
    basis->shadow_full = sh;
  }
  return basis->shadow_full;
}

bool kp_gram_congruence_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  if ((b.model_type != KP_MODEL_LINEAR && b.model_type != KP_MODEL_NONLINEAR) || b.k_pcs <= 0 || b.N != b.nvars + b.k_pcs + 1 ||
      getenv("KP_NO_GRAM_CONGRUENCE"))
    return false;
  const kp_basis* sh = gram_shadow_full(basis);
  return kp_gram5_applicable(sh) || kp_gram2_applicable(sh);
}

int kp_gram_congruence_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev) {
  kp_basis* sh = gram_shadow_full(basis);
  const BasisDev& b = basis->dev;
  const int W = b.W, Wf = sh->dev.W;
  double* GCf = (double*)ctx->workspace(10, (size_t)2 * Wf * Wf * 8);
  double* R = (double*)ctx->workspace(11, (size_t)2 * Wf * W * 8);
  if (!GCf || !R) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");
  hipStream_t rs = ctx->stream;                       // kp_gram5 / kp_gram2 reduce on the Gram stream
  int rc = kp_gram5_applicable(sh) ? kp_gram5_launch(ctx, sh, s, GCf) : kp_gram2_launch(ctx, sh, s, GCf);
  if (rc) return rc;
  CongT t{b.nvars, b.k_pcs, b.nfull, b.N, W, Wf, b.pcs};
  hipLaunchKernelGGL(kp_cong_right_kernel, dim3((Wf * W + 255) / 256, 2), dim3(256), 0, rs, GCf, t, R);
  hipLaunchKernelGGL(kp_cong_left_kernel, dim3((W * W + 255) / 256, 2), dim3(256), 0, rs, R, t, GC_dev);
  KP_HIP(ctx, hipGetLastError());
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  // timer 10 (flop EXECUTED on the matrix pipe per pair) stays what the inner launch set: the work added here - the gather, the
  // congruence products 2 (Wf^2 W + Wf W^2) flop - is per FIT and runs on the vector pipe, so it is in the kernel time of the
  // bench's roofline block but, rightly, not in its matrix-pipe flop
  return KP_OK;
}

// kp_gram3_prelift.hip - econ lift of a dim_red dictionary, once per snapshot, for the Kronecker Gram kernel (kp_gram3.hip, PRE mode).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#pragma clang diagnostic ignored "-Wc++20-extensions"      // lambdas with explicit template parameters (the hand-unrolled MFMA / lift schedule)

#include "kp_gram3_args.h"

#define KT3 8     // snapshots per tile of kp_gram3_kernel
#define PES 2     // doubles between two entries of a snapshot in the tile layout [snapshot pair][entry][2]
#define NF3 3     // single-variable powers per column

// ---------------------------------------------------------------------------------------------------------------------
// Econ lift of a dim_red dictionary, once per snapshot (round 4).  kp_gram3_kernel<.,.,true> lifts the FULL dictionary of a
// tile and projects it (pcs' psi on the matrix pipe, LDS-bound) in EVERY workgroup of a snapshot split - as much work as the
// N = 34 Gram itself.  Here one thread per snapshot forms, for both sides, [zeta ; pcs' psi_full(zeta) ; 1] (Ksysid.m:1594-1618) and the
// entries the Gram kernel's tile loader wants: [psi_x (4 G4) | psi_y (4 G4) | the 9 weights ut_a ut_b, 3 zeros] x KT3 snapshots per tile; snapshots past Ns
// give zero rows (the tail mask).  The power table of a thread lives in LDS ([entry][thread]: conflict-free); the projection
// matrix is read through wave-uniform addresses ([full column][32 components], zero padded: scalar loads, the multiply-adds
// take it as a scalar operand).  Measured at 1e5 pairs, 84 full columns, 27 components (tools/prelift_time.py under rocprofv3):
// 109 us with one (snapshot, side) per thread and a row per snapshot in memory (8-byte stores scattered over 67 MB); 80 us with
// the tile layout [entry][8 snapshots] (full 64-byte sectors); 89 us with both sides in one thread (half the scalar loads, half
// the waves); 65 us with the column's factors read without branches (id 255 reads an entry of ones), which lets the loads of
// the next column run under the 64 multiply-adds of this one.  Timing-only builds: without the multiply-adds 25 us, with ONE
// matrix row for every column (scalar-cache hits) 43 us - the 21 KB matrix streams through a 16 KB scalar cache; two passes of
// 16 components (10.5 KB each) were slower (117 us: the factors are read twice).  The multiply-adds alone are 14 us at the f64
// vector peak.  Counters (profiles/r04_new_kernels_pmc_summary.json): 6 650 VALU instructions per wave in 138 k cycles, 1.5 waves per
// SIMD (1e5 snapshots are only 1 564 waves) - latency-bound.  Tried on top and no better: only the raw values in LDS, powers formed
// where they are used (25 instead of 74 KB per workgroup: 64 us - there are not enough workgroups for the occupancy to matter); a
// quarter of the components per thread, four times the waves (80 us: the factors are formed four times); the matrix through the
// vector L1 instead of scalar loads (all lanes one address: 88 us).
// ---------------------------------------------------------------------------------------------------------------------
#define PRE_T 256
__global__ __launch_bounds__(PRE_T) void kp_gram3_pcs_transpose_kernel(const double* __restrict__ pcs, int nfull, int k, double* __restrict__ pcsT) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nfull * 32) {
    const int c = e >> 5, p = e & 31;
    pcsT[e] = p < k ? pcs[c + (size_t)p * nfull] : 0.0;
  }
}

// Round 5: the projection on the MATRIX pipe, by workgroups of four CONSUMER and four PRODUCER waves.  The round-4 kernel (one
// thread per snapshot, the matrix as scalar operands of 64 multiply-adds per column) was latency-bound: 1e5 snapshots are 1 564
// waves, 1.5 per SIMD, each walking 84 columns one after the other, and the 21 KB matrix streamed through a 16 KB scalar cache -
// 65 us where its multiply-adds are 14.  Now a tile of 8 snapshots (both sides) is the unit, 12 500 per 1e5 pairs dealt to
// persistent workgroups, four tiles per workgroup and round; per slot:
//   producer wave (two rounds ahead): raw loads; power table x_v^e of both sides ([entry][side][8 snapshots], 20 doubles per entry:
//     entries 8 dwords apart modulo the banks), double-buffered; the entries that are no components (zeta, the constant,
//     padding, the 9 weights) straight to memory;
//   consumer wave: pcs' psi of THIS round's tile by v_mfma_f64_4x4x4_4b - the four blocks = four groups of four COMPONENTS: the
//     A operand (4 snapshots x 4 full columns of the Psi tile [column][side][8 snapshots], 18 doubles per column) is ONE
//     ds_read_b64 at `lane base + immediate`, requested four steps ahead IN THE SOURCE; the B operand pcs[4 kk + k][16 t + 4 blk
//     + j] differs per lane - the whole 84 x 32 matrix is 42 registers per lane, loaded once; two MFMAs (component tiles t = 0, 1)
//     share an A operand - and, one micro-operation behind each MFMA pair, the FULL lift of the NEXT round's tile into the other
//     Psi buffer: a lane takes a (column, side) item, three 16-byte table reads at register offsets, two pairs of multiplies, a
//     16-byte store per two snapshots (168 items in 3 rounds of 24 micro-operations, branch free: idle lanes work on a spare
//     column).  A D register (snapshot = lane >> 4, component = 4 blk + j) stores 256 contiguous bytes per snapshot pair of the
//     tile layout [snapshot pair][entry][2] - whose consecutive entries the Gram kernel's loader writes into its Psi rows
//     without the 4-way bank conflict of the round-4 layout [entry][8 snapshots];
//   ONE LDS-only barrier per round (__syncthreads() would also drain the memory counter: the loads just requested, the stores
//   just issued).
// Measured (1e5 pairs, 84 -> 27 components; tools/prelift_abl5.sh = timing-only builds through KP_PM_ABL, tools/prelift_phase_probe.py
// = in-kernel cycle counters, KP_PM_ABL=32): **38 us against round 4's 65.**  The MFMAs alone are 17 us (168 per tile at 16.5
// cycles: the floor of this instruction), the 67 MB of output another ~17 us of HBM writes beside them.  The way here, each
// step measured:
//   * every wave doing all steps in turn - one wave per tile (two waves per SIMD) or a pair of waves sharing a tile (three) -
//     took 38 - 45 us however the steps were tuned (62 us with a branch per k-step: no operand prefetch; 45 us with the lane
//     forming psi[s][c] itself in front of every MFMA pair; 41 us with the Psi tile staged in LDS; 38 us once a tile's loads
//     were consumed BEFORE its stores - the memory counter retires in order, a wait for loads issued behind stores waits for the
//     stores' acknowledgements), and the time was the SUM of the steps' times: identical waves fall into step - all in the
//     MFMA phase, then all in the latency-bound phases (every phase twice its instruction time, matrix pipe 41 % busy);
//   * fixed roles with the LIFT on the producer (one producer per tile, or one per side): the consumer's round is 3 400 cycles
//     of MFMAs + 500 of stores, the producer's 5 000 - while a wave streams f64 MFMAs on a SIMD, the other waves' vector
//     instructions wait ~100 cycles each when they depend on one another (power table 1 400 - 2 000 cycles per tile, 700
//     without the MFMAs), whichever wave has priority (`s_setprio`: 41.5 -> 40.1 us): 40 - 42 us;
//   * the shipped form - the lift inside the consumer's own instruction stream (a vector instruction of the MFMA wave costs the
//     stream ~5 cycles): 50 us as first written (the lift's stores and the A-operand reads are integer LDS addresses to hipcc: it
//     will not move a read above an earlier store, so every MFMA pair waited for a read issued just in front of it), 39 us with
//     the A operands requested four steps ahead in the source, 38 us with a scheduling barrier per step (the scheduler sank the
//     table reads to their uses).  Consumer round now 4 300 cycles (3 000 of instructions) + 500 of stores, producer 5 000.
// Measured late in round 5 (-DPM_NS=2 / 1: the same pairs in two / four INDEPENDENT workgroups per CU, each with its own barrier,
// the arrangement that pays for the dense products of kp_tn_gemm.h): 46.8 and 209 us against 38.9 - with fewer pairs per workgroup
// the power tables and the raw loads are shared by fewer consumers, and at one pair the 128-register cap spills.
// What would be next: two producers per slot (table per side) to take the producer off the critical path, and a second look at
// what the consumer's 1 300 extra cycles per round are.
#ifndef PM_NS
#define PM_NS 4       // tiles (producer / consumer pairs) per workgroup
#endif
#define PM_T (128 * PM_NS)
#define PM_WGPCU (4 / PM_NS)      // workgroups per CU (8 waves per CU either way)
#define PM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PM_ES 20      // doubles per power-table entry: [side 0: 8 snapshots][side 1: 8][4 of padding]
#define PM_CS 18      // doubles per Psi-tile column: [side 0: 8][side 1: 8][2 of padding]: 16-byte writes of consecutive columns and the
                      // operand reads of two neighbouring columns both spread over the banks
// Round 5, last form: the CONSUMER lifts its own next tile between its MFMAs (a vector instruction of the MFMA wave itself costs
// the stream ~5 cycles; the same instruction from another wave of the SIMD waits ~100 when it depends on its predecessor), one
// micro-operation of the lift - a 16-byte table read, a pair of multiplies, a 16-byte store - behind each pair of MFMAs, branch
// free (idle lanes of the last round work on a spare column).  The producer wave of the slot is left with what has no vector
// work to speak of: raw loads, the power table of the tile after next, the entries that are no components.
template <int BM, int NK, int NRAW>
__global__ __launch_bounds__(PM_T, PM_WGPCU) void kp_gram3_prelift_mfma_kernel(const double* __restrict__ alpha, const double* __restrict__ beta, const double* __restrict__ u,
                                                                        int64_t Ns, int64_t ktiles, int nzeta, int D, int nfull, int k_pcs, int N, int G4,
                                                                        const uint32_t* __restrict__ recipes, const double* __restrict__ pcsT,
                                                                        double* __restrict__ out, int rl, int abl) {
  extern __shared__ double sm[];                        // per slot: tab[2][nid + 1][PM_ES] | utab[BM + 1][8] | psi[2][4 NK + 1][PM_CS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave & (PM_NS - 1);
  const bool producer = wave >= PM_NS;
  const int nid = nzeta * D;
  const int tab_d = ((nid + 1) * PM_ES + 1) & ~1;       // one power table
  const int utab_d = ((BM + 1) * 8 + 1) & ~1;
  constexpr int PSI_D = (4 * NK + 1) * PM_CS;           // one Psi tile (+ a spare column for idle lanes)
  double* wsm = sm + slot * (2 * tab_d + utab_d + 2 * PSI_D);
  double* utab = wsm + 2 * tab_d;
  double* psi = utab + utab_d;
  const unsigned wbase = (unsigned)(uintptr_t)wsm, pbase = (unsigned)(uintptr_t)psi;   // (low 32 bits of a generic LDS address = the LDS byte address)
  const unsigned TABB = (unsigned)tab_d * 8u;
  constexpr unsigned PSIB = (unsigned)PSI_D * 8u;
  const int64_t tstep = (int64_t)gridDim.x * PM_NS;
  const int64_t tile0 = (int64_t)blockIdx.x * PM_NS + slot;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) dbl2* lds_d2;
  typedef __attribute__((address_space(3))) dbl2* lds_d2w;
  typedef const __attribute__((address_space(3))) double* lds_d;
  if (producer) {
    // ================================================ producer ================================================
    if (lane < 8) utab[BM * 8 + lane] = 1.0;            // ut_0 = 1
    for (int e = lane; e < 2 * PM_CS; e += 64) {        // columns beyond the dictionary (zero operands) stay zero: cleared once, both buffers
      for (int c = nfull; c < 4 * NK; ++c) { if (e < PM_CS) psi[c * PM_CS + e] = 0.0; else psi[PSI_D + c * PM_CS + e - PM_CS] = 0.0; }
    }
    // (snapshot, eighth) roles for loads, tables and the entries that are not components
    const int ls = lane & 7, lq = lane >> 3;
    const int nsv = 2 * nzeta + BM;                     // raw columns of a snapshot: alpha | beta | u
    auto raw_ptr = [&](int sv) -> const double* {
      return sv < nzeta ? alpha + (int64_t)sv * Ns : sv < 2 * nzeta ? beta + (int64_t)(sv - nzeta) * Ns : u + (int64_t)(sv - 2 * nzeta) * Ns;
    };
    double raw[NRAW];                                   // NRAW = ceil((2 nzeta + BM) / 8) (2 or 4)
    auto load_raw = [&](int64_t tile) {
      const int64_t snap = tile * KT3 + ls;
#pragma unroll
      for (int m = 0; m < NRAW; ++m) {
        const int sv = lq + 8 * m;
        raw[m] = (sv < nsv && snap < Ns && tile < ktiles) ? raw_ptr(sv)[snap] : 0.0;
      }
    };
    // entries that are not components, fixed per lane: zeta (one per raw value), then up to NOTH of [constant | padding of the
    // last column group] x 2 sides and the 12 weight slots; a weight = product of two utab entries (ut = [1, u]; slots beyond
    // the (BM + 1)(BM + 2) / 2 - 1 pairs are zero)
    constexpr int NOTH = 4;                             // 2 npad + 12 <= 32 entries over 8 lane groups
    const int npad = 4 * G4 - N + 1;                    // constant + padding entries per side
    int oth_ent[NOTH], oth_kind[NOTH];                  // kind 0: nothing, 1: zero, 2: tail mask (the constant), 3: weight
    unsigned oth_a[NOTH], oth_b[NOTH];
    const unsigned ubase = (unsigned)(uintptr_t)utab;
#pragma unroll
    for (int t = 0; t < NOTH; ++t) {
      const int e = lq + 8 * t;
      oth_kind[t] = 0; oth_ent[t] = 0; oth_a[t] = oth_b[t] = ubase;
      if (e < 2 * npad) {
        const int side = e >= npad, j = side ? e - npad : e;
        oth_ent[t] = side * 4 * G4 + N - 1 + j;
        oth_kind[t] = j == 0 ? 2 : 1;
      } else if (e < 2 * npad + 12) {
        const int w = e - 2 * npad;
        oth_ent[t] = 8 * G4 + w;
        oth_kind[t] = 1;
        int cnt = 0;
#pragma unroll
        for (int x = 0; x <= BM; ++x)
#pragma unroll
          for (int y = x; y <= BM; ++y) {
            if (cnt == w + 1) {
              oth_kind[t] = 3;
              oth_a[t] = ubase + (unsigned)((x == 0 ? BM : x - 1) * 8 + ls) * 8u;
              oth_b[t] = ubase + (unsigned)((y - 1) * 8 + ls) * 8u;
            }
            ++cnt;
          }
      }
    }
    const int o_off = (ls >> 1) * 2 * rl + (ls & 1);
    // table(tile, buffer): power table of both sides + inputs from the raw values just consumed; then the tile's other entries
    auto table_and_entries = [&](int64_t tile, int buf) {
      const bool valid = tile * KT3 + ls < Ns;
      double* tb = wsm + buf * tab_d;
      double xs[NRAW];
#pragma unroll
      for (int m = 0; m < NRAW; ++m) {
        asm volatile("" : "+v"(raw[m]));                // (a use: the compiler's wait for the loads lands here - behind it only
        xs[m] = raw[m];                                 //  stores and loads that are a round old)
      }
#pragma unroll
      for (int m = 0; m < NRAW; ++m) {
        const int sv = lq + 8 * m;
        if (sv < 2 * nzeta) {
          const int side = sv >= nzeta, v = side ? sv - nzeta : sv;
          const double x1 = xs[m], x2 = x1 * x1, x3 = x2 * x1, x4 = x2 * x2;
          double* t = tb + (v * D) * PM_ES + side * 8 + ls;
          t[0] = x1;
          if (D > 1) t[PM_ES] = x2;
          if (D > 2) t[2 * PM_ES] = x3;
          if (D > 3) t[3 * PM_ES] = x4;
          if (D > 4) {
            double pw = x4 * x1;
            for (int e = 4; e < D; ++e) {
              t[e * PM_ES] = pw;
              pw *= x1;
            }
          }
        } else if (sv < nsv) {
          utab[(sv - 2 * nzeta) * 8 + ls] = xs[m];
        }
      }
      if (lq < 2) tb[nid * PM_ES + lq * 8 + ls] = valid ? 1.0 : 0.0;    // "no factor" = the tail mask (the full dictionary's constant column is all of it)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double oth_v[NOTH];
#pragma unroll
      for (int t = 0; t < NOTH; ++t) {
        const double wa = *reinterpret_cast<lds_d>((uintptr_t)oth_a[t]), wb = *reinterpret_cast<lds_d>((uintptr_t)oth_b[t]);
        oth_v[t] = oth_kind[t] == 3 ? wa * wb : (oth_kind[t] == 2 && valid) ? 1.0 : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next call's inputs overwrite utab)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (tile < ktiles && !(abl & 8)) {
        double* ot = out + tile * (int64_t)(KT3 * rl);
#pragma unroll
        for (int m = 0; m < NRAW; ++m) {
          const int sv = lq + 8 * m;
          if (sv < 2 * nzeta) ot[o_off + (sv < nzeta ? sv : 4 * G4 + sv - nzeta) * PES] = xs[m];
        }
#pragma unroll
        for (int t = 0; t < NOTH; ++t)
          if (oth_kind[t]) ot[o_off + oth_ent[t] * PES] = oth_v[t];
      }
    };
    // Order inside a call and across rounds: take the raw values requested a round ago -> tables -> this tile's stores -> request
    // the next tile's values.  The memory counter retires in order and a wait for loads also waits for every store issued in
    // front of them: in this order both are a whole round old when the wait comes.
    load_raw(tile0);
    table_and_entries(tile0, 0);
    load_raw(tile0 + tstep);
    PM_LDS_BARRIER();                                   // P0: table(0) is there; the consumer lifts tile 0
    table_and_entries(tile0 + tstep, 1);
    load_raw(tile0 + 2 * tstep);
    PM_LDS_BARRIER();                                   // P1: Psi(0) and table(1) are there
    int it = 0;
    long long pph[2] = {0, 0}, plast = (abl & 32) ? clock64() : 0;
    for (int64_t tb0 = (int64_t)blockIdx.x * PM_NS; tb0 < ktiles; tb0 += tstep, ++it) {   // (uniform trip count over the workgroup)
      table_and_entries(tile0 + (int64_t)(it + 2) * tstep, it & 1);      // table(it + 2) where table(it) was: lift(it) read it a round ago
      load_raw(tile0 + (int64_t)(it + 3) * tstep);
      if (abl & 32) { const long long tn_ = clock64(); pph[0] += tn_ - plast; plast = tn_; }
      PM_LDS_BARRIER();
      if (abl & 32) { const long long tn_ = clock64(); pph[1] += tn_ - plast; plast = tn_; }
    }
    if ((abl & 32) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && slot < 2)
      printf("producer (%d, %d): table+entries+loads %lld  barrier %lld cycles, %d rounds\n", (int)blockIdx.x, slot, pph[0], pph[1], it);
  } else {
    // ================================================ consumer ================================================
    const int li = lane & 3, blk = (lane >> 2) & 3, kq = lane >> 4;
    double b0[NK], b1[NK];                              // zero beyond the dictionary
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int c = 4 * kk + kq;
      const bool on = c < nfull;
      b0[kk] = on ? pcsT[(size_t)c * 32 + 4 * blk + li] : 0.0;
      b1[kk] = on ? pcsT[(size_t)c * 32 + 16 + 4 * blk + li] : 0.0;
    }
    // lift items of this lane: item = lane + 64 r -> (column, side); byte offsets of its three factors in power table 0 and of its
    // Psi column in buffer 0 (idle lanes of the last round: the entry of ones, the spare column)
    constexpr int NIT = (2 * 4 * NK + 63) / 64;
    // (one offset register per (round, factor, TABLE): the table's size is a run-time number, and an address add in front of
    // every table read is a vector instruction in the MFMA stream)
    unsigned foff[2][NIT][NF3], poff[NIT];
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int item = lane + 64 * r;
      const int side = item >= nfull, c = side ? item - nfull : item;
      const bool on = item < 2 * nfull;
      const uint32_t rc = on ? recipes[c] : 0xffffffffu;
#pragma unroll
      for (int f = 0; f < NF3; ++f) {
        int id = (int)((rc >> (8 * f)) & 255u);
        id = id == 255 ? nid : id;
        foff[0][r][f] = wbase + (unsigned)(id * PM_ES + (on ? side : 0) * 8) * 8u;
        foff[1][r][f] = foff[0][r][f] + TABB;
      }
      poff[r] = pbase + (unsigned)((on ? c : 4 * NK) * PM_CS + (on ? side : 0) * 8) * 8u;
    }
    const unsigned abase0 = pbase + (unsigned)(kq * PM_CS + li) * 8u;                   // A operand: column 4 kk + kq, snapshot 4 g + li
    const int pc = 4 * blk + li;
    const int d_off = (kq >> 1) * 2 * rl + (kq & 1) + (nzeta + pc) * PES;             // + g * 4 rl + side * 4 G4 PES (+ 16 PES)
    dbl2 f0[4], f1[4], f2[4];
    long long tph[3] = {0, 0, 0}, tlast = 0;
    // one micro-operation of the lift of the NEXT tile (table tn, Psi buffer pn): 24 per round of 64 items -
    // 12 reads, 4 + 4 pairs of multiplies, 4 stores
    auto lift_op = [&](auto slot_c, auto tn_c, auto pn_c) __attribute__((always_inline)) {
      constexpr int S = decltype(slot_c)::value;
      constexpr int TN = decltype(tn_c)::value;         // table of the next tile (0 / 1)
      constexpr unsigned pn = decltype(pn_c)::value;    // byte offset of its Psi buffer
      constexpr int r = S / 24, o = S % 24;
      if constexpr (r < NIT) {
        if constexpr (o < 12) {
          constexpr int f = o / 4, q = o % 4;
          const dbl2 v = *reinterpret_cast<lds_d2>((uintptr_t)(foff[TN][r][f] + 16u * q));
          if constexpr (f == 0) f0[q] = v; else if constexpr (f == 1) f1[q] = v; else f2[q] = v;
        } else if constexpr (o < 16) {
          f0[o - 12] = f0[o - 12] * f1[o - 12];
        } else if constexpr (o < 20) {
          f0[o - 16] = f0[o - 16] * f2[o - 16];
        } else {
          *reinterpret_cast<lds_d2w>((uintptr_t)(poff[r] + pn + 16u * (o - 20))) = f0[o - 20];
        }
      }
    };
    PM_LDS_BARRIER();                                   // P0: table(0) is there
    {                                                   // lift of tile 0, not interleaved with anything
      [&]<int... S>(std::integer_sequence<int, S...>) {
        (lift_op(std::integral_constant<int, S>{}, std::integral_constant<int, 0>{}, std::integral_constant<unsigned, 0u>{}), ...);
      }(std::make_integer_sequence<int, 24 * NIT>{});
    }
    PM_LDS_BARRIER();                                   // P1: Psi(0) and table(1) are there
    if (abl & 32) tlast = clock64();
    // one round with the parity of the CURRENT tile as a constant (buffer offsets are immediates)
    auto round = [&](auto par_c, int64_t tile) __attribute__((always_inline)) {
      constexpr unsigned P = decltype(par_c)::value;
      const unsigned abase = abase0 + P * PSIB;
      using TNc = std::integral_constant<int, 1 - (int)P>;                            // next tile: the other table,
      using PNc = std::integral_constant<unsigned, (1u - P) * PSIB>;                  // the other Psi buffer
      double acc[2][2][2];
      // The A operands are requested PF steps ahead IN THE SOURCE: the lift's stores go to the other Psi buffer, but both are
      // integer LDS addresses to the compiler - it will not move an operand read above an earlier store, so left to itself every
      // MFMA pair waited for a read issued just in front of it (measured: 50 us).
      constexpr int PF = 4, NJ = 4 * NK;
      auto a_addr = [&](int j) -> unsigned { return abase + (unsigned)((j % NK) * 4 * PM_CS * 8 + (j / (2 * NK)) * 64 + ((j / NK) % 2) * 32); };
      double av[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) av[j] = *reinterpret_cast<lds_d>((uintptr_t)a_addr(j));
      [&]<int... J>(std::integer_sequence<int, J...>) {
        ([&] {
          constexpr int side = J / (2 * NK), g = (J / NK) % 2, kk = J % NK;
          if constexpr (kk == 0) { acc[side][g][0] = 0.0; acc[side][g][1] = 0.0; }
          const double a_now = av[J % PF];
          if constexpr (J + PF < NJ) av[J % PF] = *reinterpret_cast<lds_d>((uintptr_t)a_addr(J + PF));
          acc[side][g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_now, b0[kk], acc[side][g][0], 0, 0, 0);
          acc[side][g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_now, b1[kk], acc[side][g][1], 0, 0, 0);
          lift_op(std::integral_constant<int, J>{}, TNc{}, PNc{});
          __builtin_amdgcn_sched_barrier(0);            // (the scheduler sinks the table reads to their uses otherwise: each then waits out its LDS latency)
        }(), ...);
      }(std::make_integer_sequence<int, 4 * NK>{});
      if (abl & 32) { const long long tn_ = clock64(); tph[0] += tn_ - tlast; tlast = tn_; }
      if (tile < ktiles && !(abl & 4)) {
        double* ot = out + tile * (int64_t)(KT3 * rl);
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            double* og = ot + (g * 4 * rl + side * 4 * G4 * PES);
            if (pc < k_pcs) og[d_off] = acc[side][g][0];
            if (pc + 16 < k_pcs) og[d_off + 16 * PES] = acc[side][g][1];
          }
      }
      if (abl & 32) { const long long tn_ = clock64(); tph[1] += tn_ - tlast; tlast = tn_; }
      PM_LDS_BARRIER();
      if (abl & 32) { const long long tn_ = clock64(); tph[2] += tn_ - tlast; tlast = tn_; }
    };
    int it = 0;
    for (int64_t tb0 = (int64_t)blockIdx.x * PM_NS; tb0 < ktiles; tb0 += tstep, ++it) {
      const int64_t tile = tile0 + (int64_t)it * tstep;
      if (it & 1) round(std::integral_constant<unsigned, 1>{}, tile);
      else round(std::integral_constant<unsigned, 0>{}, tile);
    }
    if ((abl & 32) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && slot < 2)
      printf("consumer (%d, %d): mfma+lift %lld  stores %lld  barrier %lld cycles, %d rounds\n", (int)blockIdx.x, slot, tph[0], tph[1], tph[2], it);
  }
}

hipError_t kp_gram3_pcs_transpose_launch(const double* pcs, int nfull, int k, double* pcsT, hipStream_t st) {
  hipLaunchKernelGGL(kp_gram3_pcs_transpose_kernel, dim3((nfull * 32 + PRE_T - 1) / PRE_T), dim3(PRE_T), 0, st, pcs, nfull, k, pcsT);
  return hipGetLastError();
}

template <int BM, int NK, int NRAW>
static hipError_t prelift_mfma_launch(const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t ktiles, int nzeta, int D, int nfull, int k_pcs, int N,
                                      int G4, const uint32_t* recipes, const double* pcsT, double* out, int rl, hipStream_t st) {
  const size_t lds = (size_t)PM_NS * ((size_t)(2 * ((((nzeta * D + 1) * PM_ES) + 1) & ~1) + ((((BM + 1) * 8) + 1) & ~1) + 2 * (4 * NK + 1) * PM_CS)) * 8;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static KpLdsCache cache;
  hipError_t e = kp_ensure_lds(cache, (const void*)kp_gram3_prelift_mfma_kernel<BM, NK, NRAW>, lds);
  if (e != hipSuccess) return e;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((ktiles + PM_NS - 1) / PM_NS, (int64_t)cus * PM_WGPCU));
  hipLaunchKernelGGL((kp_gram3_prelift_mfma_kernel<BM, NK, NRAW>), dim3((unsigned)grid), dim3(PM_T), lds, st, alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4,
                     recipes, pcsT, out, rl, kp_abl_int("KP_PM_ABL"));
  return hipGetLastError();
}

template <int BM>
static hipError_t prelift_mfma_launch_nk(int nk, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t ktiles, int nzeta, int D, int nfull, int k_pcs,
                                         int N, int G4, const uint32_t* recipes, const double* pcsT, double* out, int rl, hipStream_t st) {
  const bool few = 2 * nzeta + BM <= 16;
#define KP_PM(NKT) (few ? prelift_mfma_launch<BM, NKT, 2>(alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl, st) \
                        : prelift_mfma_launch<BM, NKT, 4>(alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl, st))
  if (nk <= 8) return KP_PM(8);
  if (nk <= 14) return KP_PM(14);
  if (nk <= 21) return KP_PM(21);
  return KP_PM(24);
#undef KP_PM
}

// Ns_pad = ktiles * KT3 (the row buffer holds whole tiles).  Needs nfull <= 96, 2 nzeta + BM <= 32 (kp_gram3_applicable's own limits).
hipError_t kp_gram3_prelift_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int D, int nfull, int k_pcs,
                                   int N, int G4, const uint32_t* recipes, const double* pcsT, double* out, int rl, hipStream_t st) {
  const int nk = (nfull + 3) / 4;
  const int64_t ktiles = Ns_pad / KT3;
  if (nk > 24 || 2 * nzeta + BM > 32 || 2 * (4 * G4 - N + 1) + 12 > 32) return hipErrorInvalidValue;
  if (BM == 1) return prelift_mfma_launch_nk<1>(nk, alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl, st);
  if (BM == 2) return prelift_mfma_launch_nk<2>(nk, alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl, st);
  return prelift_mfma_launch_nk<3>(nk, alpha, beta, u, Ns, ktiles, nzeta, D, nfull, k_pcs, N, G4, recipes, pcsT, out, rl, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same row buffer for dictionaries with fourier / gaussian blocks and no projection (def_fourierLift Ksysid.m:694-731,
// def_gaussianLift :790-817; the EXT recipes of kp_context.hip).  kp_gram3_kernel<.,.,false,true> builds their table entries -
// sincospi, exp(-|zeta - c|^2) with the centres in LDS - in every workgroup of a snapshot split and next to the MFMA loop, which
// costs it 40 registers (plans capped at 4 quads per wave).  Here: one thread per snapshot, both sides; table entries per side
// [variable v: x^1..x^Dp, cos / sin(2 pi j x) j = 1..df][gaussian centre c], a column = product of <= 3 entries (id < 128:
// variable entry, 128 + c: gaussian, 255: none), written straight to the tile layout.
// ---------------------------------------------------------------------------------------------------------------------
#define PREX_T 128
template <int BM>
__global__ __launch_bounds__(PREX_T) void kp_gram3_prelift_ext_kernel(const double* __restrict__ alpha, const double* __restrict__ beta, const double* __restrict__ u,
                                                                      int64_t Ns, int64_t Ns_pad, int nzeta, int Dp, int df, int ng, int nfull, int G4,
                                                                      const uint32_t* __restrict__ recipes, const double* __restrict__ centres,
                                                                      double* __restrict__ out, int rl) {
  extern __shared__ double tab[];                       // [nzeta * D + ng][PREX_T], then one entry of ones
  const int tid = threadIdx.x;
  const int64_t snap = (int64_t)blockIdx.x * PREX_T + tid;
  if (snap >= Ns_pad) return;
  const bool valid = snap < Ns;
  const int D = Dp + 2 * df, nv = nzeta * D, ne = nv + ng;
  // raw values of both sides first (all loads in flight together), then the entries from registers
  constexpr int NZMAX = 16;
  double xa[NZMAX], xb[NZMAX];
#pragma unroll
  for (int v = 0; v < NZMAX; ++v) {
    xa[v] = (valid && v < nzeta) ? alpha[(int64_t)v * Ns + snap] : 0.0;
    xb[v] = (valid && v < nzeta) ? beta[(int64_t)v * Ns + snap] : 0.0;
  }
  // one side at a time through ONE table (27 entries x 128 threads = 27 KB for 20 gaussians on 6 states: five workgroups per CU;
  // a table per side left two)
  double* t0 = tab + tid;
  double* ox = out + (snap / KT3) * (int64_t)(KT3 * rl) + ((snap % KT3) >> 1) * (int64_t)(2 * rl) + (snap & 1);   // tile layout [snapshot pair][entry][2]
  double* oy = ox + (int64_t)4 * G4 * PES;
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const double* xs = side ? xb : xa;
#pragma unroll
    for (int v = 0; v < NZMAX; ++v) {
      if (v < nzeta) {
        double pw = xs[v];
        for (int e = 0; e < Dp; ++e) {
          t0[(v * D + e) * PREX_T] = pw;
          pw *= xs[v];
        }
        if (df > 0) {                                   // harmonics by the angle-addition recurrence, as kp_gram3_kernel's loader
          double s1, c1;
          sincospi(2.0 * xs[v], &s1, &c1);
          double cj = c1, sj = s1, cm = 1.0, sm1 = 0.0;
          for (int h = 0; h < df; ++h) {
            t0[(v * D + Dp + 2 * h) * PREX_T] = cj;
            t0[(v * D + Dp + 2 * h + 1) * PREX_T] = sj;
            const double cn = 2.0 * c1 * cj - cm, sn = 2.0 * c1 * sj - sm1;
            cm = cj; sm1 = sj; cj = cn; sj = sn;
          }
        }
      }
    }
    for (int c = 0; c < ng; ++c) {                      // exp(-|zeta - centre|^2), the centres through wave-uniform addresses
      double r2 = 0.0;
#pragma unroll
      for (int i = 0; i < NZMAX; ++i)
        if (i < nzeta) {
          const double dlt = xs[i] - centres[c * nzeta + i];
          r2 = fma(dlt, dlt, r2);
        }
      t0[(nv + c) * PREX_T] = exp(-r2);
    }
    t0[(size_t)ne * PREX_T] = 1.0;
    double* o = side ? oy : ox;
#pragma unroll 2
    for (int c = 0; c < nfull; ++c) {
      const uint32_t r = recipes[c];
      double ps = 1.0;
#pragma unroll
      for (int f = 0; f < NF3; ++f) {
        const int id = (int)((r >> (8 * f)) & 255u);
        ps *= t0[(id == 255 ? ne : id >= 128 ? nv + (id - 128) : id) * PREX_T];
      }
      o[c * PES] = valid ? ps : 0.0;
    }
  }
  for (int j = nfull; j < 4 * G4; ++j) { ox[j * PES] = 0.0; oy[j * PES] = 0.0; }
  {
    double ut[BM + 1];
    ut[0] = 1.0;
#pragma unroll
    for (int i = 0; i < BM; ++i) ut[1 + i] = valid ? u[(int64_t)i * Ns + snap] : 0.0;
    double* w = ox + (int64_t)8 * G4 * PES;
    int cnt = 0;
#pragma unroll
    for (int x = 0; x <= BM; ++x)
#pragma unroll
      for (int y = x; y <= BM; ++y) {
        if (cnt > 0) w[(cnt - 1) * PES] = valid ? ut[x] * ut[y] : 0.0;
        ++cnt;
      }
    for (int j = cnt - 1; j < 12; ++j) w[j * PES] = 0.0;
  }
}

hipError_t kp_gram3_prelift_ext_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int Dp, int df, int ng,
                                       int nfull, int G4, const uint32_t* recipes, const double* centres, double* out, int rl, hipStream_t st) {
  const size_t plds = ((size_t)(nzeta * (Dp + 2 * df) + ng) + 1) * PREX_T * 8;
  const dim3 pg((unsigned)((Ns_pad + PREX_T - 1) / PREX_T));
#define KP_PRELIFTX(M) hipLaunchKernelGGL(kp_gram3_prelift_ext_kernel<M>, pg, dim3(PREX_T), plds, st, alpha, beta, u, Ns, Ns_pad, nzeta, Dp, df, ng, nfull, G4, recipes, centres, out, rl)
  if (BM == 1) KP_PRELIFTX(1);
  else if (BM == 2) KP_PRELIFTX(2);
  else KP_PRELIFTX(3);
#undef KP_PRELIFTX
  return hipGetLastError();
}

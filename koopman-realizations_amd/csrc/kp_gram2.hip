// Fused lift + Gram kernel, second generation (monomial dictionaries without pcs):
//   * v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4): the only f64 matrix instruction that
//     reaches the 78.6 TFLOP/s datasheet rate on gfx950 (16.5 cycles/instr measured;
//     v_mfma_f64_16x16x4_f64 takes ~140).  A 16x16 output tile = 4 instructions: the B fragment
//     (16 columns x 4 snapshots) against the A fragment rotated by 0/4/8/12 lanes inside each
//     16-lane row (DPP row_ror), each giving one "block diagonal" of the tile.
//   * a wave's job = ONE tile-row of Psi_x (A fragment loaded and rotated once per k-step) against
//     NACC column tiles of [Psi_x | Psi_y]: 1 LDS read + 1 address add per 4 MFMAs.
//   * the lift of snapshot tile t+1 is cut into per-column chunks issued between the MFMAs of
//     tile t, so it runs in the matrix pipe's shadow.
// Replaces the per-row lift loop of Ksysid.get_Koopman (Ksysid.m:1030-1065) and the products
// Px'Px, Px'Py (Ksysid.m:1114,1125).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kp_internal.h"

#ifndef KP_ABL
#define KP_ABL 0   // timing-only ablations (results wrong): 1 no lift in loop, 2 also no raw store, 3 no MFMA
#endif
#define KT 8    // snapshots per LDS tile (two k-steps)
#define NW 8    // waves per workgroup (two per SIMD: one wave's LDS/VALU issue hides behind the other's MFMAs)
#define NTHR (64 * NW)
#define CPT 3   // dictionary columns per lifting thread (32 column slots x 3 => nfull <= 96)

struct Gram2Args {
  BasisDev b;
  const double* alpha;
  const double* beta;
  const double* u;
  int64_t Ns;
  int Wp;               // padded row length of the Psi tiles (doubles), == 16 (mod 32)
  int nsuper;           // workgroups per snapshot split
  int ktiles_per_split;
  int D;                // depth of the power table
  const uint32_t* recipes;  // [nfull]
  const uint32_t* desc; // [njobs][1 + NACC]: a_off, then b_off per tile (doubles, rel. to Psi buffer)
  const int* tile_out;  // [njobs][NACC] output tile id or -1
  double* part;         // [nsplit][ntile_out][4][64]
  int ntile_out;
};

template <int CTRL>
__device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// BM: 0 = no input expansion (linear / nonlinear rows), 1..3 = bilinear with m = BM inputs.
template <int NACC, int BM>
__global__ __launch_bounds__(NTHR, 2) void kp_gram2_kernel(Gram2Args a) {
  extern __shared__ double sm[];
  const BasisDev& b = a.b;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int super = blockIdx.x % a.nsuper;
  const int split = blockIdx.x / a.nsuper;
  const int job = super * NW + wave;
  const int nzm = b.nzeta + b.m;
  const int nrawrows = 2 * nzm;
  const int D = a.D;
  const int Wp = a.Wp;
  // LDS (doubles): pow[2][(nrawrows*D + 1)][KT] (last row: ones) | psi[2][2 sides][KT][Wp]
  const int NID = nrawrows * D + 1;              // power-table entries per snapshot (last: the constant 1)
  const int pow_stride = NID * KT;                // layout [snapshot][id]: lanes with different ids hit different banks
  const int psi_base = 2 * pow_stride;
  const int psi_stride = 2 * KT * Wp;
  const int trash = 2 * psi_stride;   // one spare row (offset from psi_base): target of masked-off column writes

  // ---- MFMA operand offsets ----
  const int lane_off = (lane >> 4) * Wp + (lane & 15);
  const uint32_t* jd = a.desc + (size_t)job * (1 + NACC);
  const int ao = lane_off + (int)jd[0];
  int bo[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) bo[q] = lane_off + (int)jd[1 + q];

  double acc[NACC][4];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[q][s] = 0.0;

  // ---- one-time LDS setup ----
  for (int e = tid; e < 2 * psi_stride + Wp; e += NTHR) sm[psi_base + e] = 0.0;   // padding columns stay zero
  if (tid < 2 * KT) sm[(tid / KT) * pow_stride + (tid % KT) * NID + nrawrows * D] = 1.0;

  // ---- lifting thread constants ----
  const int jl = tid & 15, combo = tid >> 4, ls = combo & (KT - 1), lside = (combo >> 3) & 1, half = combo >> 4;
  int foff[CPT][4];
  int woff[CPT];
  bool wok[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = jl + 16 * (2 * i + half);
    wok[i] = c < b.nfull;
    const uint32_t r = wok[i] ? a.recipes[c] : 0xffffffffu;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int id = (int)((r >> (8 * f)) & 255u);
      foff[i][f] = ls * NID + (id == 255 ? nrawrows * D : lside * nzm * D + id);
    }
    woff[i] = wok[i] ? (lside * KT + ls) * Wp + c : trash + jl;   // relative to the Psi buffer of the tile
  }
  const int uoff = ls * NID + b.nzeta * D;        // + j*D : u_j (e = 1) of snapshot ls
  const bool lin = b.model_type == KP_MODEL_LINEAR;
  const int N = b.N, m = b.m;

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT - 1) / KT;
  const int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // ---- raw loader (thread -> (row, snapshot)); rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)] ----
  // raw loader: value e = tid + j*NTHR of the tile -> (row e / KT, snapshot e % KT), up to LR per thread
  constexpr int LR = 3;
  struct RawRegs { double v[LR]; };
  bool ld_on[LR];
  int ld_r[LR], ld_s[LR];
  const double* ld_src[LR];
#pragma unroll
  for (int j = 0; j < LR; ++j) {
    const int e = tid + j * NTHR;
    ld_on[j] = e < nrawrows * KT;
    ld_r[j] = e / KT;
    ld_s[j] = e % KT;
    ld_src[j] = nullptr;
    if (ld_on[j]) {
      int rr = ld_r[j] % nzm;
      ld_src[j] = rr < b.nzeta ? ((ld_r[j] < nzm ? a.alpha : a.beta) + (int64_t)rr * a.Ns) : (a.u + (int64_t)(rr - b.nzeta) * a.Ns);
    }
  }
  auto load_raw = [&](int64_t kt) -> RawRegs {
    RawRegs x;
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      int64_t i = kt * KT + ld_s[j];
      x.v[j] = (ld_on[j] && i < a.Ns) ? ld_src[j][i] : 0.0;
    }
    return x;
  };
  auto store_raw = [&](int buf, const RawRegs& x) {   // powers x^1..x^D
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      if (!ld_on[j]) continue;
      double* dst = sm + buf * pow_stride + ld_s[j] * NID + ld_r[j] * D;
      double p = x.v[j];
      for (int e = 0; e < D; ++e) {
        dst[e] = p;
        p *= x.v[j];
      }
    }
  };

  // lift of snapshot tile kt (power-table buffer rb -> Psi buffer pb), cut into per-column chunks
  // with separate read and write stages so LDS latency hides behind the MFMAs in between
  double uv0 = 0.0, uv1 = 0.0, uv2 = 0.0, vmask = 0.0;
  double lf[CPT][4];
  auto lift_begin = [&](int rb, int pb, int64_t kt) {
    const double* T = sm + rb * pow_stride;
    vmask = (kt * KT + ls) < a.Ns ? 1.0 : 0.0;
    if (BM > 0) uv0 = T[uoff];
    if (BM > 1) uv1 = T[uoff + D];
    if (BM > 2) uv2 = T[uoff + 2 * D];
    if (BM == 0 && lin && half == 0 && jl < m) {   // [psi , u]  (Ksysid.m:1062)
      double* P = sm + psi_base + pb * psi_stride;
      P[(lside * KT + ls) * Wp + N + jl] = T[uoff + jl * D] * vmask;
    }
  };
  auto lift_read = [&](int i, int rb) {
    const double* T = sm + rb * pow_stride;
#pragma unroll
    for (int f = 0; f < 4; ++f) lf[i][f] = T[foff[i][f]];
  };
  auto lift_write = [&](int i, int pb) {
    // a masked-off column writes into the spare row; for valid columns the offset is inside buffer pb
    double* P = sm + psi_base + (wok[i] ? pb * psi_stride : 0) + woff[i];
    const double v = (lf[i][0] * lf[i][1]) * (lf[i][2] * lf[i][3]) * vmask;
    P[0] = v;
    if (BM > 0) P[N] = v * uv0;
    if (BM > 1) P[2 * N] = v * uv1;
    if (BM > 2) P[3 * N] = v * uv2;
  };

  // prologue: tile 0 -> power table -> Psi buffer 0; tile 1 -> power table 1
  store_raw(0, load_raw(kt0));
  __syncthreads();
  lift_begin(0, 0, kt0);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    lift_read(i, 0);
    lift_write(i, 0);
  }
  store_raw(1, load_raw(kt0 + 1));
  __syncthreads();

  constexpr int NSTEP = (KT / 4) * NACC;          // tile steps (4 MFMAs each) per snapshot tile
  constexpr int SP = NSTEP / CPT > 0 ? NSTEP / CPT : 1;   // tile steps between lift chunks
  constexpr int LAG = SP / 2 > 0 ? SP / 2 : 1;            // tile steps between a chunk's reads and its writes
  for (int t = 0; t < nkt; ++t) {
    const RawRegs rawreg = load_raw(kt0 + t + 2);
    const int cur = t & 1, nxt = cur ^ 1;
    const double* P = sm + psi_base + cur * psi_stride;
    lift_begin(nxt, nxt, kt0 + t + 1);
    {
      // operand fetch runs PF tile steps ahead of the MFMAs that consume it
      constexpr int PF = 4;
      double bvs[NSTEP];
      double af[KT / 4][4];
#pragma unroll
      for (int kk = 0; kk < KT / 4; ++kk) af[kk][0] = P[kk * 4 * Wp + ao];
#pragma unroll
      for (int i = 0; i < PF && i < NSTEP; ++i) bvs[i] = P[(i / NACC) * 4 * Wp + bo[i % NACC]];
#pragma unroll
      for (int kk = 0; kk < KT / 4; ++kk) {
        af[kk][1] = row_ror<0x124>(af[kk][0]);     // group blk holds original group (blk-1)&3
        af[kk][2] = row_ror<0x128>(af[kk][0]);
        af[kk][3] = row_ror<0x12c>(af[kk][0]);
      }
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int kk = step / NACC, q = step % NACC;
        if (step + PF < NSTEP) bvs[step + PF] = P[((step + PF) / NACC) * 4 * Wp + bo[(step + PF) % NACC]];
        const double bv = bvs[step];
#if KP_ABL != 3
        acc[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][0], bv, acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][1], bv, acc[q][1], 0, 0, 0);
        acc[q][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][2], bv, acc[q][2], 0, 0, 0);
        acc[q][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[kk][3], bv, acc[q][3], 0, 0, 0);
#else
        acc[q][0] += bv * af[kk][0];
#endif
        // lift of the NEXT snapshot tile: chunk i reads at step i*SP, writes LAG steps later
#if KP_ABL != 1 && KP_ABL != 2
        if (step % SP == 0 && step / SP < CPT) lift_read(step / SP, nxt);
        if (step >= LAG && (step - LAG) % SP == 0 && (step - LAG) / SP < CPT) lift_write((step - LAG) / SP, nxt);
#endif
      }
#pragma unroll
      for (int i = 0; i < CPT; ++i) {              // chunks that did not fit inside the MFMA loop (tiny NACC)
        if (i * SP >= NSTEP) lift_read(i, nxt);
        if (i * SP + LAG >= NSTEP) lift_write(i, nxt);
      }
    }
#if KP_ABL != 2
    store_raw(cur, rawreg);                        // raw tile t+2 -> power table `cur` (read while lifting tile t)
#endif
    __syncthreads();
  }

  // epilogue: partial tiles, [split][tile][s][lane]
#pragma unroll
  for (int q = 0; q < NACC; ++q) {
    const int to = a.tile_out[(size_t)job * NACC + q];
    if (to >= 0) {
      double* dst = a.part + ((size_t)split * a.ntile_out + to) * 256 + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[64 * s] = acc[q][s];
    }
  }
}

// Partial-tile reduction for the 4x4x4 register layout: instr s, lane l (blk = (l>>2)&3):
// row = 16*tr + 4*((blk - s)&3) + (l>>4),  col = 16*tc + 4*blk + (l&3).
__global__ __launch_bounds__(256) void kp_gram2_reduce_kernel(const double* __restrict__ part, int nsplit, int ntile_out,
                                                              const int* __restrict__ tile_info /* [ntile][3] kind,tr,tc */, int W,
                                                              double* __restrict__ G, double* __restrict__ C) {
  const int tile = blockIdx.x;
  const int t = threadIdx.x;
  double sum = 0.0;
  for (int p = 0; p < nsplit; ++p) sum += part[((size_t)p * ntile_out + tile) * 256 + t];
  const int s = t >> 6, l = t & 63, blk = (l >> 2) & 3;
  const int kind = tile_info[tile * 3], tr = tile_info[tile * 3 + 1], tc = tile_info[tile * 3 + 2];
  const int i = tr * 16 + 4 * ((blk - s) & 3) + (l >> 4);
  const int j = tc * 16 + 4 * blk + (l & 3);
  if (i < W && j < W) {
    if (kind == 0) {
      if (tr != tc) {
        G[(size_t)j * W + i] = sum;
        G[(size_t)i * W + j] = sum;
      } else if (i <= j) {           // diagonal tile: write the upper half and mirror it (exact symmetry)
        G[(size_t)j * W + i] = sum;
        G[(size_t)i * W + j] = sum;
      }
    } else {
      C[(size_t)j * W + i] = sum;
    }
  }
}

struct kp_gram2_plan {
  int nt = 0, Wp = 0, nacc = 0, njobs = 0, nsuper = 0, ntile_out = 0;
  char* tab = nullptr;   // device: desc | tile_out | tile_info
  size_t off_to = 0, off_ti = 0;
};

void kp_gram2_plan_free(kp_gram2_plan* p) {
  if (!p) return;
  if (p->tab) (void)hipFree(p->tab);
  delete p;
}

// Tiles hosted by tile-row r of Psi_x: all C tiles (r, *) and a balanced share of the G tiles
// touching r (G(r,c) may be computed as (r,c) or as its transpose (c,r); the reduction writes both
// triangles).  Rows longer than the largest accumulator count are cut into several jobs.
static int make_plan2(kp_ctx* ctx, int W, kp_gram2_plan** out) {
  kp_gram2_plan* p = new kp_gram2_plan();
  const int nt = (W + 15) / 16;
  p->nt = nt;
  int wp = nt * 16;
  while (wp % 32 != 16) wp += 16;
  p->Wp = wp;
  struct Tile { int kind, tr, tc; };
  std::vector<std::vector<Tile>> rows(nt);
  for (int r = 0; r < nt; ++r) {
    rows[r].push_back({0, r, r});
    for (int c = 0; c < nt; ++c) rows[r].push_back({1, r, c});
  }
  for (int d = 1; d <= nt / 2; ++d)          // circulant assignment of the off-diagonal G tiles
    for (int r = 0; r < nt; ++r) {
      int c = (r + d) % nt;
      if (2 * d == nt && r >= nt / 2) continue;   // antipodal pairs appear twice
      rows[r].push_back({0, r, c});
    }
  // tiles per wave-job: the candidate that wastes the fewest accumulator slots over whole workgroups
  static const int cand[] = {4, 6, 8, 11, 12, 16};
  int nacc = 16;
  double best_eff = -1.0;
  size_t total = 0;
  for (auto& v : rows) total += v.size();
  for (int c : cand) {
    size_t nj = 0;
    for (auto& v : rows) nj += (v.size() + c - 1) / c;
    size_t slots = (nj + NW - 1) / NW * NW * (size_t)c;
    double eff = (double)total / (double)slots;
    if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && c > nacc)) {
      best_eff = eff;
      nacc = c;
    }
  }
  if (const char* ov = getenv("KP_GRAM2_NACC")) {   // tuning override
    int v = atoi(ov);
    for (int c : cand)
      if (c == v) nacc = v;
  }
  p->nacc = nacc;
  std::vector<uint32_t> desc;
  std::vector<int> tile_out, tile_info;
  int id = 0, njobs = 0;
  for (int r = 0; r < nt; ++r) {
    const size_t nchunk = (rows[r].size() + nacc - 1) / nacc, per = (rows[r].size() + nchunk - 1) / nchunk;   // even chunks
    for (size_t s0 = 0; s0 < rows[r].size(); s0 += per) {
      desc.push_back((uint32_t)(r * 16));
      for (int q = 0; q < nacc; ++q) {
        if ((size_t)q < per && s0 + q < rows[r].size()) {
          const Tile& t = rows[r][s0 + q];
          desc.push_back((uint32_t)((t.kind ? KT * p->Wp : 0) + t.tc * 16));
          tile_out.push_back(id++);
          tile_info.push_back(t.kind); tile_info.push_back(t.tr); tile_info.push_back(t.tc);
        } else {
          desc.push_back(0u);
          tile_out.push_back(-1);
        }
      }
      ++njobs;
    }
  }
  while (njobs % NW) {   // pad to whole workgroups with idle jobs
    desc.push_back(0u);
    for (int q = 0; q < nacc; ++q) { desc.push_back(0u); tile_out.push_back(-1); }
    ++njobs;
  }
  p->njobs = njobs;
  p->nsuper = njobs / NW;
  p->ntile_out = id;
  size_t b_desc = desc.size() * 4, b_to = tile_out.size() * 4, b_ti = tile_info.size() * 4;
  p->off_to = b_desc;
  p->off_ti = b_desc + b_to;
  hipError_t e = hipMalloc((void**)&p->tab, b_desc + b_to + b_ti);
  if (e == hipSuccess) e = hipMemcpy(p->tab, desc.data(), b_desc, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->tab + p->off_to, tile_out.data(), b_to, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->tab + p->off_ti, tile_info.data(), b_ti, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    kp_gram2_plan_free(p);
    return ctx->fail(KP_ERR_HIP, std::string("kp_fit_gram: plan upload: ") + hipGetErrorString(e));
  }
  *out = p;
  return KP_OK;
}

template <int NACC, int BM>
static hipError_t launch2b(const Gram2Args& a, int grid, size_t lds, hipStream_t st) {
  static KpLdsCache lds_cache;
  {
    hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_gram2_kernel<NACC, BM>, lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((kp_gram2_kernel<NACC, BM>), dim3(grid), dim3(NTHR), lds, st, a);
  return hipGetLastError();
}

template <int NACC>
static hipError_t launch2(const Gram2Args& a, int bm, int grid, size_t lds, hipStream_t st) {
  switch (bm) {
    case 1: return launch2b<NACC, 1>(a, grid, lds, st);
    case 2: return launch2b<NACC, 2>(a, grid, lds, st);
    case 3: return launch2b<NACC, 3>(a, grid, lds, st);
    default: return launch2b<NACC, 0>(a, grid, lds, st);
  }
}

bool kp_gram2_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  return basis->fast && b.k_pcs == 0 && b.nfull <= 32 * CPT && b.m <= 16 && (b.model_type != KP_MODEL_BILINEAR || (b.m >= 1 && b.m <= 3)) &&
         2 * (b.nzeta + b.m) * KT <= 3 * 256;
}

int kp_gram2_launch(kp_ctx* ctx, const kp_basis* basis_c, const kp_snapshots* s, double* GC_dev) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W;
  if (!basis->plan2) {
    int rc = make_plan2(ctx, W, &basis->plan2);
    if (rc) return rc;
  }
  kp_gram2_plan& plan = *basis->plan2;
  const int D = basis->pow_depth;
  const int nraw = 2 * (b.nzeta + b.m);
  size_t lds = ((size_t)2 * (nraw * D + 1) * KT + (size_t)2 * 2 * KT * plan.Wp + plan.Wp) * sizeof(double);
  const int bm = b.model_type == KP_MODEL_BILINEAR ? b.m : 0;
  if (lds > 160 * 1024 || (uint32_t)(2 * KT * plan.Wp) > 65535u)
    return kp_gram_wide_launch(ctx, basis, s, GC_dev);   // W > ~580: lifted panels in HBM + TN products (kp_wide.hip)
  int64_t ktiles = (s->Ns + KT - 1) / KT;
  int ncu = std::max(8, (ctx->num_cu > 0 ? ctx->num_cu : 256) - ctx->reserve_cus);
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, ncu / plan.nsuper > 0 ? ncu / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);
  size_t b_part = (size_t)nsplit * plan.ntile_out * 256 * 8;
  double* part = (double*)ctx->workspace(4, b_part);
  if (!part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");

  Gram2Args a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.Wp = plan.Wp;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.D = D;
  a.recipes = (const uint32_t*)basis->d_recipes;
  a.desc = (const uint32_t*)plan.tab;
  a.tile_out = (const int*)(plan.tab + plan.off_to);
  a.part = part;
  a.ntile_out = plan.ntile_out;
  const int grid = plan.nsuper * nsplit;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], ctx->stream));
  hipError_t e;
  switch (plan.nacc) {
    case 4: e = launch2<4>(a, bm, grid, lds, ctx->stream); break;
    case 6: e = launch2<6>(a, bm, grid, lds, ctx->stream); break;
    case 8: e = launch2<8>(a, bm, grid, lds, ctx->stream); break;
    case 11: e = launch2<11>(a, bm, grid, lds, ctx->stream); break;
    case 12: e = launch2<12>(a, bm, grid, lds, ctx->stream); break;
    default: e = launch2<16>(a, bm, grid, lds, ctx->stream); break;
  }
  KP_HIP(ctx, e);
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], ctx->stream));
  hipLaunchKernelGGL(kp_gram2_reduce_kernel, dim3(plan.ntile_out), dim3(256), 0, ctx->stream, part, nsplit, plan.ntile_out,
                     (const int*)(plan.tab + plan.off_ti), W, GC_dev, GC_dev + (size_t)W * W);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], ctx->stream));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  ctx->timers[10] = (double)plan.njobs * plan.nacc * 512.0;   // executed on the matrix pipe per pair (timer 10)
  return KP_OK;
}

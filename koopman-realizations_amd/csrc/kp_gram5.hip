// Fused lift + Gram kernel for LINEAR and NONLINEAR monomial dictionaries (Px = [psi(zeta), u] and
// Px = psi([zeta; u]), Ksysid.m:1039-1040, 1062): dense  G = Px'Px (symmetric: circulant half of the
// 16-column tiles)  and  C = Px'Py  with v_mfma_f64_4x4x4_4b_f64, built by the rules of kp_gram3_kernel
// (kp_gram3.hip): nothing on the VALU overlaps with the FP64 MFMA stream, LDS reads do up to about one per
// MFMA, so the LDS layout is a compile-time constant, every operand read is `ds_read_b64 vaddr offset:imm`
// (tile loop unrolled by two), the tail mask is the power table's constant entry, and the lift assigns
// (side, column) items to threads for all 8 snapshots of a tile with 16-byte power-table reads.
// A dense product has no weights to reuse an operand register for, and one LDS read per operand and MFMA
// would saturate the LDS (2 x 512 B per 16.5 cycles and SIMD), so a 16 x 16 output tile is formed by 4 MFMAs:
// the B fragment (4 column groups) against the A fragment rotated by 0/4/8/12 lanes inside each 16-lane row
// (DPP row_ror, 6 VALU instructions per k-step and wave); a wave holds one A tile and NT B tiles
// (4 NT accumulators): (1 + NT) / (4 NT) LDS reads per MFMA.
//
// Replaces the per-row lift loop of Ksysid.get_Koopman (Ksysid.m:1030-1065) and the products PxTPx, PxTPy
// (Ksysid.m:1114,1125 / inside `\` :1069) for model_type 'linear' and 'nonlinear'.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kp_internal.h"

#define KT5 8       // snapshots per LDS tile (two k-steps)
#define XW5 224     // columns per side (W <= 224)
// LDS row (doubles): Px [0,224) | Py [224,448) | 16 spare; columns >= W of a side stay zero
#define RS5 464     // = 16 mod 32
#define YOFF5 XW5
#define NIDMAX5 128
#define PST5 10
#define POWBUF5 (PST5 * NIDMAX5)
#define PSIBUF5 (KT5 * RS5)
#define PSI05 (2 * POWBUF5)
#define LDS5_DOUBLES (PSI05 + 2 * PSIBUF5)
#define CPT5 2      // (side, column) items per lifting thread: 2 W <= 512

struct Gram5Args {
  BasisDev b;
  const double* alpha;   // >= 64 doubles of padding behind every array (kp_snapshots_upload)
  const double* beta;
  const double* u;
  int64_t Ns;
  int W;                // columns per side
  int NTL;              // 16-column tiles per side
  int nsuper;           // workgroups per snapshot split
  int ktiles_per_split;
  int D;
  const uint32_t* recipes;   // [W]: columns of Px as products of power-table entries over [zeta, u]
  const uint32_t* desc;      // [njobs][1 + NT]: A tile, then the B tiles (< NTL: Px tile, < 2 NTL: Py tile, 255: padding)
  double* part;              // [nsplit][njobs][NT][4][64]
  int njobs;
};

template <int CTRL>
__device__ __forceinline__ double row_ror5(double v) {       // lane l of a 16-lane row receives from lane (l - n) mod 16
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// NT: B tiles per wave; NF5: single-variable powers per column (3, or 4 for dictionaries with 4-variable monomials)
template <int NT, int NF5>
__global__ __launch_bounds__(256, 2) void kp_gram5_kernel(Gram5Args a) {
  extern __shared__ __align__(16) double sm[];
  const BasisDev& b = a.b;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int super = blockIdx.x % a.nsuper;
  const int split = blockIdx.x / a.nsuper;
  const int job = super * 4 + wave;
  const int nzm = b.nzeta + b.m;
  const int nrawrows = 2 * nzm;
  const int D = a.D;
  const int CID = nrawrows * D;                    // power-table id of the constant 1 (0 for snapshots past Ns)

  // ---- MFMA operand offsets (doubles, Psi buffer 0): row (lane>>4) of k-step 0, column lane & 15 of the tile ----
  const uint32_t* jd = a.desc + (size_t)job * (1 + NT);
  const int lcol = (lane >> 4) * RS5 + (lane & 15);
  const int ao = PSI05 + lcol + 16 * (int)jd[0];
  int bo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int tb = (int)jd[1 + t];
    if (tb >= 2 * a.NTL) tb = 0;                       // padding: any valid tile (the reduce kernel drops it)
    bo[t] = PSI05 + lcol + (tb < a.NTL ? 16 * tb : YOFF5 + 16 * (tb - a.NTL));
  }

  double acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;

  for (int e = tid; e < LDS5_DOUBLES; e += 256) sm[e] = 0.0;

  // ---- lifting items: (side, column) e = tid + 256 i, all KT5 snapshots of the tile ----
  int fa[CPT5][NF5];
  int woff[CPT5];
#pragma unroll
  for (int i = 0; i < CPT5; ++i) {
    const int e = min(tid + 256 * i, 2 * a.W - 1);    // surplus threads repeat the last item (identical writes)
    const int side = e >= a.W ? 1 : 0;
    const int col = e - side * a.W;
    const uint32_t r = a.recipes[col];
#pragma unroll
    for (int f = 0; f < NF5; ++f) {
      const int id = (int)((r >> (8 * f)) & 255u);
      fa[i][f] = (id == 255 ? CID : side * nzm * D + id) * PST5;
    }
    woff[i] = PSI05 + side * YOFF5 + col;
  }

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT5 - 1) / KT5;
  const int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // ---- raw loader; rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)]; value e = tid + j*256 -> (row e/KT5, snapshot e%KT5) ----
  constexpr int LR = 2;
  const int nld = (nrawrows * KT5 + 255) / 256;
  struct RawRegs { double v[LR]; bool ok; };
  bool ld_on[LR];
  const double* ld_ptr[LR];
  const int ld_s = tid & (KT5 - 1);
  const int ld_dst0 = (tid / KT5) * D * PST5 + ld_s;
  int ld_rem = (int)max((int64_t)-1000000, min((int64_t)1 << 30, a.Ns - (kt0 * KT5 + ld_s)));
#pragma unroll
  for (int j = 0; j < LR; ++j) {
    const int e = tid + j * 256;
    ld_on[j] = e < nrawrows * KT5;
    const int r = ld_on[j] ? e / KT5 : 0;
    const int rr = r % nzm;
    const double* src = rr < b.nzeta ? ((r < nzm ? a.alpha : a.beta) + (int64_t)rr * a.Ns) : (a.u + (int64_t)(rr - b.nzeta) * a.Ns);
    ld_ptr[j] = src + kt0 * KT5 + ld_s;
  }
  auto load_raw = [&]() __attribute__((always_inline)) -> RawRegs {
    RawRegs x;
    x.ok = ld_rem > 0;
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      x.v[j] = 0.0;
      if (j < nld) {
        const double v = *ld_ptr[j];
        x.v[j] = x.ok ? v : 0.0;
        ld_ptr[j] += KT5;
      }
    }
    ld_rem -= KT5;
    return x;
  };
  auto store_raw = [&](auto buf_c, const RawRegs& x) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      if (j < nld && ld_on[j]) {
        double* dst = sm + BUF * POWBUF5 + ld_dst0 + j * 32 * D * PST5;
        double p = x.v[j];
        for (int e = 0; e < D; ++e) {
          dst[e * PST5] = p;
          p *= x.v[j];
        }
      }
    }
    if (tid < KT5) sm[BUF * POWBUF5 + CID * PST5 + tid] = x.ok ? 1.0 : 0.0;
  };

  // ---- lift: chunk = (item i, snapshot pair ch); one register set, write of chunk k precedes read of chunk k+1 ----
  constexpr int NCH = CPT5 * (KT5 / 2);
  double2 lf[NF5];
  auto lift_read = [&](int k, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const int i = k / (KT5 / 2), ch = k % (KT5 / 2);
#pragma unroll
    for (int f = 0; f < NF5; ++f) lf[f] = *reinterpret_cast<const double2*>(&sm[BUF * POWBUF5 + fa[i][f] + 2 * ch]);
  };
  auto lift_write = [&](int k, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const int i = k / (KT5 / 2), ch = k % (KT5 / 2);
    double px = (lf[0].x * lf[1].x) * lf[2].x, py = (lf[0].y * lf[1].y) * lf[2].y;
    if (NF5 > 3) { px *= lf[NF5 - 1].x; py *= lf[NF5 - 1].y; }
    sm[BUF * PSIBUF5 + woff[i] + (2 * ch) * RS5] = px;
    sm[BUF * PSIBUF5 + woff[i] + (2 * ch + 1) * RS5] = py;
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  __syncthreads();
  store_raw(B0{}, load_raw());
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    lift_read(k, B0{});
    lift_write(k, B0{});
  }
  store_raw(B1{}, load_raw());
  __syncthreads();

  constexpr int NSTEP = (KT5 / 4) * NT;               // B tiles (4 MFMAs each) per snapshot tile
  constexpr int SP = NSTEP / NCH > 0 ? NSTEP / NCH : 1;
  constexpr int LAG = SP / 2 > 0 ? SP / 2 : 1;
  constexpr int PF = NSTEP < 3 ? NSTEP : 3;

  auto tile = [&](auto cur_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value;
    using NXT = std::integral_constant<int, 1 - CUR>;
    constexpr int PB = CUR * PSIBUF5;
    const RawRegs rawreg = load_raw();
    double bvs[NSTEP];
    double af[4], avn;
#pragma unroll
    for (int i = 0; i < PF; ++i) bvs[i] = sm[PB + (i / NT) * 4 * RS5 + bo[i % NT]];
    avn = sm[PB + ao];
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int kk = step / NT, t = step % NT;
      if (t == 0) {
        af[0] = avn;
        af[1] = row_ror5<0x124>(avn);                  // block blk holds group (blk - 1) & 3
        af[2] = row_ror5<0x128>(avn);
        af[3] = row_ror5<0x12c>(avn);
        if (kk + 1 < KT5 / 4) avn = sm[PB + (kk + 1) * 4 * RS5 + ao];
      }
      if (step + PF < NSTEP) bvs[step + PF] = sm[PB + ((step + PF) / NT) * 4 * RS5 + bo[(step + PF) % NT]];
      const double bv = bvs[step];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[r], bv, acc[t][r], 0, 0, 0);
      if (step >= LAG && (step - LAG) % SP == 0 && (step - LAG) / SP < NCH) lift_write((step - LAG) / SP, NXT{});
      if (step % SP == 0 && step / SP < NCH) lift_read(step / SP, NXT{});
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (k * SP >= NSTEP) lift_read(k, NXT{});
      if (k * SP + LAG >= NSTEP) lift_write(k, NXT{});
    }
    store_raw(cur_c, rawreg);
    __syncthreads();
  };
  {
    int t = 0;
    for (; t + 1 < nkt; t += 2) {
      tile(B0{});
      tile(B1{});
    }
    if (t < nkt) tile(B0{});
  }

  // epilogue: [split][job][t][r][lane]
  double* dst = a.part + (((size_t)split * a.njobs + job) * NT) * 4 * 64 + lane;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(t * 4 + r) * 64] = acc[t][r];
}

// Sums the partials of one (job, B tile, rotation) block vector in split order and scatters its 4 blocks (4x4 each)
// into G and C.  Rotation r: block blk is (A group (blk - r) & 3)' x (B group blk);  D lane: column l & 3, row l >> 4.
__global__ __launch_bounds__(256) void kp_gram5_reduce_kernel(const double* __restrict__ part, int nsplit, int njobs, int NT,
                                                             const uint32_t* __restrict__ desc, int NTL, int W,
                                                             double* __restrict__ G, double* __restrict__ C) {
  const int idx = blockIdx.x;                 // (job*NT + t)*4 + r
  const int r = idx & 3, jt = idx >> 2, t = jt % NT, job = jt / NT;
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;   // 4 waves share the split sum (fixed order: deterministic)
  __shared__ double red4[4][64];
  const size_t per_split = (size_t)njobs * NT * 4 * 64;
  double s = 0.0;
  for (int p = wv; p < nsplit; p += 4) s += part[(size_t)p * per_split + (size_t)idx * 64 + l];
  red4[wv][l] = s;
  __syncthreads();
  if (wv) return;
  s = (red4[0][l] + red4[1][l]) + (red4[2][l] + red4[3][l]);
  const uint32_t* jd = desc + (size_t)job * (1 + NT);
  const int ta = (int)jd[0], tb = (int)jd[1 + t];
  if (tb >= 2 * NTL) return;                  // padding
  const int blk = (l >> 2) & 3;
  const int ia = 16 * ta + 4 * ((blk - r) & 3) + (l >> 4);
  if (ia >= W) return;
  if (tb < NTL) {                             // Px'Px (symmetric)
    const int jb = 16 * tb + 4 * blk + (l & 3);
    if (jb >= W) return;
    if (ta == tb && ia > jb) return;          // diagonal tile: keep the upper half, mirror below (exact symmetry)
    G[(size_t)jb * W + ia] = s;
    G[(size_t)ia * W + jb] = s;
  } else {                                    // Px'Py
    const int jb = 16 * (tb - NTL) + 4 * blk + (l & 3);
    if (jb >= W) return;
    C[(size_t)jb * W + ia] = s;
  }
}

struct kp_gram5_plan {
  int NTL = 0, nt = 0, njobs = 0, nsuper = 0;
  uint32_t* desc = nullptr;      // device
  uint32_t* recipes = nullptr;   // device, [W]
};

void kp_gram5_plan_free(kp_gram5_plan* p) {
  if (!p) return;
  if (p->desc) (void)hipFree(p->desc);
  if (p->recipes) (void)hipFree(p->recipes);
  delete p;
}

static const int kNt5[] = {1, 2, 3, 4, 6, 8, 11, 12};   // 16 tiles (64 accumulators) spill

static int make_plan5(kp_ctx* ctx, const kp_basis* basis, kp_gram5_plan** out) {
  const BasisDev& b = basis->dev;
  const int W = b.W;
  kp_gram5_plan* p = new kp_gram5_plan();
  const int NTL = (W + 15) / 16;
  p->NTL = NTL;
  // A tile g against its circulant half of the Px tiles (g, g+1, ..., antipodal pairs once) and all Py tiles
  std::vector<std::vector<int>> rows(NTL);
  size_t maxr = 0;
  for (int g = 0; g < NTL; ++g) {
    for (int d = 0; d <= NTL / 2; ++d) {
      if (d > 0 && 2 * d == NTL && g >= NTL / 2) continue;
      rows[g].push_back((g + d) % NTL);
    }
    for (int h = 0; h < NTL; ++h) rows[g].push_back(NTL + h);
    maxr = std::max(maxr, rows[g].size());
  }
  // B tiles per wave: rows are cut into jobs of nt tiles; fill whole workgroups (4 waves) and amortise the per-tile
  // lift (cost ~ waves x (MFMA cycles + per-tile VALU share))
  int nt = kNt5[0];
  double best = 1e300;
  for (int c : kNt5) {
    int jobs = 0;
    for (auto& r : rows) jobs += (int)((r.size() + c - 1) / c);
    int waves = (jobs + 3) / 4 * 4;
    double cost = (double)waves * (c * 4 * 33.0 + 350.0);
    if (cost < best) { best = cost; nt = c; }
  }
  if (const char* ov = getenv("KP_GRAM5_NT")) {
    int v = atoi(ov);
    for (int c : kNt5)
      if (c == v) nt = v;
  }
  p->nt = nt;
  std::vector<uint32_t> desc;
  int njobs = 0;
  for (int g = 0; g < NTL; ++g)
    for (size_t q0 = 0; q0 < rows[g].size(); q0 += nt) {
      desc.push_back((uint32_t)g);
      for (int q = 0; q < nt; ++q) desc.push_back(q0 + q < rows[g].size() ? (uint32_t)rows[g][q0 + q] : 255u);
      ++njobs;
    }
  while (njobs % 4) {
    desc.push_back(0u);
    for (int q = 0; q < nt; ++q) desc.push_back(255u);
    ++njobs;
  }
  p->njobs = njobs;
  p->nsuper = njobs / 4;
  // recipes of the W columns of Px over the variables [zeta, u]: psi columns, then (linear models) the inputs
  std::vector<uint32_t> rec(W, 0xffffffffu);
  const int D = basis->pow_depth;
  for (int c = 0; c < W; ++c) {
    if (c < b.nfull) rec[c] = basis->h_recipes[c];
    else rec[c] = 0xffffff00u | (uint32_t)((b.nzeta + (c - b.nfull)) * D);      // u_j to the power 1
  }
  hipError_t e = hipMalloc((void**)&p->desc, desc.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(p->desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&p->recipes, rec.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(p->recipes, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    kp_gram5_plan_free(p);
    return ctx->fail(KP_ERR_HIP, std::string("kp_fit_gram: plan upload: ") + hipGetErrorString(e));
  }
  *out = p;
  return KP_OK;
}

template <int NT, int NF>
static hipError_t launch5b(const Gram5Args& a, int grid, size_t lds, hipStream_t st) {
  static KpLdsCache lds_cache;
  {
    hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_gram5_kernel<NT, NF>, lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((kp_gram5_kernel<NT, NF>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int NT>
static hipError_t launch5(const Gram5Args& a, int nf, int grid, size_t lds, hipStream_t st) {
  return nf > 3 ? launch5b<NT, 4>(a, grid, lds, st) : launch5b<NT, 3>(a, grid, lds, st);
}

bool kp_gram5_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  if (getenv("KP_NO_GRAM5")) return false;
  return (b.model_type == KP_MODEL_LINEAR || b.model_type == KP_MODEL_NONLINEAR) && basis->fast && basis->max_factors <= 4 &&
         b.k_pcs == 0 && b.N == b.nfull && b.W <= XW5 && 2 * b.W <= 256 * CPT5 && 2 * (b.nzeta + b.m) * KT5 <= 2 * 256 &&
         2 * (b.nzeta + b.m) * basis->pow_depth + 1 <= NIDMAX5 && (int)basis->h_recipes.size() >= b.nfull &&
         (b.nzeta + b.m) * basis->pow_depth <= 254;
}

int kp_gram5_launch(kp_ctx* ctx, const kp_basis* basis_c, const kp_snapshots* s, double* GC_dev) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W;
  if (!basis->plan5) {
    int rc = make_plan5(ctx, basis, &basis->plan5);
    if (rc) return rc;
  }
  kp_gram5_plan& plan = *basis->plan5;
  const size_t lds = (size_t)LDS5_DOUBLES * sizeof(double);
  int64_t ktiles = (s->Ns + KT5 - 1) / KT5;
  int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int64_t slots = (int64_t)std::max(8, ncu - ctx->reserve_cus) * 2;      // two workgroups share a CU
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, slots / plan.nsuper > 0 ? slots / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);
  size_t per_split = (size_t)plan.njobs * plan.nt * 4 * 64;
  double* part = (double*)ctx->workspace(4, (size_t)nsplit * per_split * 8);
  if (!part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");

  Gram5Args a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.W = W;
  a.NTL = plan.NTL;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.D = basis->pow_depth;
  a.recipes = plan.recipes;
  a.desc = plan.desc;
  a.part = part;
  a.njobs = plan.njobs;
  const int grid = plan.nsuper * nsplit;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], ctx->stream));
  hipError_t e;
  switch (plan.nt) {
    case 1: e = launch5<1>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 2: e = launch5<2>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 3: e = launch5<3>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 4: e = launch5<4>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 6: e = launch5<6>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 8: e = launch5<8>(a, basis->max_factors, grid, lds, ctx->stream); break;
    case 11: e = launch5<11>(a, basis->max_factors, grid, lds, ctx->stream); break;
    default: e = launch5<12>(a, basis->max_factors, grid, lds, ctx->stream); break;
  }
  KP_HIP(ctx, e);
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], ctx->stream));
  hipLaunchKernelGGL(kp_gram5_reduce_kernel, dim3(plan.njobs * plan.nt * 4), dim3(256), 0, ctx->stream, part, nsplit, plan.njobs, plan.nt,
                     plan.desc, plan.NTL, W, GC_dev, GC_dev + (size_t)W * W);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], ctx->stream));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  ctx->timers[10] = (double)plan.njobs * plan.nt * 512.0;     // executed on the matrix pipe per pair (timer 10)
  return KP_OK;
}

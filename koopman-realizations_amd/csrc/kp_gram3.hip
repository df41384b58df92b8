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
// Replaces the per-row lift loop of Ksysid.get_Koopman (Ksysid.m:1030-1065) and the products
// PxTPx, PxTPy (Ksysid.m:1114,1125) for model_type 'bilinear'.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kp_internal.h"

#ifndef KP_ABL3
#define KP_ABL3 0
#endif
#define KT3 8     // snapshots per LDS tile (two k-steps)
#define CPT3 6    // dictionary columns per lifting thread (16 column lanes => N <= 96)

struct Gram3Args {
  BasisDev b;
  const double* alpha;
  const double* beta;
  const double* u;
  int64_t Ns;
  int RS;               // LDS row stride (doubles): [psi_x (Np4) | psi_y (Np4) | zero group (4) | u (4)] padded to 16 mod 32
  int Np4;              // N rounded up to a multiple of 4
  int nsuper;           // workgroups per snapshot split
  int ktiles_per_split;
  int D;
  const uint32_t* recipes;   // [nfull]
  const uint32_t* desc;      // [njobs][1 + NQ]: A group, then per quad 4 packed B group ids (8 bit each)
  double* part;              // [nsplit][njobs][NQ][NWT][64]
  int njobs;
};

template <int NQ, int BM>
__global__ __launch_bounds__(256, 2) void kp_gram3_kernel(Gram3Args a) {
  constexpr int NWT = (BM + 1) * (BM + 2) / 2;
  extern __shared__ double sm[];
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
  const int RS = a.RS;
  const int UOFF = 2 * a.Np4 + 4;
  // LDS (doubles): pow[2][(nrawrows*D + 1)][KT3] (last row: ones) | psi[2][KT3][RS] | spare row
  const int NID = nrawrows * D + 1;              // power-table entries per snapshot (last: the constant 1)
  const int pow_stride = NID * KT3;                // layout [snapshot][id]: lanes with different ids hit different banks
  const int psi_base = 2 * pow_stride;
  const int psi_stride = KT3 * RS;
  const int trash = 2 * psi_stride;

  // ---- MFMA operand offsets (doubles, relative to the Psi buffer): row (lane>>4) of the k-step ----
  const uint32_t* jd = a.desc + (size_t)job * (1 + NQ);
  const int lrow = (lane >> 4) * RS, blk = (lane >> 2) & 3, lc = lane & 3;
  const int ao = lrow + 4 * (int)jd[0] + lc;             // A group replicated over the 4 blocks
  int bo[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) bo[q] = lrow + 4 * (int)((jd[1 + q] >> (8 * blk)) & 255u) + lc;
  const int uo = lrow + UOFF;

  double acc[NQ][NWT];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NWT; ++w) acc[q][w] = 0.0;

  // ---- one-time LDS setup: Psi buffers zero (padding columns and the zero group stay zero) ----
  for (int e = tid; e < 2 * psi_stride + RS; e += 256) sm[psi_base + e] = 0.0;
  if (tid < 2 * KT3) sm[(tid / KT3) * pow_stride + (tid % KT3) * NID + nrawrows * D] = 1.0;

  // ---- lifting thread constants ----
  const int jl = tid & 15, combo = tid >> 4, ls = combo & (KT3 - 1), lside = combo >> 3;
  int foff[CPT3][4];
  int woff[CPT3];
  bool wok[CPT3];
#pragma unroll
  for (int i = 0; i < CPT3; ++i) {
    const int c = jl + 16 * i;
    wok[i] = c < b.nfull;
    const uint32_t r = wok[i] ? a.recipes[c] : 0xffffffffu;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int id = (int)((r >> (8 * f)) & 255u);
      foff[i][f] = ls * NID + (id == 255 ? nrawrows * D : lside * nzm * D + id);
    }
    woff[i] = wok[i] ? ls * RS + lside * a.Np4 + c : trash + jl;
  }
  const int uoff_pow = ls * NID + b.nzeta * D;     // + j*D : u_j (e = 1) of snapshot ls in the power table

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT3 - 1) / KT3;
  const int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // ---- raw loader; rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)] ----
  // raw loader: value e = tid + j*256 of the tile -> (row e / KT3, snapshot e % KT3), up to LR per thread
  constexpr int LR = 3;
  struct RawRegs { double v[LR]; };
  bool ld_on[LR];
  int ld_r[LR], ld_s[LR];
  const double* ld_src[LR];
#pragma unroll
  for (int j = 0; j < LR; ++j) {
    const int e = tid + j * 256;
    ld_on[j] = e < nrawrows * KT3;
    ld_r[j] = e / KT3;
    ld_s[j] = e % KT3;
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
      int64_t i = kt * KT3 + ld_s[j];
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

  // ---- lift of snapshot tile kt: power-table buffer rb -> Psi buffer pb, in pipelined chunks ----
  double vmask = 0.0;
  double lf[CPT3][4];
  auto lift_begin = [&](int rb, int pb, int64_t kt) {
    const double* T = sm + rb * pow_stride;
    vmask = (kt * KT3 + ls) < a.Ns ? 1.0 : 0.0;
    if (lside == 0 && jl < BM) {                    // u_j of this snapshot next to the Psi row
      double* P = sm + psi_base + pb * psi_stride;
      P[ls * RS + UOFF + jl] = T[uoff_pow + jl * D] * vmask;
    }
  };
  auto lift_read = [&](int i, int rb) {
    const double* T = sm + rb * pow_stride;
#pragma unroll
    for (int f = 0; f < 4; ++f) lf[i][f] = T[foff[i][f]];
  };
  auto lift_write = [&](int i, int pb) {
    double* P = sm + psi_base + (wok[i] ? pb * psi_stride : 0) + woff[i];
    P[0] = (lf[i][0] * lf[i][1]) * (lf[i][2] * lf[i][3]) * vmask;
  };

  store_raw(0, load_raw(kt0));
  __syncthreads();
  lift_begin(0, 0, kt0);
#pragma unroll
  for (int i = 0; i < CPT3; ++i) {
    lift_read(i, 0);
    lift_write(i, 0);
  }
  store_raw(1, load_raw(kt0 + 1));
  __syncthreads();

  constexpr int NSTEP = (KT3 / 4) * NQ;                 // quad steps (NWT MFMAs each) per snapshot tile
  constexpr int SP = NSTEP / CPT3 > 0 ? NSTEP / CPT3 : 1;
  constexpr int LAG = SP / 2 > 0 ? SP / 2 : 1;
  constexpr int PF = NSTEP < 3 ? NSTEP : 3;
  for (int t = 0; t < nkt; ++t) {
    const RawRegs rawreg = load_raw(kt0 + t + 2);
    const int cur = t & 1, nxt = cur ^ 1;
    const double* P = sm + psi_base + cur * psi_stride;
    lift_begin(nxt, nxt, kt0 + t + 1);
    {
      double bvs[NSTEP];
      double aw[NWT];
      double avn, utn[3] = {0.0, 0.0, 0.0};              // raw A fragment and u of the NEXT k-step (prefetched)
#pragma unroll
      for (int i = 0; i < PF; ++i) bvs[i] = P[(i / NQ) * 4 * RS + bo[i % NQ]];
      avn = P[ao];
#pragma unroll
      for (int j = 0; j < BM; ++j) utn[j] = P[uo + j];
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int kk = step / NQ, q = step % NQ;
        if (q == 0) {
          // weighted A fragments of this k-step: w_ab = ut_a ut_b, ut = [1; u]
          double ut[4];
          ut[0] = 1.0;
#pragma unroll
          for (int j = 0; j < BM; ++j) ut[1 + j] = utn[j];
          int w = 0;
#pragma unroll
          for (int x = 0; x <= BM; ++x)
#pragma unroll
            for (int y = x; y <= BM; ++y) {
              aw[w] = x == 0 ? (y == 0 ? avn : avn * ut[y]) : avn * (ut[x] * ut[y]);
              ++w;
            }
          if (kk + 1 < KT3 / 4) {
            const double* Pn = P + (kk + 1) * 4 * RS;
            avn = Pn[ao];
#pragma unroll
            for (int j = 0; j < BM; ++j) utn[j] = Pn[uo + j];
          }
        }
#if KP_ABL3 != 4 && KP_ABL3 != 6
        if (step + PF < NSTEP) bvs[step + PF] = P[((step + PF) / NQ) * 4 * RS + bo[(step + PF) % NQ]];
        const double bv = bvs[step];
#else
        const double bv = bvs[step % PF];
#endif
#if KP_ABL3 != 3
#pragma unroll
        for (int w = 0; w < NWT; ++w) acc[q][w] = __builtin_amdgcn_mfma_f64_4x4x4f64(aw[w], bv, acc[q][w], 0, 0, 0);
#else
        acc[q][0] += bv * aw[0];
#endif
#if KP_ABL3 != 1 && KP_ABL3 != 6
        if (step % SP == 0 && step / SP < CPT3) lift_read(step / SP, nxt);
        if (step >= LAG && (step - LAG) % SP == 0 && (step - LAG) / SP < CPT3) lift_write((step - LAG) / SP, nxt);
#endif
      }
#if KP_ABL3 != 1 && KP_ABL3 != 6
#pragma unroll
      for (int i = 0; i < CPT3; ++i) {
        if (i * SP >= NSTEP) lift_read(i, nxt);
        if (i * SP + LAG >= NSTEP) lift_write(i, nxt);
      }
#endif
    }
    store_raw(cur, rawreg);
#if KP_ABL3 != 5 && KP_ABL3 != 6
    __syncthreads();
#endif
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
__global__ __launch_bounds__(64) void kp_gram3_reduce_kernel(const double* __restrict__ part, int nsplit, int njobs, int NQ, int NWT,
                                                             int BM, const uint32_t* __restrict__ desc, int G4, int N, int W,
                                                             double* __restrict__ G, double* __restrict__ C) {
  const int idx = blockIdx.x;                 // (job*NQ + q)*NWT + w
  const int w = idx % NWT, jq = idx / NWT, q = jq % NQ, job = jq / NQ;
  const int l = threadIdx.x;
  const size_t per_split = (size_t)njobs * NQ * NWT * 64;
  double s = 0.0;
  for (int p = 0; p < nsplit; ++p) s += part[(size_t)p * per_split + (size_t)idx * 64 + l];
  const uint32_t* jd = desc + (size_t)job * (1 + NQ);
  const int ga = (int)jd[0];
  const int gb = (int)((jd[1 + q] >> (8 * ((l >> 2) & 3))) & 255u);
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
  int G4 = 0, Np4 = 0, RS = 0, nq = 0, njobs = 0, nsuper = 0;
  uint32_t* desc = nullptr;  // device
};

void kp_gram3_plan_free(kp_gram3_plan* p) {
  if (!p) return;
  if (p->desc) (void)hipFree(p->desc);
  delete p;
}

// Row g of psi_x (one 4-column group) is paired with: its circulant half of the psi_x groups
// (g, g+1, ..., g+floor(G4/2) mod G4; antipodal pairs once) and all G4 groups of psi_y.
static int make_plan3(kp_ctx* ctx, int N, kp_gram3_plan** out) {
  kp_gram3_plan* p = new kp_gram3_plan();
  const int G4 = (N + 3) / 4;
  p->G4 = G4;
  p->Np4 = 4 * G4;
  int rs = 2 * p->Np4 + 8;
  while (rs % 32 != 16) rs += 4;
  p->RS = rs;
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
  static const int cand[] = {1, 2, 3, 4, 6, 8};
  int nq = 8;
  for (int c : cand)
    if ((size_t)c >= maxq) { nq = c; break; }
  if (const char* ov = getenv("KP_GRAM3_NQ")) {   // tuning override: rows are cut into jobs of <= nq quads
    int v = atoi(ov);
    for (int c : cand)
      if (c == v) nq = v;
  }
  p->nq = nq;
  std::vector<uint32_t> desc;
  int njobs = 0;
  for (int g = 0; g < G4; ++g) {
    const size_t nquads = (rows[g].size() + 3) / 4;
    for (size_t q0 = 0; q0 < nquads; q0 += nq) {          // rows longer than nq quads become several jobs
      desc.push_back((uint32_t)g);
      for (int q = 0; q < nq; ++q) {
        uint32_t packed = 0;
        for (int k = 0; k < 4; ++k) {
          size_t idx = (q0 + q) * 4 + k;
          int gb = ((size_t)(q0 + q) < nquads && idx < rows[g].size()) ? rows[g][idx] : ZG;
          packed |= (uint32_t)gb << (8 * k);
        }
        desc.push_back(packed);
      }
      ++njobs;
    }
  }
  while (njobs % 4) {
    desc.push_back(0u);
    uint32_t z = (uint32_t)ZG * 0x01010101u;
    for (int q = 0; q < nq; ++q) desc.push_back(z);
    ++njobs;
  }
  p->njobs = njobs;
  p->nsuper = njobs / 4;
  hipError_t e = hipMalloc((void**)&p->desc, desc.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(p->desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    kp_gram3_plan_free(p);
    return ctx->fail(KP_ERR_HIP, std::string("kp_fit_gram: plan upload: ") + hipGetErrorString(e));
  }
  *out = p;
  return KP_OK;
}

template <int NQ, int BM>
static hipError_t launch3b(const Gram3Args& a, int grid, size_t lds, hipStream_t st) {
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kp_gram3_kernel<NQ, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set = lds;
  }
  hipLaunchKernelGGL((kp_gram3_kernel<NQ, BM>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int NQ>
static hipError_t launch3(const Gram3Args& a, int bm, int grid, size_t lds, hipStream_t st) {
  switch (bm) {
    case 1: return launch3b<NQ, 1>(a, grid, lds, st);
    case 2: return launch3b<NQ, 2>(a, grid, lds, st);
    default: return launch3b<NQ, 3>(a, grid, lds, st);
  }
}

bool kp_gram3_applicable(const kp_basis* basis) {
  const BasisDev& b = basis->dev;
  if (getenv("KP_NO_GRAM3")) return false;
  return b.model_type == KP_MODEL_BILINEAR && basis->fast && b.k_pcs == 0 && b.nfull <= 16 * CPT3 && b.m >= 1 && b.m <= 3 &&
         2 * (b.nzeta + b.m) * KT3 <= 3 * 256 && 2 * ((b.nfull + 3) / 4) < 255;
}

int kp_gram3_launch(kp_ctx* ctx, const kp_basis* basis_c, const kp_snapshots* s, double* GC_dev) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W, N = b.N;
  if (!basis->plan3) {
    int rc = make_plan3(ctx, N, &basis->plan3);
    if (rc) return rc;
  }
  kp_gram3_plan& plan = *basis->plan3;
  const int D = basis->pow_depth;
  const int nraw = 2 * (b.nzeta + b.m);
  const int BM = b.m, NWT = (BM + 1) * (BM + 2) / 2;
  size_t lds = ((size_t)2 * (nraw * D + 1) * KT3 + (size_t)2 * KT3 * plan.RS + plan.RS) * sizeof(double);
  if (lds > 160 * 1024) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: dictionary too wide");
  int64_t ktiles = (s->Ns + KT3 - 1) / KT3;
  int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int wg_per_cu = 2;                                  // __launch_bounds__(256, 2): two workgroups share a CU
  if (const char* ov = getenv("KP_GRAM3_WGPCU")) wg_per_cu = std::max(1, atoi(ov));
  int64_t slots = (int64_t)std::max(8, ncu - ctx->reserve_cus) * wg_per_cu;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, slots / plan.nsuper > 0 ? slots / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);
  size_t per_split = (size_t)plan.njobs * plan.nq * NWT * 64;
  double* part = (double*)ctx->workspace(4, (size_t)nsplit * per_split * 8);
  if (!part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");

  Gram3Args a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.RS = plan.RS;
  a.Np4 = plan.Np4;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.D = D;
  a.recipes = (const uint32_t*)basis->d_recipes;
  a.desc = plan.desc;
  a.part = part;
  a.njobs = plan.njobs;
  const int grid = plan.nsuper * nsplit;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], ctx->stream));
  hipError_t e;
  switch (plan.nq) {
    case 1: e = launch3<1>(a, BM, grid, lds, ctx->stream); break;
    case 2: e = launch3<2>(a, BM, grid, lds, ctx->stream); break;
    case 3: e = launch3<3>(a, BM, grid, lds, ctx->stream); break;
    case 4: e = launch3<4>(a, BM, grid, lds, ctx->stream); break;
    case 6: e = launch3<6>(a, BM, grid, lds, ctx->stream); break;
    default: e = launch3<8>(a, BM, grid, lds, ctx->stream); break;
  }
  KP_HIP(ctx, e);
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], ctx->stream));
  hipLaunchKernelGGL(kp_gram3_reduce_kernel, dim3(plan.njobs * plan.nq * NWT), dim3(64), 0, ctx->stream, part, nsplit, plan.njobs,
                     plan.nq, NWT, BM, plan.desc, plan.G4, N, W, GC_dev, GC_dev + (size_t)W * W);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], ctx->stream));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  return KP_OK;
}

// Fused lift + Gram kernel:  G = Px'Px, C = Px'Py  without materialising Px, Py in HBM.
// Replaces the per-row lift loop of Ksysid.get_Koopman (Ksysid.m:1030-1065) and the
// accumulations of solve_KoopmanQP (Ksysid.m:1114,1125).
//
// Decomposition (gfx950): the [G | C] output is cut into 16x16 f64 tiles (G: upper
// triangle only).  A wave owns NACC tiles as v_mfma_f64_16x16x4_f64 accumulators held in
// registers for its whole snapshot range; a 4-wave workgroup (one wave per SIMD) shares
// one LDS-staged tile of lifted observables Psi_x | Psi_y (KT snapshots x Wp columns,
// double buffered, regenerated on the fly from the raw snapshot columns) and the grid is
// (output super-tiles) x (snapshot splits).  Per-split partial tiles are combined by a
// second kernel in a fixed order, so results are bitwise reproducible run to run.
#include <algorithm>

#include "kp_internal.h"

#ifndef KP_ABLATE
#define KP_ABLATE 0
#endif
#define KT 8  // snapshots per LDS tile (two k-steps of the 16x16x4 MFMA)

typedef double double4_t __attribute__((ext_vector_type(4)));

struct GramArgs {
  BasisDev b;
  const double* alpha;
  const double* beta;
  const double* u;
  int64_t Ns;
  int Wp;               // padded row length of the Psi tiles (doubles), == 16 (mod 32)
  int nsuper;           // workgroups per snapshot split
  int ktiles_per_split; // KT-snapshot tiles per split
  int D;                // depth of the power table (max exponent), >= 1
  int pcs_in_lds;       // copy the pcs matrix into LDS
  const uint32_t* recipes;  // [nfull] 4 x 8-bit factor ids (fast path)
  const uint32_t* desc; // [nsuper*4][NACC]  a_off | b_off << 16  (doubles, rel. to buffer)
  const int* tile_out;  // [nsuper*4][NACC]  output tile id or -1
  double* part;         // [nsplit][ntile_out][4][64]
  int ntile_out;
};

// LDS carve-up (in doubles)
struct GramLds {
  int pow;      // [2][nrawrows][D][KT] power table x^e (e = 1..D) of the raw variables
  int ones;     // [KT] ones
  int rec;      // [nfull] recipes (uint32, two per double)
  int pcs;      // [k_pcs][nfull] (optional)
  int full;     // [2 sides][nfull][KT]   (only with pcs)
  int psi;      // [2][2 sides][KT][Wp]
  int total;
};

static __host__ __device__ inline GramLds gram_lds(const BasisDev& b, int Wp, int D, int pcs_in_lds) {
  GramLds l;
  int nraw = 2 * (b.nzeta + b.m);
  l.pow = 0;
  l.ones = l.pow + 2 * nraw * D * KT;
  l.rec = l.ones + KT;
  l.pcs = l.rec + (b.nfull + 1) / 2;
  l.full = l.pcs + (pcs_in_lds ? b.k_pcs * b.nfull : 0);
  l.psi = l.full + (b.k_pcs ? 2 * b.nfull * KT : 0);
  l.psi = (l.psi + 1) & ~1;
  l.total = l.psi + 2 * 2 * KT * Wp;
  return l;
}

// FAST: every full-basis column is a product of <= 4 powers of single variables (monomial
// dictionaries): value = prod_f pow[id_f], ids packed in a 32-bit recipe (255 = 1.0).
template <int NACC, bool FAST>
__global__ __launch_bounds__(256, 1) void kp_gram_kernel(GramArgs a) {
  extern __shared__ double sm[];
  const BasisDev& b = a.b;
  const GramLds L = gram_lds(b, a.Wp, a.D, a.pcs_in_lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int super = blockIdx.x % a.nsuper;
  const int split = blockIdx.x / a.nsuper;
  const int job = super * 4 + wave;
  const int nrawrows = 2 * (b.nzeta + b.m);
  const int nzm = b.nzeta + b.m;
  const int D = a.D;

  // per-tile operand offsets (doubles, relative to the Psi buffer), lane part pre-added
  const int lane_off = (lane >> 4) * a.Wp + (lane & 15);
  int ao[NACC], bo[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    uint32_t d = a.desc[job * NACC + t];
    ao[t] = lane_off + (int)(d & 0xffffu);
    bo[t] = lane_off + (int)(d >> 16);
  }

  double4_t acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // one-time LDS setup: zero both Psi buffers (padding columns [W, Wp) are never written
  // again), ones row, recipes, pcs
  for (int e = tid; e < 2 * 2 * KT * a.Wp; e += 256) sm[L.psi + e] = 0.0;
  if (tid < KT) sm[L.ones + tid] = 1.0;
  uint32_t* rec = (uint32_t*)(sm + L.rec);
  if (FAST)
    for (int e = tid; e < b.nfull; e += 256) rec[e] = a.recipes[e];
  if (a.pcs_in_lds)
    for (int e = tid; e < b.k_pcs * b.nfull; e += 256) sm[L.pcs + e] = b.pcs[e];
  const double* pcs = a.pcs_in_lds ? (const double*)(sm + L.pcs) : b.pcs;

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT - 1) / KT;
  int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // raw tile loader: thread e < nrawrows*KT handles (row r, snapshot s);
  // rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)]
  // raw loader: value e = tid + j*256 of the tile -> (row e / KT, snapshot e % KT), up to LR per thread
  constexpr int LR = 3;
  struct RawRegs { double v[LR]; };
  bool ld_on[LR];
  int ld_r[LR], ld_s[LR];
  const double* ld_src[LR];
#pragma unroll
  for (int j = 0; j < LR; ++j) {
    const int e = tid + j * 256;
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
      double* dst = sm + L.pow + ((buf * nrawrows + ld_r[j]) * D) * KT + ld_s[j];
      double p = x.v[j];
      for (int e = 0; e < D; ++e) {
        dst[e * KT] = p;
        p *= x.v[j];
      }
    }
  };

  // lift of one KT tile from power-table buffer rb into Psi buffer pb
  const int jl = tid & 15, combo = tid >> 4, ls = combo & (KT - 1), lside = combo >> 3;
  auto lift_tile = [&](int rb, int pb, int64_t kt) {
    const double* tab = sm + L.pow + rb * nrawrows * D * KT;
    const double* side_tab = tab + lside * nzm * D * KT + ls;     // + id*KT -> x_v^e of this side / snapshot
    const double* ones = sm + L.ones;
    double* psi_row = sm + L.psi + (pb * 2 + lside) * KT * a.Wp + ls * a.Wp;
    const bool valid = (kt * KT + ls) < a.Ns;
    double uv[4] = {0.0, 0.0, 0.0, 0.0};
    const bool bil = b.model_type == KP_MODEL_BILINEAR;
    if (bil && b.m <= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < b.m) uv[i] = tab[(b.nzeta + i) * D * KT + ls];
    }
    auto eval_full = [&](int c) -> double {
      if (FAST) {
        uint32_t r = rec[c];
        double v = 1.0;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          uint32_t id = (r >> (8 * f)) & 255u;
          const double* src = id == 255u ? ones : side_tab + id * KT;
          v *= *src;
        }
        return v;
      } else {
        // generic columns (fourier, gaussian, high-factor monomials): variables are the e = 1 entries
        return kp_eval_col(b, b.cols[c], side_tab, D * KT);
      }
    };
    auto put = [&](int c, double val) {
      psi_row[c] = val;
      if (bil) {
        if (b.m <= 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i < b.m) psi_row[(i + 1) * b.N + c] = val * uv[i];
        } else {
          for (int i = 0; i < b.m; ++i) psi_row[(i + 1) * b.N + c] = val * tab[(b.nzeta + i) * D * KT + ls];
        }
      }
    };
    if (b.k_pcs == 0) {
      for (int c = jl; c < b.nfull; c += 16) put(c, valid ? eval_full(c) : 0.0);
    } else {
      double* full = sm + L.full + lside * b.nfull * KT;
      for (int c = jl; c < b.nfull; c += 16) full[c * KT + ls] = valid ? eval_full(c) : 0.0;
      __syncthreads();
      for (int c = jl; c < b.N; c += 16) {
        double val;
        if (c < b.nvars)
          val = side_tab[c * D * KT];
        else if (c < b.nvars + b.k_pcs) {
          const double* pc = pcs + (size_t)(c - b.nvars) * b.nfull;
          val = 0.0;
          for (int i = 0; i < b.nfull; ++i) val += pc[i] * full[i * KT + ls];
        } else
          val = 1.0;
        put(c, valid ? val : 0.0);
      }
    }
    if (b.model_type == KP_MODEL_LINEAR) {  // [psi , u]
      for (int i = jl; i < b.m; i += 16) psi_row[b.N + i] = valid ? tab[(b.nzeta + i) * D * KT + ls] : 0.0;
    }
  };

  // prologue: raw tile 0 -> power table, lift it, raw tile 1 -> power table
  store_raw(0, load_raw(kt0));
  __syncthreads();
  if (nkt > 0) lift_tile(0, 0, kt0);
  store_raw(1, load_raw(kt0 + 1));
  __syncthreads();

  for (int t = 0; t < nkt; ++t) {
    // (1) prefetch raw tile t+2 into a register
    const RawRegs rawreg = load_raw(kt0 + t + 2);
    // (2) lift tile t+1 into the other Psi buffer
#if KP_ABLATE != 1
    if (t + 1 < nkt) lift_tile((t + 1) & 1, (t + 1) & 1, kt0 + t + 1);
#endif
    // (3) MFMA over tile t
    const double* P = sm + L.psi + (t & 1) * 2 * KT * a.Wp;
#if KP_ABLATE != 2
    {
      // software-pipelined operand fetch: LDS reads run PF MFMAs ahead of their use
      constexpr int NM = (KT / 4) * NACC;
      constexpr int PF = 3;
      double av[NM], bv[NM];
#pragma unroll
      for (int i = 0; i < PF && i < NM; ++i) {
        const double* Pk = P + (i / NACC) * 4 * a.Wp;
        av[i] = Pk[ao[i % NACC]];
        bv[i] = Pk[bo[i % NACC]];
      }
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (i + PF < NM) {
          const double* Pk = P + ((i + PF) / NACC) * 4 * a.Wp;
          av[i + PF] = Pk[ao[(i + PF) % NACC]];
          bv[i + PF] = Pk[bo[(i + PF) % NACC]];
        }
        acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc[i % NACC], 0, 0, 0);
      }
    }
#endif
    // (4) raw tile t+2 -> power-table buffer t&1 (last read while lifting tile t)
    store_raw(t & 1, rawreg);
    __syncthreads();
  }

  // epilogue: partial tiles, [split][tile][reg][lane]
#pragma unroll
  for (int q = 0; q < NACC; ++q) {
    int to = a.tile_out[job * NACC + q];
    if (to >= 0) {
      double* dst = a.part + ((size_t)split * a.ntile_out + to) * 256 + lane;
      dst[0] = acc[q][0];
      dst[64] = acc[q][1];
      dst[128] = acc[q][2];
      dst[192] = acc[q][3];
    }
  }
}

// Sums the per-split partials of each output tile in split order and scatters into the
// column-major G (both triangles) and C.
__global__ __launch_bounds__(256) void kp_gram_reduce_kernel(const double* __restrict__ part, int nsplit, int ntile_out,
                                                             const int* __restrict__ tile_info /* [ntile][3] kind,tr,tc */, int W,
                                                             double* __restrict__ G, double* __restrict__ C) {
  const int tile = blockIdx.x;
  const int t = threadIdx.x;
  double s = 0.0;
  for (int p = 0; p < nsplit; ++p) s += part[((size_t)p * ntile_out + tile) * 256 + t];
  const int reg = t >> 6, lane = t & 63;
  const int kind = tile_info[tile * 3], tr = tile_info[tile * 3 + 1], tc = tile_info[tile * 3 + 2];
  const int i = tr * 16 + (lane >> 4) + 4 * reg;  // v_mfma_f64_16x16x4: row = (lane>>4) + 4*reg, col = lane&15
  const int j = tc * 16 + (lane & 15);
  if (i < W && j < W) {
    if (kind == 0) {
      G[(size_t)j * W + i] = s;
      if (tr != tc) G[(size_t)i * W + j] = s;
    } else {
      C[(size_t)j * W + i] = s;
    }
  }
}

// Host plan: tile -> wave assignment and device tables, built once per dictionary.
struct kp_gram_plan {
  int nt = 0, Wp = 0, nacc = 0, njobs = 0, nsuper = 0, ntile_out = 0;
  char* tab = nullptr;   // device: desc | tile_out | tile_info
  size_t off_to = 0, off_ti = 0;
  bool attr_set = false;
};

void kp_gram_plan_free(kp_gram_plan* p) {
  if (!p) return;
  if (p->tab) (void)hipFree(p->tab);
  delete p;
}

static int make_plan(kp_ctx* ctx, int W, kp_gram_plan** out) {
  kp_gram_plan* p = new kp_gram_plan();
  p->nt = (W + 15) / 16;
  int wp = p->nt * 16;
  while (wp % 32 != 16) wp += 16;  // conflict-free ds_read_b64 of rows k, k+1
  p->Wp = wp;
  int ntile = p->nt * (p->nt + 1) / 2 + p->nt * p->nt;
  p->ntile_out = ntile;
  static const int cand[] = {8, 16, 24, 28, 32};
  int best = 8;
  long best_cost = -1;
  for (int c : cand) {
    long slots = (long)((ntile + 4 * c - 1) / (4 * c)) * 4 * c;
    if (best_cost < 0 || slots < best_cost || (slots == best_cost && c > best)) {
      best_cost = slots;
      best = c;
    }
  }
  p->nacc = best;
  p->nsuper = (ntile + 4 * best - 1) / (4 * best);
  p->njobs = p->nsuper * 4;
  std::vector<uint32_t> desc((size_t)p->njobs * best, 0u);
  std::vector<int> tile_out((size_t)p->njobs * best, -1), tile_info;
  int id = 0;
  auto push = [&](int kind, int tr, int tc) {
    uint32_t a_off = (uint32_t)(tr * 16);
    uint32_t b_off = (uint32_t)((kind ? KT * p->Wp : 0) + tc * 16);
    // deal tiles round-robin over jobs so every wave carries the same load
    int job = id % p->njobs, slot = id / p->njobs;
    desc[(size_t)job * best + slot] = a_off | (b_off << 16);
    tile_out[(size_t)job * best + slot] = id;
    tile_info.push_back(kind);
    tile_info.push_back(tr);
    tile_info.push_back(tc);
    ++id;
  };
  for (int tr = 0; tr < p->nt; ++tr)
    for (int tc = tr; tc < p->nt; ++tc) push(0, tr, tc);
  for (int tr = 0; tr < p->nt; ++tr)
    for (int tc = 0; tc < p->nt; ++tc) push(1, tr, tc);
  size_t b_desc = desc.size() * 4, b_to = tile_out.size() * 4, b_ti = tile_info.size() * 4;
  p->off_to = b_desc;
  p->off_ti = b_desc + b_to;
  hipError_t e = hipMalloc((void**)&p->tab, b_desc + b_to + b_ti);
  if (e == hipSuccess) e = hipMemcpy(p->tab, desc.data(), b_desc, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->tab + p->off_to, tile_out.data(), b_to, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->tab + p->off_ti, tile_info.data(), b_ti, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    kp_gram_plan_free(p);
    return ctx->fail(KP_ERR_HIP, std::string("kp_fit_gram: plan upload: ") + hipGetErrorString(e));
  }
  *out = p;
  return KP_OK;
}

template <int NACC, bool FAST>
static hipError_t launch_gram(const GramArgs& a, int grid, size_t lds, hipStream_t st, bool) {
  static KpLdsCache lds_cache;   // per instantiation and device: largest dynamic-LDS size granted so far
  {
    hipError_t e = kp_ensure_lds(lds_cache, (const void*)kp_gram_kernel<NACC, FAST>, lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((kp_gram_kernel<NACC, FAST>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <bool FAST>
static hipError_t launch_gram_n(int nacc, const GramArgs& a, int grid, size_t lds, hipStream_t st, bool set_attr) {
  switch (nacc) {
    case 8: return launch_gram<8, FAST>(a, grid, lds, st, set_attr);
    case 16: return launch_gram<16, FAST>(a, grid, lds, st, set_attr);
    case 24: return launch_gram<24, FAST>(a, grid, lds, st, set_attr);
    case 28: return launch_gram<28, FAST>(a, grid, lds, st, set_attr);
    default: return launch_gram<32, FAST>(a, grid, lds, st, set_attr);
  }
}

int kp_gram_launch(kp_ctx* ctx, const kp_basis* basis_c, const kp_snapshots* s, double* GC_dev) {
  kp_basis* basis = const_cast<kp_basis*>(basis_c);
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W;
  if (!basis->plan) {
    int rc = make_plan(ctx, W, &basis->plan);
    if (rc) return rc;
  }
  kp_gram_plan& plan = *basis->plan;
  if ((uint32_t)(2 * KT * plan.Wp) > 65535u) return kp_gram_wide_launch(ctx, basis, s, GC_dev);
  const bool fast = basis->fast;
  const int D = fast ? basis->pow_depth : 1;
  const int pcs_in_lds = (b.k_pcs > 0 && (size_t)b.k_pcs * b.nfull * 8 <= 40 * 1024) ? 1 : 0;
  GramLds L = gram_lds(b, plan.Wp, D, pcs_in_lds);
  size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) return kp_gram_wide_launch(ctx, basis, s, GC_dev);   // W > ~580: lifted panels in HBM + TN products (kp_wide.hip)
  if (2 * (b.nzeta + b.m) * KT > 3 * 256) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: too many raw columns");
  int64_t ktiles = (s->Ns + KT - 1) / KT;
  int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, ncu / plan.nsuper > 0 ? ncu / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);

  size_t b_part = (size_t)nsplit * plan.ntile_out * 256 * 8;
  double* part = (double*)ctx->workspace(4, b_part);
  if (!part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");

  GramArgs a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.Wp = plan.Wp;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.D = D;
  a.pcs_in_lds = pcs_in_lds;
  a.recipes = (const uint32_t*)basis->d_recipes;
  a.desc = (const uint32_t*)plan.tab;
  a.tile_out = (const int*)(plan.tab + plan.off_to);
  a.part = part;
  a.ntile_out = plan.ntile_out;
  int grid = plan.nsuper * nsplit;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], ctx->stream));
  hipError_t e = fast ? launch_gram_n<true>(plan.nacc, a, grid, lds, ctx->stream, !plan.attr_set)
                      : launch_gram_n<false>(plan.nacc, a, grid, lds, ctx->stream, !plan.attr_set);
  KP_HIP(ctx, e);
  plan.attr_set = true;
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], ctx->stream));
  hipLaunchKernelGGL(kp_gram_reduce_kernel, dim3(plan.ntile_out), dim3(256), 0, ctx->stream, part, nsplit, plan.ntile_out,
                     (const int*)(plan.tab + plan.off_ti), W, GC_dev, GC_dev + (size_t)W * W);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], ctx->stream));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  ctx->timers[10] = (double)plan.njobs * plan.nacc * 512.0;   // executed on the matrix pipe per pair: 16 x 16 tiles incl. padding (timer 10)
  return KP_OK;
}

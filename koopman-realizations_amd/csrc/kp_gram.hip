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
  const uint32_t* desc; // [nsuper*4][NACC]  a_off | b_off << 16  (doubles, rel. to buffer)
  const int* tile_out;  // [nsuper*4][NACC]  output tile id or -1
  double* part;         // [nsplit][ntile_out][4][64]
  int ntile_out;
};

// LDS carve-up (doubles)
struct GramLds {
  int raw;      // [2][nrawrows][KT]
  int full;     // [2 sides][nfull][KT]   (only with pcs)
  int psi;      // [2][2 sides][KT][Wp]
  int total;
};

static __host__ __device__ inline GramLds gram_lds(const BasisDev& b, int Wp) {
  GramLds l;
  int nraw = 2 * (b.nzeta + b.m);
  l.raw = 0;
  l.full = l.raw + 2 * nraw * KT;
  l.psi = l.full + (b.k_pcs ? 2 * b.nfull * KT : 0);
  l.total = l.psi + 2 * 2 * KT * Wp;
  return l;
}

// Writes psi_econ column c (value val, snapshot s of side `side`) with the model-type
// expansion of Ksysid.m:1034-1064 into the Psi tile.
__device__ __forceinline__ void put_psi(const BasisDev& b, double* psi_side, int Wp, int s, int c, double val,
                                        const double* uvals /* [m][KT] */) {
  double* row = psi_side + s * Wp;
  row[c] = val;
  if (b.model_type == KP_MODEL_BILINEAR) {
    for (int i = 0; i < b.m; ++i) row[(i + 1) * b.N + c] = val * uvals[i * KT + s];
  }
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void kp_gram_kernel(GramArgs a) {
  extern __shared__ double sm[];
  const BasisDev& b = a.b;
  const GramLds L = gram_lds(b, a.Wp);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int super = blockIdx.x % a.nsuper;
  const int split = blockIdx.x / a.nsuper;
  const int job = super * 4 + wave;
  const int nrawrows = 2 * (b.nzeta + b.m);
  const int nzm = b.nzeta + b.m;

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

  // zero both Psi buffers once: padding columns [W, Wp) are never written again
  for (int e = tid; e < 2 * 2 * KT * a.Wp; e += 256) sm[L.psi + e] = 0.0;

  const int64_t kt0 = (int64_t)split * a.ktiles_per_split;
  const int64_t ktiles_total = (a.Ns + KT - 1) / KT;
  int nkt = (int)max((int64_t)0, min((int64_t)a.ktiles_per_split, ktiles_total - kt0));

  // raw tile loader: thread e < nrawrows*KT handles (row r, snapshot s)
  auto load_raw = [&](int64_t kt) -> double {
    if (tid >= nrawrows * KT) return 0.0;
    int r = tid / KT, s = tid % KT;
    int64_t i = kt * KT + s;
    if (i >= a.Ns) return 0.0;
    int rr = r % nzm;          // rows: [alpha(nzeta) u(m) | beta(nzeta) u(m)]
    const double* src = rr < b.nzeta ? ((r < nzm ? a.alpha : a.beta) + (int64_t)rr * a.Ns) : (a.u + (int64_t)(rr - b.nzeta) * a.Ns);
    return src[i];
  };
  auto store_raw = [&](int buf, double v) {
    if (tid < nrawrows * KT) sm[L.raw + buf * nrawrows * KT + tid] = v;
  };

  // lift of one KT tile from raw buffer rb into Psi buffer pb
  const int jl = tid & 15, combo = tid >> 4, ls = combo & (KT - 1), lside = combo >> 3;
  auto lift_tile = [&](int rb, int pb, int64_t kt) {
    const double* raw = sm + L.raw + rb * nrawrows * KT;
    const double* vars = raw + lside * nzm * KT;   // this side's variables, stride KT
    const double* uvals = raw + b.nzeta * KT;
    double* psi_side = sm + L.psi + (pb * 2 + lside) * KT * a.Wp;
    const bool valid = (kt * KT + ls) < a.Ns;
    if (b.k_pcs == 0) {
      for (int c = jl; c < b.nfull; c += 16) {
        double val = valid ? kp_eval_col(b, b.cols[c], vars + ls, KT) : 0.0;
        put_psi(b, psi_side, a.Wp, ls, c, val, uvals);
      }
    } else {
      double* full = sm + L.full + lside * b.nfull * KT;
      for (int c = jl; c < b.nfull; c += 16) full[c * KT + ls] = valid ? kp_eval_col(b, b.cols[c], vars + ls, KT) : 0.0;
      __syncthreads();
      for (int c = jl; c < b.N; c += 16) {
        double val;
        if (c < b.nvars)
          val = vars[c * KT + ls];
        else if (c < b.nvars + b.k_pcs) {
          const double* pc = b.pcs + (size_t)(c - b.nvars) * b.nfull;
          val = 0.0;
          for (int i = 0; i < b.nfull; ++i) val += pc[i] * full[i * KT + ls];
        } else
          val = 1.0;
        put_psi(b, psi_side, a.Wp, ls, c, valid ? val : 0.0, uvals);
      }
    }
    if (b.model_type == KP_MODEL_LINEAR) {  // [psi , u]
      for (int i = jl; i < b.m; i += 16) psi_side[ls * a.Wp + b.N + i] = valid ? uvals[i * KT + ls] : 0.0;
    }
  };

  // prologue: raw tile 0 -> LDS, lift it, raw tile 1 -> LDS
  store_raw(0, load_raw(kt0));
  __syncthreads();
  if (nkt > 0) lift_tile(0, 0, kt0);
  store_raw(1, load_raw(kt0 + 1));
  __syncthreads();

  for (int t = 0; t < nkt; ++t) {
    // (1) prefetch raw tile t+2 into a register
    double rawreg = load_raw(kt0 + t + 2);
    // (2) lift tile t+1 into the other Psi buffer
    if (t + 1 < nkt) lift_tile((t + 1) & 1, (t + 1) & 1, kt0 + t + 1);
    // (3) MFMA over tile t
    const double* P = sm + L.psi + (t & 1) * 2 * KT * a.Wp;
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
    // (4) raw tile t+2 -> raw buffer t&1 (last read while lifting tile t)
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

struct GramPlan {
  int nt, Wp, nacc, njobs, nsuper, ntile_out;
  std::vector<uint32_t> desc;
  std::vector<int> tile_out, tile_info;
};

static void make_plan(int W, GramPlan& p) {
  p.nt = (W + 15) / 16;
  int wp = p.nt * 16;
  while (wp % 32 != 16) wp += 16;  // conflict-free ds_read_b64 of rows k, k+1
  p.Wp = wp;
  int ntile = p.nt * (p.nt + 1) / 2 + p.nt * p.nt;
  p.ntile_out = ntile;
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
  p.nacc = best;
  p.nsuper = (ntile + 4 * best - 1) / (4 * best);
  p.njobs = p.nsuper * 4;
  p.desc.assign((size_t)p.njobs * best, 0u);
  p.tile_out.assign((size_t)p.njobs * best, -1);
  p.tile_info.clear();
  int id = 0;
  auto push = [&](int kind, int tr, int tc) {
    uint32_t a_off = (uint32_t)(tr * 16);
    uint32_t b_off = (uint32_t)((kind ? KT * p.Wp : 0) + tc * 16);
    // deal tiles round-robin over jobs so every wave carries the same load
    int job = id % p.njobs, slot = id / p.njobs;
    p.desc[(size_t)job * best + slot] = a_off | (b_off << 16);
    p.tile_out[(size_t)job * best + slot] = id;
    p.tile_info.push_back(kind);
    p.tile_info.push_back(tr);
    p.tile_info.push_back(tc);
    ++id;
  };
  for (int tr = 0; tr < p.nt; ++tr)
    for (int tc = tr; tc < p.nt; ++tc) push(0, tr, tc);
  for (int tr = 0; tr < p.nt; ++tr)
    for (int tc = 0; tc < p.nt; ++tc) push(1, tr, tc);
}

template <int NACC>
static hipError_t launch_gram(const GramArgs& a, int grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)kp_gram_kernel<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kp_gram_kernel<NACC>, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

int kp_gram_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev) {
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W;
  GramPlan plan;
  make_plan(W, plan);
  if ((uint32_t)(2 * KT * plan.Wp) > 65535u) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: dictionary too wide");
  GramLds L = gram_lds(b, plan.Wp);
  size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: dictionary too wide for the LDS-staged tile (W > ~580)");
  if (2 * (b.nzeta + b.m) * KT > 256) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: too many raw columns");
  int64_t ktiles = (s->Ns + KT - 1) / KT;
  int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ktiles, ncu / plan.nsuper > 0 ? ncu / plan.nsuper : 1));
  int kps = (int)((ktiles + nsplit - 1) / nsplit);
  if (kps < 1) kps = 1;
  nsplit = (int)std::max<int64_t>(1, (ktiles + kps - 1) / kps);

  size_t b_desc = plan.desc.size() * 4, b_to = plan.tile_out.size() * 4, b_ti = plan.tile_info.size() * 4;
  size_t b_part = (size_t)nsplit * plan.ntile_out * 256 * 8;
  char* tab = (char*)ctx->workspace(3, b_desc + b_to + b_ti);
  double* part = (double*)ctx->workspace(4, b_part);
  if (!tab || !part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory");
  KP_HIP(ctx, hipMemcpyAsync(tab, plan.desc.data(), b_desc, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(tab + b_desc, plan.tile_out.data(), b_to, hipMemcpyHostToDevice, ctx->stream));
  KP_HIP(ctx, hipMemcpyAsync(tab + b_desc + b_to, plan.tile_info.data(), b_ti, hipMemcpyHostToDevice, ctx->stream));
  // the host vectors die at return; make sure the copies are done (small, once per call)
  KP_HIP(ctx, hipStreamSynchronize(ctx->stream));

  GramArgs a;
  a.b = b;
  a.alpha = s->alpha;
  a.beta = s->beta;
  a.u = s->u;
  a.Ns = s->Ns;
  a.Wp = plan.Wp;
  a.nsuper = plan.nsuper;
  a.ktiles_per_split = kps;
  a.desc = (const uint32_t*)tab;
  a.tile_out = (const int*)(tab + b_desc);
  a.part = part;
  a.ntile_out = plan.ntile_out;
  int grid = plan.nsuper * nsplit;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipError_t e;
  switch (plan.nacc) {
    case 8: e = launch_gram<8>(a, grid, lds, ctx->stream); break;
    case 16: e = launch_gram<16>(a, grid, lds, ctx->stream); break;
    case 24: e = launch_gram<24>(a, grid, lds, ctx->stream); break;
    case 28: e = launch_gram<28>(a, grid, lds, ctx->stream); break;
    default: e = launch_gram<32>(a, grid, lds, ctx->stream); break;
  }
  KP_HIP(ctx, e);
  hipLaunchKernelGGL(kp_gram_reduce_kernel, dim3(plan.ntile_out), dim3(256), 0, ctx->stream, part, nsplit, plan.ntile_out,
                     (const int*)(tab + b_desc + b_to), W, GC_dev, GC_dev + (size_t)W * W);
  KP_HIP(ctx, hipGetLastError());
  KP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  return KP_OK;
}

// Gram matrices of WIDE dictionaries: G = Px'Px, C = Px'Py (Ksysid.m:1114, 1125; inside `\` at :1069) for rows too wide for the
// kernels that keep a lifted tile of the whole row in LDS (kp_gram*.hip stop near W = 580).  The reference accepts such
// dictionaries as they come: `def_fourierLift` on the arm's six states gives 728 functions (Ksysid.m:694-731; linear row 738
// columns, bilinear row 2 940), poly-3 on a delay-embedded state 816 (Ksysid.m:868-907).
//
// At these widths the fused form buys nothing - a 128 x 64 output tile re-reads its operands from L2 once per 16 snapshots
// whether they were lifted on the fly or not, and G alone (69 MB at W = 2 940) is far beyond any on-chip store - so the path
// is: lift a PANEL of snapshots into HBM (kp_lift_kernel; column-major, i.e. contiguous along the snapshots = along the
// contraction index of both products), then two TN products on the matrix pipe (kp_tn_gemm.h): the upper tiles of Px'Px
// and all of Px'Py, split over the snapshots when the tiles alone do not fill the chip, partial sums added in split order,
// panels accumulated in panel order (bitwise reproducible); G's lower triangle is a copy of the upper one (exactly symmetric).
// 288 GB of HBM hold a panel of ~1 GB per side without thought; the panel length only bounds the workspace.
// Bilinear dictionaries go through the Kronecker form (below): weighted products of the N-wide panel of psi.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kp_internal.h"
#include "kp_tn_gemm.h"

static size_t wide_panel_bytes() {      // (read per call: the tests shrink the panel to exercise the accumulation over panels)
  const char* e = getenv("KP_WIDE_PANEL_MB");
  const long mb = e ? atol(e) : 1024;
  return (size_t)std::max(1L, mb) << 20;
}

int kp_gram_wide_launch(kp_ctx* ctx, const kp_basis* basis, const kp_snapshots* s, double* GC_dev) {
  const BasisDev& b = basis->dev;
  if (s->nzeta != b.nzeta || s->m != b.m) return ctx->fail(KP_ERR_ARG, "kp_fit_gram: snapshot/basis dimension mismatch");
  const int W = b.W;
  const int64_t Ns = s->Ns;
  hipStream_t st = ctx->stream;
  // Bilinear rows are Psi = [psi, u_1 psi, ..., u_m psi] (Ksysid.m:510-511, 1594-1604), so with ut = [1; u] block (a, b) of
  // Px'Px is  sum_k ut_a ut_b psi_x psi_x'  - the SAME symmetric N x N matrix at (a, b) and (b, a) - and block (a, b) of Px'Py
  // is  sum_k ut_a ut_b psi_x psi_y' = block (b, a): (m+1)(m+2)/2 weighted products of the N-wide panel instead of the dense
  // products of the N(m+1)-wide one, 62.5 % of their flops at m = 3 (the structure kp_gram3.hip is built on), and a panel a
  // quarter as wide to lift, write and re-read.  The weight rides on the A operand of the TN product (kp_tn_gemm.h).
  // KP_WIDE_DENSE=1 (read per call) keeps the dense form, which the tests compare with.
  const bool kron = b.model_type == KP_MODEL_BILINEAR && b.m > 0 && W == b.N * (b.m + 1) && s->u && getenv("KP_WIDE_DENSE") == nullptr;
  const int Wp = kron ? b.N : W;                     // width of the lifted panel
  const int npair = kron ? (b.m + 1) * (b.m + 2) / 2 : 1;
  // panel length: a multiple of 64 rows, at most the budget, at most what 32-bit tile offsets reach
  int64_t nc = (int64_t)(wide_panel_bytes() / ((size_t)8 * Wp));
  nc = std::max<int64_t>(1024, std::min<int64_t>(nc, (int64_t)1 << 20)) / 64 * 64;
  if (nc > Ns) nc = std::max<int64_t>(64, (Ns + 63) / 64 * 64);
  double* Px = (double*)ctx->workspace(15, (size_t)nc * Wp * 8);
  double* Py = (double*)ctx->workspace(16, (size_t)nc * Wp * 8);
  if (!Px || !Py) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory (lifted panel of a wide dictionary)");
  const int slots = 2 * (ctx->num_cu > 0 ? ctx->num_cu : 256);
  const int nsplit_g = tng_pick_splits(Wp, Wp, (int)std::min<int64_t>(nc, Ns), 1, slots);
  const int nsplit_c = tng_pick_splits(Wp, Wp, (int)std::min<int64_t>(nc, Ns), 0, slots);
  const int nsp = std::max(nsplit_g, nsplit_c);
  double* part = nsp > 1 ? (double*)ctx->workspace(17, (size_t)nsp * Wp * Wp * 8) : nullptr;
  if (nsp > 1 && !part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory (split partials of a wide dictionary)");
  double* G = GC_dev;
  double* C = GC_dev + (size_t)W * W;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, st));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], st));
  if (Ns == 0) KP_HIP(ctx, hipMemsetAsync(GC_dev, 0, (size_t)2 * W * W * 8, st));
  for (int64_t r0 = 0; r0 < Ns; r0 += nc) {
    const int64_t rows = std::min(nc, Ns - r0);
    const int what = kron ? KP_LIFT_ECON : KP_LIFT_ROW;
    int rc = kp_lift_dev_ld(ctx, basis, what, s->alpha + r0, s->u ? s->u + r0 : nullptr, rows, Ns, Px, nc);
    if (!rc) rc = kp_lift_dev_ld(ctx, basis, what, s->beta + r0, s->u ? s->u + r0 : nullptr, rows, Ns, Py, nc);
    if (rc) return rc;
    const double beta = r0 > 0 ? 1.0 : 0.0;
    if (!kron) {
      KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Px, nc, W, W, (int)rows, G, W, 1.0, beta, 1, nsplit_g, part));
      KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Py, nc, W, W, (int)rows, C, W, 1.0, beta, 0, nsplit_c, part));
    } else {
      const int N = b.N;
      for (int ia = 0; ia <= b.m; ++ia)
        for (int ib = ia; ib <= b.m; ++ib) {
          const double* wa = ia ? s->u + (int64_t)(ia - 1) * Ns + r0 : nullptr;     // column ia - 1 of u (leading dimension Ns)
          const double* wb = ib ? s->u + (int64_t)(ib - 1) * Ns + r0 : nullptr;
          const size_t blk = (size_t)ia * N + (size_t)ib * N * W;
          if (kp_abl_int("KP_WIDE_NOWEIGHT")) wa = wb = nullptr;   // TIMING EXPERIMENT ONLY (wrong Grams; -DKP_ABLATIONS builds)
          KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Px, nc, N, N, (int)rows, G + blk, W, 1.0, beta, 1, nsplit_g, part, wa, wb));
          KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Py, nc, N, N, (int)rows, C + blk, W, 1.0, beta, 0, nsplit_c, part, wa, wb));
        }
    }
  }
  if (kron && Ns > 0) {
    // off-diagonal blocks: the lower half of the (symmetric) block of G; block (b, a) of C is block (a, b) as it stands
    const int N = b.N;
    for (int ia = 0; ia <= b.m; ++ia)
      for (int ib = ia + 1; ib <= b.m; ++ib) {
        double* Gab = G + (size_t)ia * N + (size_t)ib * N * W;
        hipLaunchKernelGGL(kp_mirror_upper_kernel, dim3((N + 15) / 16, (N + 15) / 16), dim3(256), 0, st, Gab, N, (int64_t)W);
        KP_HIP(ctx, hipGetLastError());
        KP_HIP(ctx, hipMemcpy2DAsync(C + (size_t)ib * N + (size_t)ia * N * W, (size_t)W * 8, C + (size_t)ia * N + (size_t)ib * N * W, (size_t)W * 8,
                                     (size_t)N * 8, (size_t)N, hipMemcpyDeviceToDevice, st));
      }
  }
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], st));
  if (Ns > 0) {
    hipLaunchKernelGGL(kp_mirror_upper_kernel, dim3((W + 15) / 16, (W + 15) / 16), dim3(256), 0, st, G, W, (int64_t)W);
    KP_HIP(ctx, hipGetLastError());
  }
  KP_HIP(ctx, hipEventRecord(ctx->ev1, st));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], st));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  // executed on the matrix pipe per pair: the output tiles (128 rows x 64 or 96 columns) that meet the upper triangle of a
  // product, and all of the others - of the dense products, or of every weighted one
  auto tiles = [&](int tri, int nsplit) {
    // (the split count kp_tn_gemm ends up with for the first panel: the contraction range of a split is a multiple of 16)
    const int K0 = (int)std::min<int64_t>(nc, Ns);
    const int kper = std::max(TNG_KB, ((K0 + nsplit - 1) / nsplit + TNG_KB - 1) / TNG_KB * TNG_KB);
    const int ns_eff = nsplit > 1 ? (K0 + kper - 1) / kper : 1;
    const TngShape& sh = tng_shapes()[tng_pick_shape(Wp, Wp, ns_eff, tri)];
    const double tm = sh.tm, tn = sh.tn, nrt = std::ceil(Wp / tm), nct = std::ceil(Wp / tn);
    double cnt = 0;
    for (int r = 0; r < (int)nrt; ++r)
      for (int c = 0; c < (int)nct; ++c) cnt += (!tri || r * tm <= c * tn + tn - 1) ? 1 : 0;
    return 2.0 * tm * tn * cnt;
  };
  ctx->timers[10] = npair * (tiles(1, nsplit_g) + tiles(0, nsplit_c));
  return KP_OK;
}

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
  // panel length: a multiple of 64 rows, at most the budget, at most what 32-bit tile offsets reach
  int64_t nc = (int64_t)(wide_panel_bytes() / ((size_t)8 * W));
  nc = std::max<int64_t>(1024, std::min<int64_t>(nc, (int64_t)3 << 20)) / 64 * 64;
  if (nc > Ns) nc = std::max<int64_t>(64, (Ns + 63) / 64 * 64);
  double* Px = (double*)ctx->workspace(15, (size_t)nc * W * 8);
  double* Py = (double*)ctx->workspace(16, (size_t)nc * W * 8);
  if (!Px || !Py) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory (lifted panel of a wide dictionary)");
  const int slots = 2 * (ctx->num_cu > 0 ? ctx->num_cu : 256);
  const int nsplit_g = tng_pick_splits(W, W, (int)std::min<int64_t>(nc, Ns), 1, slots);
  const int nsplit_c = tng_pick_splits(W, W, (int)std::min<int64_t>(nc, Ns), 0, slots);
  const int nsp = std::max(nsplit_g, nsplit_c);
  double* part = nsp > 1 ? (double*)ctx->workspace(17, (size_t)nsp * W * W * 8) : nullptr;
  if (nsp > 1 && !part) return ctx->fail(KP_ERR_HIP, "kp_fit_gram: out of device memory (split partials of a wide dictionary)");
  double* G = GC_dev;
  double* C = GC_dev + (size_t)W * W;
  KP_HIP(ctx, hipEventRecord(ctx->ev0, st));
  KP_HIP(ctx, hipEventRecord(ctx->evp[0], st));
  if (Ns == 0) KP_HIP(ctx, hipMemsetAsync(GC_dev, 0, (size_t)2 * W * W * 8, st));
  for (int64_t r0 = 0; r0 < Ns; r0 += nc) {
    const int64_t rows = std::min(nc, Ns - r0);
    int rc = kp_lift_dev_ld(ctx, basis, KP_LIFT_ROW, s->alpha + r0, s->u ? s->u + r0 : nullptr, rows, Ns, Px, nc);
    if (!rc) rc = kp_lift_dev_ld(ctx, basis, KP_LIFT_ROW, s->beta + r0, s->u ? s->u + r0 : nullptr, rows, Ns, Py, nc);
    if (rc) return rc;
    const double beta = r0 > 0 ? 1.0 : 0.0;
    KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Px, nc, W, W, (int)rows, G, W, 1.0, beta, 1, nsplit_g, part));
    KP_HIP(ctx, kp_tn_gemm(st, Px, nc, Py, nc, W, W, (int)rows, C, W, 1.0, beta, 0, nsplit_c, part));
  }
  KP_HIP(ctx, hipEventRecord(ctx->evp[1], st));
  if (Ns > 0) {
    hipLaunchKernelGGL(kp_mirror_upper_kernel, dim3((W + 15) / 16, (W + 15) / 16), dim3(256), 0, st, G, W, (int64_t)W);
    KP_HIP(ctx, hipGetLastError());
  }
  KP_HIP(ctx, hipEventRecord(ctx->ev1, st));
  KP_HIP(ctx, hipEventRecord(ctx->evp[2], st));
  ctx->gram_flops_per_pair = (double)W * (W + 1) + 2.0 * W * W;
  // executed on the matrix pipe per pair: the output tiles (128 rows x 64 or 96 columns) that meet the upper triangle of G, and all of C
  auto tiles = [&](int tri, int nsplit) {
    const double tm = 128.0, tn = (double)tng_tile_cols(W, W, nsplit, tri), nrt = std::ceil(W / tm), nct = std::ceil(W / tn);
    double cnt = 0;
    for (int r = 0; r < (int)nrt; ++r)
      for (int c = 0; c < (int)nct; ++c) cnt += (!tri || r * tm <= c * tn + tn - 1) ? 1 : 0;
    return 2.0 * tm * tn * cnt;
  };
  ctx->timers[10] = tiles(1, nsplit_g) + tiles(0, nsplit_c);
  return KP_OK;
}

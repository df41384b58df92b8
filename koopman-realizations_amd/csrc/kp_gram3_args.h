// Arguments of the Kronecker Gram kernels (kp_gram3.hip: 4-wave workgroups; kp_gram6.hip: 8-wave workgroups with the weighted
// A operands in LDS), shared so that kp_gram3_launch can hand one plan / partial layout to either.
#pragma once
#include "kp_internal.h"

struct Gram3Args {
  BasisDev b;
  const double* alpha;   // allocated with >= 64 doubles of zero padding (kp_snapshots_upload): prefetch never leaves the buffer
  const double* beta;
  const double* u;
  int64_t Ns;
  int G4;               // 4-column groups per side
  int nsuper;           // workgroups per snapshot split
  int ktiles_per_split;
  int D;
  const uint32_t* recipes;   // [nfull]
  const uint32_t* desc;      // [njobs][1 + NQ]: a0 | a1 << 8 | qs << 16 (quads < qs use A group a0, the rest a1),
                             // then per quad 4 packed B group ids (8 bit each)
  double* part;              // [nsplit][njobs][NQ][NWT][64]
  int njobs;
  const double* pcs;         // nfull x k_pcs (column-major) or nullptr: econ lift [zeta | pcs' psi_full | 1] (Ksysid.m:1594-1618)
  int nfull4;                // nfull rounded up to a multiple of 4
  // EXT: table entries per variable = Dp powers, then df (cos, sin) pairs (D = Dp + 2 df); ng gaussian centres (nzeta each)
  int Dp, df, ng;
  const double* centres;
  // PRE (dim_red dictionaries, kp_gram3_prelift_kernel): per tile of 8 snapshots the entries [psi_x (4 G4) | psi_y (4 G4) | 9 weights + 3
  // zeros] of the econ lift as [entry][snapshot], zeros past Ns; pre_rl = entries per snapshot
  const double* pre = nullptr;
  int pre_rl = 0;
};

// one job per wave: kp_gram6_kernel<NQ, G4C> (kp_gram6.hip); hipErrorInvalidValue when no instantiation serves (nq, G4)
bool kp_gram6_serves(int nq, int G4);
hipError_t kp_gram6_launch_kernel(const Gram3Args& a, int nq, int grid, hipStream_t st);

// kp_gram3_prelift.hip: the projection matrix as [full column][32 components] (zero padded), and the econ lift of every snapshot in the
// tile layout of Gram3Args::pre
hipError_t kp_gram3_pcs_transpose_launch(const double* pcs, int nfull, int k, double* pcsT, hipStream_t st);
hipError_t kp_gram3_prelift_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int D, int nfull, int k_pcs,
                                   int N, int G4, const uint32_t* recipes, const double* pcsT, double* out, int rl, hipStream_t st);
// ... of a dictionary with fourier / gaussian blocks (EXT recipes, no projection); at most 64 KB of LDS: nzeta (Dp + 2 df) + ng + 1 <= 64 entries, nzeta <= 16
hipError_t kp_gram3_prelift_ext_launch(int BM, const double* alpha, const double* beta, const double* u, int64_t Ns, int64_t Ns_pad, int nzeta, int Dp, int df, int ng,
                                       int nfull, int G4, const uint32_t* recipes, const double* centres, double* out, int rl, hipStream_t st);

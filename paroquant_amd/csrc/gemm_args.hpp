// Kernel argument block shared by the prefill GEMM variants (gemm.hip, gemm3.hip).
#pragma once
#include "common.hpp"

namespace paro {

struct GemmArgs {
  const u32x4* wq;
  const unsigned* sz;
  const unsigned short* bias;
  const unsigned short* xrot;  // [nparts][rows][K]
  unsigned short* y;
  int K, N, G, rows;
  int tstride, gstride;        // 1-KiB chunk index of tile (t, g) = t * tstride + g * gstride
  int ksplit, gps;             // K-split (grid.z) of the v2 kernel: groups per split; 1 = none
  float* partial;              // [ksplit][rows][N] fp32 partial sums when ksplit > 1
  PartTable pt;                // column blocks of BN_TILES tiles
  // grouped (mixture-of-experts) launches of variant 4: row block blockIdx.y multiplies by the weights of expert
  // block_expert[blockIdx.y] (device memory; -1 = an unused block of the padded row space); null = one weight set
  const int* block_expert = nullptr;
  long long wq_estride = 0, sz_estride = 0;   // u32x4 / u32 elements between consecutive experts' packed buffers
  int n_experts = 0;                          // block_expert entries outside [0, n_experts) are skipped on the device
  int cb0 = 0;                                // variant 4: first column block of this launch (per-partition launches of a merged projection)
  // variant 44 (the north star's fused prefill form, an EXPERIMENT build -- gemm3.hip DIAG 4): `xrot` is the UN-rotated x [rows][K] and every
  // workgroup rotates the slab it stages with the dense per-group matrices rmat[p][g][n][k] (paro_linear_t.rmat) on the matrix cores
  const unsigned short* rmat = nullptr;
};

}  // namespace paro

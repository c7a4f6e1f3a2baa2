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
};

}  // namespace paro

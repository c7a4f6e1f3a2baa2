// One translation unit per (activation type, pre-rotated?, tiles per wave) of the fused GEMV, so that the
// variants build in parallel (see the Makefile): compiled 16 times with
//   -DPARO_INST_AT=f16|bf16 -DPARO_INST_PRE=0|1 -DPARO_INST_TPW=1|2|4|8 -DPARO_INST_NAME=launch_gemv_<...>
#include "gemv_impl.hpp"

namespace paro {
int PARO_INST_NAME(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  return launch_rows<PARO_INST_AT, PARO_INST_TPW, PARO_INST_PRE != 0>(a, waves, grid, st);
}
}  // namespace paro

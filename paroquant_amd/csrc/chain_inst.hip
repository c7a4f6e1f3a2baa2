// One translation unit per (activation type, row class) of the decode-chain GEMV (chain_impl.hpp), built in parallel:
//   -DPARO_CHAIN_AT=f16|bf16 -DPARO_CHAIN_MB=1|4|8|16 -DPARO_CHAIN_NAME=launch_chain_<type>_m<rows>
#include "chain_impl.hpp"

namespace paro {
int PARO_CHAIN_NAME(const ChainArgs& a, int waves, bool pair, dim3 grid, hipStream_t st) {
  return chain_launch_mb<PARO_CHAIN_AT, PARO_CHAIN_MB>(a, waves, pair, grid, st);
}
}  // namespace paro

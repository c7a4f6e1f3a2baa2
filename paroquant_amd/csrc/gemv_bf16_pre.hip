// Instantiations of the fused GEMV for bf16 activations, pre-rotated input (see gemv_impl.hpp).
#include "gemv_impl.hpp"

namespace paro {
int launch_gemv_bf16_pre(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st) {
  return launch_gemv_variant<bf16, true>(a, tpw, waves, grid, st);
}
}  // namespace paro

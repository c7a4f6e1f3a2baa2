// Instantiations of the fused GEMV for f16 activations (see gemv_impl.hpp).
#include "gemv_impl.hpp"

namespace paro {
int launch_gemv_f16(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st) {
  return launch_gemv_variant<f16, false>(a, tpw, waves, grid, st);
}
}  // namespace paro

// ABI plumbing: version, thread-local error string, and the M-based dispatcher.
#include "common.hpp"

namespace paro {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace paro

extern "C" int paro_abi_version(void) { return PARO_ABI_VERSION; }

extern "C" const char* paro_last_error(void) { return paro::error_buffer(); }

// RotateQuantizedLinear.forward / ParoQuantLinearMethod.apply entry point
// (transformers/modules.py:57-71, vllm/plugin.py:281-311): small batches stream the weights
// once through the fused GEMV kernel, everything else goes rotate pre-pass + MFMA GEMM.
extern "C" int paro_w4a16_linear(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (rows <= 16) return paro_w4a16_gemv(L, x, y, rows, workspace, workspace_bytes, 0, 0, 0, -1, stream);
  return paro_w4a16_gemm(L, x, y, rows, workspace, workspace_bytes, stream);
}

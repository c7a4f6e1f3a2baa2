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
  // 17..32 rows on a wide output (>= 1024 column tiles): two passes of the 16-row GEMV beat the MFMA kernel,
  // whose per-group cost does not shrink with M (measured, gate_up 4096 -> 28672: M = 32 50 vs 80 us; for
  // the narrow shapes the K-split GEMM wins: o_proj 27 vs 34 us)
  if (rows <= 32 && L && L->N / 16 >= 1024 && L->act_dtype == PARO_DTYPE_F16) {
    const int64_t esz = 2;
    int rc = paro_w4a16_gemv(L, x, y, 16, workspace, workspace_bytes, 0, 0, 0, -1, stream);
    if (rc != PARO_OK) return rc;
    return paro_w4a16_gemv(L, (const char*)x + 16 * L->K * esz, (char*)y + 16 * L->N * esz, rows - 16, workspace,
                           workspace_bytes, 0, 0, 0, -1, stream);
  }
  return paro_w4a16_gemm(L, x, y, rows, workspace, workspace_bytes, stream);
}

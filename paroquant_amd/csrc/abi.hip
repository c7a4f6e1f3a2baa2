// ABI plumbing: version, thread-local error string, and the M-based dispatcher.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace paro {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
int device_cu_count() {
  static int cus = 0;
  if (cus <= 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      v = 256;   // MI355X
    cus = v;
  }
  return cus;
}
}  // namespace paro

extern "C" int paro_workspace_status(const void* workspace, void* stream) {
  using namespace paro;
  if (!workspace) return fail(PARO_ERR_INVALID, "null workspace");
  unsigned word = 0;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e1 = hipMemcpyAsync(&word, (const char*)workspace + PARO_WS_STATUS_OFFSET, 4, hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  if (e1 != hipSuccess || e2 != hipSuccess)
    return fail(PARO_ERR_LAUNCH, "paro_workspace_status: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  if (word == PARO_WS_STATUS_GIVEUP)
    return fail(PARO_ERR_LAUNCH, "a K-split reducer gave up waiting for a partial sum (outputs of that launch are NaN); "
                                 "zero-fill the workspace before reusing it");
  if (word != 0) return fail(PARO_ERR_LAUNCH, "workspace status word is 0x%x: the workspace was not zero-filled", word);
  return PARO_OK;
}

extern "C" int paro_abi_version(void) { return PARO_ABI_VERSION; }

extern "C" const char* paro_last_error(void) { return paro::error_buffer(); }

// RotateQuantizedLinear.forward / ParoQuantLinearMethod.apply entry point
// (transformers/modules.py:57-71, vllm/plugin.py:281-311): small batches stream the weights
// once through the fused GEMV kernel, everything else goes rotate pre-pass + MFMA GEMM.
// (9..16 rows: the alternative second launch behind a rotation launch -- the chain family's kernel -- was measured SLOWER than the GEMV on
// rotated x, profiles/r05_rows_boundary.jsonl, and its host branch was removed in round 6: profiles/NOTES.md 5.4.)
extern "C" int paro_w4a16_linear(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (rows <= 16) return paro_w4a16_gemv(L, x, y, rows, workspace, workspace_bytes, 0, 0, 0, -1, stream);
  // 17..32 rows: the GEMV kernel on pre-rotated activations with 2 MFMA row tiles per weight fragment (measured,
  // Llama-3-8B shapes at 24 rows, us GEMV / MFMA GEMM: qkv 25 / 30, o 24 / 26, gate_up 31 / 68, down 24 / 45); from 33
  // rows on the GEMM's 64- / 128-row blocks with a K-split are ahead on every shape (48 rows: gate_up 33 vs 70, down 29
  // vs 36, qkv 27 vs 27).  PARO_SKINNY=0 routes everything above 16 rows to the GEMM (A/B runs).
  static const int skinny = getenv("PARO_SKINNY") ? atoi(getenv("PARO_SKINNY")) : 1;
  // Round 6: behind the schedule pre-pass (rotate.hip; x in MFMA-fragment order) the GEMV with 4 row tiles is ahead of the GEMM up to
  // 64 rows on every output below 1024 tiles (profiles/r06_skinny_routes.jsonl, us GEMV / GEMM at 48 rows: Qwen3-4B qkv 12.8 / 18.3,
  // o 13.2 / 15.9, down 18.8 / 22.3; Llama-3-8B qkv 17.0 / 21.5, o 14.6 / 18.2, down 26.6 / 27.1); wide merged projections keep the GEMM
  // from 33 rows on (gate_up 26.9 / 24.6, 48.5 / 33.6: 2-tile blocks re-read 64 rows of x per 32 columns).  PARO_SKINNY_MAX overrides.
  static const int skinny_env = getenv("PARO_SKINNY_MAX") ? atoi(getenv("PARO_SKINNY_MAX")) : -1;
  static const int sched_env = getenv("PARO_PREROT_SCHED") ? atoi(getenv("PARO_PREROT_SCHED")) : 1;
  const int skinny_max = skinny_env >= 0 ? skinny_env : ((L && L->rot && L->krot <= 8 && sched_env != 0 && L->N / 16 < 1024) ? 64 : 32);
  if (skinny && rows <= skinny_max && rows <= 64 && L)   // fp16 and bf16 alike
    return paro_w4a16_gemv(L, x, y, rows, workspace, workspace_bytes, 0, 0, 0, 1, stream);
  return paro_w4a16_gemm(L, x, y, rows, workspace, workspace_bytes, PARO_GEMM_AUTO, stream);
}

/*
 * paro_abi.h -- C ABI of libparo_mi355x.so: the MI355X (gfx950 / CDNA4) native
 * implementation of ParoQuant's inference hot path
 *
 *     y = rotate(x * channel_scales; pairs, theta) @ dequant(qweight, qzeros, scales) (+ bias)
 *
 * Plain `extern "C"`, raw device pointers and sizes only (no torch / C++ types).
 * Every entry point cites the reference interface (z-lab/paroquant v0.1.16,
 * paths relative to the reference root) that a maintainer would re-bind to it;
 * the ctypes stubs are shown in INTEGRATION.md and live in
 * paroquant_amd/_native.py.
 *
 * Contract
 *  - All device buffers are owned by the caller (torch's caching allocator);
 *    the library never allocates, frees or retains device memory.  Outputs and
 *    workspaces are caller-allocated.
 *  - Inputs are contiguous row-major (the reference assumes this silently,
 *    rotation.cu:48-50).
 *  - Kernels are enqueued asynchronously on the caller's `hipStream_t`
 *    (`stream`, passed as void*; the reference uses the current stream,
 *    rotation.cu:82).  No host synchronisation, no host reads of device memory,
 *    no global mutable state except a thread-local error string: every call is
 *    re-entrant and HIP-graph capturable.
 *  - Functions return PARO_OK (0) or a negative error code and never throw;
 *    `paro_last_error()` returns the thread-local message (the Python shim
 *    raises RuntimeError with it, mirroring TORCH_CHECK at rotation.cu:66,92,108,114,123).
 */
#ifndef PARO_ABI_H
#define PARO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARO_ABI_VERSION 1

/* element types of activations / rotation parameters */
#define PARO_DTYPE_F32 0
#define PARO_DTYPE_F16 1
#define PARO_DTYPE_BF16 2

#define PARO_OK 0
#define PARO_ERR_INVALID (-1)     /* bad shape / argument (reference: TORCH_CHECK / ValueError) */
#define PARO_ERR_UNSUPPORTED (-2) /* valid in principle, not compiled (e.g. group_size not in {64,128}) */
#define PARO_ERR_LAUNCH (-3)      /* HIP launch error */

#define PARO_MAX_PARTS 8          /* merged projections per linear (qkv = 3, gate_up = 2) */
#define PARO_GROUP 128            /* quantisation group == rotation group at inference */
#define PARO_TILE_N 16            /* columns per packed tile */

int paro_abi_version(void);
const char* paro_last_error(void);

/* ---------------------------------------------------------------------------
 * paro_rotate -- replaces the CUDA implementation behind
 *   torch.ops.rotation.rotate(Tensor x, Tensor idx_ij, Tensor theta,
 *                             Tensor? scales=None, int group_size=128) -> Tensor
 * (paroquant/kernels/cuda/rotation.cu:111-135; kernel :10-43; accessors
 * rotation.cuh:16-75,91-173).
 *
 *   x, out   [rows, hidden] of x_dtype (out may alias x)
 *   idx_ij   int16 [krot, hidden]   group-local pair indices, (i,j) = idx[r, g*GS+2t], idx[r, g*GS+2t+1]
 *   theta    [krot, hidden/2] of param_dtype
 *   scales   [hidden] of param_dtype, or NULL
 *   group_size in {64, 128}; 1 <= krot <= 16; hidden % group_size == 0 (rotation.cu:66)
 *
 * Numerics: x*scales and all krot stages are kept in fp32 and rounded ONCE to
 * x_dtype (the reference re-rounds to half after every stage, rotation.cuh:152-153);
 * theta is read in its own dtype (no bf16 down-cast, cf. rotation.cu:75).
 */
int paro_rotate(const void* x, void* out, const int16_t* idx_ij, const void* theta, const void* scales,
                int64_t rows, int64_t hidden, int krot, int group_size, int x_dtype, int param_dtype,
                void* stream);

/* ---------------------------------------------------------------------------
 * One-time weight repack: AWQ checkpoint layout -> CDNA4 tile layout.
 * Replaces the per-partition AWQ->Marlin conversion of
 * ParoQuantLinearMethod.process_weights_after_loading / _convert_partition
 * (paroquant/inference/backends/vllm/plugin.py:208-279).
 *
 *   qweight int32 [K, N/8]      nibble p of word c = column 8c + (0,2,4,6,1,3,5,7)[p]  (cli/convert.py:19,149-155)
 *   qzeros  int32 [K/128, N/8]  same packing
 *   out_wq  uint32 [N/16][K/128][64][4]  tile (t,g) = 1 KiB; lane l = (kb = l>>4, n = l&15), word i holds
 *           k = 128g + 32i + 8kb + e (e = 0..7) of column 16t+n; element e sits in nibble (e>>1) + 4*(e&1)
 *           -- the MFMA 16x16x32 B-fragment order, two k-adjacent nibbles 16 bits apart.
 *   out_zq  uint32 [K/128][N/8]  natural nibble order (nibble j of word c = column 8c + j)
 *   K % 128 == 0, N % 16 == 0.
 */
int64_t paro_packed_qweight_bytes(int64_t K, int64_t N);
int64_t paro_packed_qzeros_bytes(int64_t K, int64_t N);
int paro_repack_awq(const int32_t* qweight, const int32_t* qzeros, int64_t K, int64_t N, void* out_wq,
                    void* out_zq, void* stream);

/* ---------------------------------------------------------------------------
 * Fused rotate + INT4 dequant + matmul.  Replaces, in one call,
 *   rotate -> AWQ/Marlin GEMM  of RotateQuantizedLinear.forward
 *   (paroquant/inference/backends/transformers/modules.py:57-71) and the
 *   per-partition loop + torch.cat + bias of ParoQuantLinearMethod.apply
 *   (paroquant/inference/backends/vllm/plugin.py:281-311).
 */
typedef struct paro_linear {
  int64_t K;                          /* in_features (per TP partition), multiple of 128 */
  int64_t N;                          /* total out_features = sum(part_cols) */
  int32_t n_parts;                    /* merged projections with distinct rotations (1..PARO_MAX_PARTS) */
  int32_t krot;                       /* rotation stages (1..16), normally 8 */
  int32_t part_cols[PARO_MAX_PARTS];  /* columns per partition, each a multiple of 16 */
  int32_t act_dtype;                  /* PARO_DTYPE_F16 | PARO_DTYPE_BF16: dtype of x, y, bias */
  int32_t reserved;
  const void* wq;                     /* packed weights (paro_repack_awq) */
  const void* zq;                     /* packed zeros   (paro_repack_awq) */
  const void* scales;                 /* fp16 [K/128, N]            (checkpoint layout, unchanged) */
  const int16_t* pairs;               /* int16 [n_parts, krot, K]   (checkpoint layout) */
  const void* theta;                  /* fp16 [n_parts, krot, K/2]  (checkpoint layout) */
  const void* channel_scales;         /* fp16 [n_parts, K]          (checkpoint layout) */
  const void* bias;                   /* act_dtype [N] or NULL */
} paro_linear_t;

/* Bytes of caller-provided scratch the fused ops may need for `rows` rows
 * (split-K slabs + arrival counters for the GEMV path, rotated activations for
 * the GEMM path).  The first PARO_WS_COUNTER_BYTES of the workspace hold
 * arrival counters: the caller zero-fills the workspace ONCE after allocating
 * it (the kernels leave the counters at zero on exit). */
#define PARO_WS_COUNTER_BYTES 16384
int64_t paro_linear_workspace_bytes(const paro_linear_t* L, int64_t rows);

/* Decode / small-batch path (rows <= 16): one launch; x is rotated per
 * 128-channel group inside the workgroup that streams that group's INT4 tiles.
 * tiles_per_wave in {0 (auto), 1, 2, 4, 8}; ksplit >= 0 (0 = auto). */
int paro_w4a16_gemv(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                    int64_t workspace_bytes, int tiles_per_wave, int ksplit, void* stream);

/* Prefill path (any rows): rotate pre-pass into the workspace, then an
 * LDS-staged MFMA GEMM with in-register (q - z) * s dequant. */
int paro_w4a16_gemm(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* Dispatcher used by the Python operator: gemv for rows <= 16, gemm otherwise. */
int paro_w4a16_linear(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* Dequantise packed weights back to a dense [K, N] matrix of act_dtype
 * (debug / verification aid; W[k,n] = (q - z) * s rounded once). */
int paro_dequant_packed(const paro_linear_t* L, void* out_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARO_ABI_H */

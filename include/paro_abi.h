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

#define PARO_ABI_VERSION 19

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
 * One-time repack: checkpoint layout -> CDNA4 kernel layout.  Replaces the
 * per-partition AWQ->Marlin conversion of ParoQuantLinearMethod.
 * process_weights_after_loading / _convert_partition
 * (paroquant/inference/backends/vllm/plugin.py:208-279).
 *
 * Inputs (checkpoint format, cli/convert.py:149-155,194-203,264-277):
 *   qweight int32 [K, N/8]      nibble p of word c = column 8c + (0,2,4,6,1,3,5,7)[p]
 *   qzeros  int32 [K/gs, N/8]   same packing; gs = group_size of the QUANTISATION, 128 or 64 (the rotation always
 *                               works on 128-channel groups at inference: RotateQuantizedLinear.forward calls
 *                               rotate() without a group size, transformers/modules.py:59, and forwards group_size
 *                               only to the AWQ matmul, :60-69)
 *   scales  fp16  [K/gs, N]
 *   part_cols[n_parts]          columns of each merged partition (multiples of 16, sum = N)
 * Outputs:
 *   out_wq  uint32 [N/16][K/128][64][4] (wq_order 0) or [K/128][N/16][64][4] (wq_order 1); tile (t,g) = 1 KiB; lane l = (kb = l>>4, n = l&15), word i holds
 *           k = 128g + 32i + 8kb + e (e = 0..7) of column 16t+n; element e sits in nibble (e>>1) + 4*(e&1)
 *           -- the v_mfma_f32_16x16x32 B-fragment order, the two k-adjacent nibbles 16 bits apart.
 *   out_sz  uint32 [K/gs][Tsz/4][16][4]  one word per (quantisation group, column): lo16 = scale (fp16 bits),
 *           hi16 = fp16(zero_point).  Column tiles are indexed in a padded tile space: partition p
 *           starts at the sum of its predecessors' tile counts rounded up to 8 (Tsz = that sum over all
 *           partitions); word ((g*Tsz/4 + ts/4)*16 + n)*4 + ts%4 belongs to padded tile ts, column n.
 *           Padding words are zero.
 */
int64_t paro_packed_qweight_bytes(int64_t K, int64_t N);
int64_t paro_packed_sz_bytes(int64_t K, int group_size, int n_parts, const int32_t* part_cols);
int paro_repack_awq(const int32_t* qweight, const int32_t* qzeros, const void* scales, int64_t K, int64_t N, int group_size,
                    int n_parts, const int32_t* part_cols, int wq_order, void* out_wq, void* out_sz, void* stream);

/* Rotation parameters -> the register EXCHANGE SCHEDULE of every (partition, group), 3072 bytes each:
 *   out_rot uint32 [n_parts][K/128][3][64 lanes][4]
 * The fused GEMV keeps a group's rotation state in registers: lane l holds both members (A, B) of one
 * pair of the current stage; a stage is  keep' = P A + Q B,  give' = P B - Q A,  after which the lane
 * keeps keep' and fetches the give' of ONE other lane (ds_bpermute).  Which member a lane keeps, the
 * (i, j) orientation and running signs are folded here into (P, Q) = (cos a, sin a),
 * a in {+-theta + k pi/2}  (xi' = c xi + s xj, xj' = c xj - s xi, rotation.cuh:53-56;
 * (i, j) = pairs[p, r, 128g + 2e], pairs[p, r, 128g + 2e + 1], theta = theta[p, r, 64g + e],
 * rotation.cuh:126-127), stored as signed 16-bit fixed point in units of 2^-14 (absolute error
 * <= 3.1e-5; +-1 and 0 exact):
 *   chunk 0 / 1 : stage words t = 0..3 / 4..7 :  Q << 16 | P
 *   chunk 2     : {4 * src_lane of stages 0..3, one byte each; the same for stages 4..7;
 *                  final Q << 16 | P;  2 ch_a | 2 ch_b << 8 | (sigma < 0) << 31}
 *                 final stage: out[ch_a] = P A + Q B, out[ch_b] = sigma (P B - Q A)
 * Stage t = 0 is the identity on the natural layout (channels 2l, 2l+1), stage t >= 1 is checkpoint
 * stage t - 1, the last checkpoint stage is the final one.
 * Requires krot <= 8 (larger krot uses the unfused rotate + GEMV route).
 * `status` is a caller-owned int32 in device memory (the library never allocates): it is zeroed on the
 * stream and then set non-zero by the kernel when a stage is not a perfect matching of its 128 channels
 * (the condition optim/rotation.py:36-37 raises on at conversion).  The call itself neither allocates
 * nor synchronises; the caller reads `status` after the stream has drained (the Python shim raises
 * RuntimeError "illegal pair" from it). */
int64_t paro_packed_rot_bytes(int64_t K, int n_parts);
int paro_pack_rotation(const int16_t* pairs, const void* theta, int64_t K, int n_parts, int krot, void* out_rot,
                       int32_t* status, void* stream);

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
  int32_t wq_order;                   /* tile order of wq: 0 = [tile][group], 1 = [group][tile] */
  const void* wq;                     /* packed INT4 tiles        (paro_repack_awq) */
  const void* sz;                     /* packed scale/zero words  (paro_repack_awq) */
  const void* rot;                    /* packed rotation words    (paro_pack_rotation); NULL iff krot > 8 */
  const int16_t* pairs;               /* int16 [n_parts, krot, K]   (checkpoint layout) */
  const void* theta;                  /* fp16 [n_parts, krot, K/2]  (checkpoint layout) */
  const void* channel_scales;         /* fp16 [n_parts, K]          (checkpoint layout) */
  const void* bias;                   /* act_dtype [N] or NULL */
  const void* rmat;                   /* optional: dense per-group rotation matrices, act_dtype
                                         [n_parts][K/128][128 n][128 k] = (diag(cs) G_1..G_krot)^T, used by
                                         the prefill pre-pass on the matrix cores; NULL = stage kernel */
  int32_t group_size;                 /* quantisation group: 128 (0 is read as 128) or 64 channels per (scale, zero) */
  int32_t launch_hint;                /* v16: 0 = the built-in launch-shape rules; else the shape MEASURED for this layer at load time
                                         (PARO_LAUNCH_HINT(tiles_per_wave, ksplit, waves)): used by ONE-ROW launches whose knobs are
                                         all auto -- paro_w4a16_linear, paro_w4a16_gemv(.., 0, 0, 0, ..), the fused entry -- and never
                                         for the deferred K-split route (that one fixes its own split).  An illegal hint is clamped
                                         like explicit knobs are.  The reference has no counterpart: its INT4 GEMM is a third-party
                                         kernel (AutoAWQ / Marlin) with its own shape selection behind plugin.py:251-311. */
} paro_linear_t;
#define PARO_LAUNCH_HINT(tiles_per_wave, ksplit, waves) (((tiles_per_wave) & 0xff) | (((ksplit) & 0xff) << 8) | (((waves) & 0xff) << 16))

/* Bytes of caller-provided scratch the fused ops may need for `rows` rows
 * (split-K granules for the GEMV paths, rotated activations for the GEMM path).  The first PARO_WS_COUNTER_BYTES of the
 * workspace hold one K-split EPOCH word per column block: the caller zero-fills the workspace ONCE after allocating it;
 * every launch that splits a block tags its {tag, partial} granules with (block, epoch + 1) and the block's reducer
 * advances the word, so nothing is re-armed between launches and whatever else was written to the granule area (the
 * prefill path's rotated activations, a late producer of a launch that gave up) can never be read as a partial sum.
 * A workspace must not be shared by launches that may run concurrently on different streams. */
#define PARO_WS_COUNTER_BYTES 16384
/* Last word of the counter area: sticky status of the in-launch K-split.  0 = healthy.  A reducer that
 * gives up waiting for a partial (bounded spin; cannot happen while the whole grid is resident, which
 * paro_w4a16_gemv checks before every K-split launch) stores PARO_WS_STATUS_GIVEUP here and writes NaN
 * instead of a partial sum -- a failed hand-off is never a silently wrong number.  After a give-up the
 * workspace must be zero-filled again before reuse. */
#define PARO_WS_STATUS_OFFSET (PARO_WS_COUNTER_BYTES - 4)
#define PARO_WS_STATUS_GIVEUP 0xDEADu
int64_t paro_linear_workspace_bytes(const paro_linear_t* L, int64_t rows);
/* Diagnostic (the ONLY entry point that synchronises): copies the status word back after draining
 * `stream`.  Returns PARO_OK, or PARO_ERR_LAUNCH with a message when a K-split gave up.  Call it at
 * graph-capture / teardown time, never on the hot path. */
int paro_workspace_status(const void* workspace, void* stream);

/* Decode / small-batch path (rows <= 64; above 16 rows always with the rotate pre-pass): one launch; x is rotated per
 * 128-channel group inside the workgroup that streams that group's INT4 tiles.
 * Launch-shape knobs (0 = auto): tiles_per_wave in {1, 2, 4, 8};
 * ksplit >= 1; waves per workgroup in {4,8,16} (16: <= 4 rows and <= 4 tiles).  mode: 0 = rotation replicated in every workgroup, 1 = rotate
 * pre-pass kernel into the workspace then the same GEMV on rotated x (krot <= 8: the schedule pre-pass -- the in-kernel rotation's
 * arithmetic, so the same bits as mode 0 on the same launch shape -- handing x over in MFMA-fragment order; krot > 8: the stage kernel,
 * plain rows), 3 (v17) = rotation SHARED inside the launch -- producer workgroups in
 * front of the grid rotate every (partition, group, pair of rows) once and hand it to the column blocks as {two channels, launch tag}
 * granules in the workspace; plain calls of 1..16 rows, bit-identical to mode 0, falls back to mode 0 when the grid cannot be resident at
 * once -- -1 = auto (mode 0 up to 4 rows, mode 3 from 5 rows where it fits the chip, else mode 0 up to 8 rows / for small projections and
 * the pre-pass above; PARO_SHARED_ROT_MIN_ROWS moves the threshold),
 * 2 (v10) = x IS ALREADY ROTATED by the caller: [n_parts][rows][K] in the activation type, partition p rotated with
 * partition p's pairs / theta / channel_scales (rotation::rotate, or the epilogue of whatever kernel produced x) -- the
 * same pre-rotated kernels as mode 1 without the pre-pass launch. */
/* The launch shape paro_w4a16_gemv resolves the knobs to (in: 0 / -1 = auto, out: final values) -- host-only,
 * touches no device memory: for tooling, logs and tests of the heuristics. */
int paro_gemv_launch_shape(const paro_linear_t* L, int64_t rows, int* tiles_per_wave, int* ksplit, int* waves, int* mode);
int paro_w4a16_gemv(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                    int64_t workspace_bytes, int tiles_per_wave, int ksplit, int waves, int mode, void* stream);

/* Decode-layer fusions either side of the linear (the reference runs RMSNorm, SiLU*mul and the
 * residual add as separate framework kernels around `rotate -> GEMM`):
 *   prologue PARO_PROLOGUE_RMSNORM   y = GEMV(x) * rsqrt(mean_k(x_k^2) + eps).  The norm WEIGHT is not an argument:
 *            fold it into `channel_scales` at load time (cs'[p][k] = cs[p][k] * w[k]); the remaining scalar
 *            commutes with the rotation and the matmul, and sum(x^2) is gathered while the kernel seeds the rotation.
 *   prologue PARO_PROLOGUE_SILU_MUL  x_k = silu(gate_k) * up_k, gate = x[row][0..K), up = x[row][K..2K) (the merged
 *            gate_up projection's output), evaluated in fp32 before the rotation.
 *   residual                          y[row][col] += residual[row][col]  (act_dtype, row stride N), or NULL.
 *   all-reduce (v9)                   ar_peers != NULL: the linear is a ROW-PARALLEL shard and y becomes
 *            the sum of the world's partial outputs (+ bias + residual, added once, after the sum): the output threads
 *            exchange {fp32 partial, epoch} granules through the ranks' paro_allreduce buffers (sized with
 *            ar_max_elems >= N) and sum them in rank order -- one rounding, bit-identical on every rank, no separate
 *            all-reduce launch.  One row, not with the RMSNorm prologue (a norm over a K shard is not the layer's norm)
 *            nor with expert slots.  Every rank must issue the same sequence of such launches on its buffer.
 *   deferred K-split reduction (v12)  The narrow, deep linears of a decoder layer (o_proj, down_proj) are K-split, and the
 *            in-launch hand-off of the partial sums (write-through granules, a polling reducer) is 1.0 .. 1.4 us of a 5 .. 8 us
 *            launch (profiles/r03_ab_nopoll.txt, r03_parts_micro.jsonl).  With parts_out the splits just leave their fp32 partial sums
 *                parts_out[col][slot]   fp32 [N][PARO_MAX_PARTIALS]: slot 0 = the LAST split, slot q = split q - 1, the rest zero
 *            (y is not written and may be NULL; no bias, no residual) and exit; the NEXT launch (the RMSNorm-prologue linear of
 *            the following block: gate_up after o_proj, the next layer's qkv after down_proj) takes parts_in = that array and
 *            x = the residual stream BEFORE the producer's output, and completes, while it seeds its rotation,
 *                x'[k] = round(x[k] + (((parts[k][0] + parts[k][1]) + parts[k][2]) + parts[k][3]))
 *            -- the summation order and the single rounding of the in-launch reducer, so both routes give the same bits.
 *            x_out (optional, must not alias x) receives x' (the new residual stream; written by one workgroup).
 *            One row; in-kernel rotation; the consumer's prologue is NONE or RMSNORM.  A consumer may split K itself -- every K-slice
 *            completes (and, in column block 0, stores to x_out) exactly the channels it reads, so x_out is written piecewise, once
 *            per channel -- and may in turn leave partial sums (parts_in and parts_out in one call: a chain of linears).  The RMSNorm
 *            prologue on a K-split consumer is only defined together with parts_out (no workgroup sees all of x: see below); an
 *            unsplit RMSNorm consumer covers all of K per workgroup.  x_out must not alias x, parts_in, parts_out or y.
 *            parts_out_n: paro_gemv_parts_count(L) -- the split of the launch shape chosen for a launch nobody polls in (the
 *            narrow deep linears as in the automatic shape; mid-width ones such as qkv split 2-way); 0 = this layer does
 *            not split, use the ordinary route -- or any 2..PARO_MAX_PARTIALS that K / 128 can be cut into.
 *            RMSNorm prologue + parts_out: no workgroup of a K-split launch sees all of x, so the norm's scalar travels with the
 *            partial sums: the buffer is [N + 1][PARO_MAX_PARTIALS], the partial sums are UN-normalised and row N holds the K-slices'
 *            sums of squares in the same slots; the consumer multiplies by rsqrt(sum(row N) / K + eps) (paro_attn_decode_parts for
 *            the qkv projection: each q / k / v element is read by one workgroup there, so completing it costs next to nothing).
 *            paro_parts_finish completes the sum without a linear behind it (the last layer's down_proj in front of the
 *            final norm): out[k] = round(x[k] + sum) in the same order; x may be NULL.
 * rows <= 4, krot <= 8 (in-kernel rotation); the launch shape is chosen automatically. */
#define PARO_MAX_PARTIALS 4
#define PARO_PROLOGUE_NONE 0
#define PARO_PROLOGUE_RMSNORM 1
#define PARO_PROLOGUE_SILU_MUL 2
#define PARO_PROLOGUE_GELU_TANH_MUL 3   /* v11: x_k = gelu_tanh(gate_k) * up_k -- the Gemma families' MLP activation (the reference lists
                                           gemma-4 checkpoints, README.md:89-93); same input layout as SILU_MUL */
/* v18 -- ATTENTION TAIL of the qkv projection (decode harness): the batch-1 decode attention that consumes this
 * projection's q / k / v (paro_attn_decode_split on partial sums) runs in the SAME launch.  The grid gets one more row, dispatched behind
 * every projection workgroup, whose workgroups are the attention's (KV head, 64 / 128-position chunk): they request their K / V cache
 * lines at once (nothing of that depends on this token) while the projection streams its weights, then poll q / k / v, which the
 * projection's K-slices leave as 8-byte {fp32 partial sum, launch tag} granules (write-through; tag = the hardware dispatch id of the
 * launch, as in GEMV mode 3) -- one in-launch hand-over instead of a launch boundary plus the attention's cold start.  Same arithmetic,
 * same bits as paro_w4a16_gemv_fused(parts_out) followed by paro_attn_decode_split(qkv_parts).
 * Needs: one row, PARO_PROLOGUE_RMSNORM, parts_out (whose buffer is then [(N + 1)][PARO_MAX_PARTIALS] 8-BYTE granules, i.e. twice the
 * floats, 16-byte aligned, and is NOT readable as plain partial sums afterwards), N == (n_heads + 2 n_kv_heads) * head_dim, head_dim
 * 128, at most 4 query heads per KV head, group_size 128, a launch shape of 4 or 8 waves that K-splits (paro_attn_tail_supported;
 * otherwise PARO_ERR_UNSUPPORTED: use the two launches).  The remaining fields are paro_attn_decode_split's arguments of the same names. */
typedef struct paro_attn_tail {
  void* kcache;
  void* vcache;
  float* attn_parts;
  const int32_t* pos;
  const float* rope;
  const void* q_norm_w;
  const void* k_norm_w;
  float eps, scale;
  int32_t n_heads, n_kv_heads, head_dim, max_positions;
  void* workspace;
  int64_t workspace_bytes;
} paro_attn_tail_t;
typedef struct paro_fusion {
  int32_t prologue;
  float eps;             /* RMSNorm epsilon */
  int64_t x_stride;      /* elements between rows of x; 0 = dense (K, or 2 K for SILU_MUL) */
  const void* residual;
  const void* const* ar_peers; /* HOST array [ar_world] of the ranks' buffers as mapped in this process (copied into the
                                  launch arguments), or NULL */
  void* ar_own;                /* this rank's buffer (== ar_peers[ar_rank]) */
  void* ar_state;              /* (16 + ar_max_elems / 16) u32 of ORDINARY device memory, zero when the buffers are created,
                                  never touched by the caller afterwards: give-up flag + one epoch per 16-column tile */
  int32_t ar_world, ar_rank;
  int64_t ar_max_elems;        /* the element count the buffers were sized with */
  float* parts_out;            /* v12: fp32 [N][PARO_MAX_PARTIALS], or NULL */
  const float* parts_in;       /* v12: fp32 [K][PARO_MAX_PARTIALS] left by the producer of x's missing term, or NULL */
  void* x_out;                 /* v12: act_dtype [K], the completed x (with parts_in), or NULL */
  int32_t parts_out_n;         /* v12: the K-split of the parts_out launch */
  int32_t attn_head_dim;       /* v14: head_dim of the attention that produced attn_in (64, 128 or 256; divides K) */
  const float* attn_in;        /* v14: x is the attention output handed over UN-MERGED by paro_attn_decode_split: fp32
                                  [K][4] + [K / head_dim][8] (paro_attn_parts_floats), 16-byte aligned; `x` is ignored (may be NULL).  This launch
                                  completes out = sum_c 2^(m_c - M) o_c / sum_c 2^(m_c - M) l_c per element while it seeds its rotation --
                                  the same bits as paro_attn_finish followed by the plain launch.  One row, no prologue, no residual;
                                  parts_out allowed (o_proj of a decoder layer: attention slots in, K-split partial sums out). */
  const paro_attn_tail_t* attn_tail;   /* v18: run the consuming decode attention in this launch (above), or NULL */
} paro_fusion_t;
int paro_gemv_parts_count(const paro_linear_t* L);
/* v18: 1 when paro_fusion_t.attn_tail can run this qkv projection with its decode attention in one launch (the conditions above), 0 when
 * the two launches are the route, < 0 on a bad descriptor. */
int paro_attn_tail_supported(const paro_linear_t* L, int n_heads, int n_kv_heads, int head_dim, int max_positions);
int paro_parts_finish(const void* x, const float* parts, int64_t K, void* out, int act_dtype, void* stream);
int paro_w4a16_gemv_fused(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                          int64_t workspace_bytes, const paro_fusion_t* fusion, void* stream);

/* Mixture-of-experts decode: the SAME fused GEMV over `n_slots` (token, expert) slots in one launch.
 * All experts of a projection share one rotation (cli/convert.py:280-379 exports one pairs / theta / channel_scales
 * set per projection; mlx/modules.py:159-212 applies it once before gate/up and once before down), so `L` carries
 * that rotation and the packed buffers of expert 0, and slot s reads
 *     wq + expert_idx[s] * wq_stride_bytes,  sz + expert_idx[s] * sz_stride_bytes,
 *     x + (s / x_slot_div) * x_slot_stride,  writes y + s * y_slot_stride.
 * expert_idx lives in DEVICE memory (router output; graph-capturable).  With the SILU_MUL prologue this is the
 * experts' down projection on the slot's own gate|up vector; without a prologue the merged gate_up projection
 * (x_slot_div = experts per token: the k slots of a token share its x).  rows <= 4, no K-split, no residual. */
typedef struct paro_experts {
  const int32_t* expert_idx;
  int32_t n_slots;
  int32_t x_slot_div;
  int64_t wq_stride_bytes, sz_stride_bytes;
  int64_t x_slot_stride, y_slot_stride;   /* elements */
  int32_t n_experts;                      /* v13: experts behind L->wq / L->sz.  A slot whose id is outside [0, n_experts) reads expert 0's
                                             weights (never out of bounds) and its outputs are NaN -- checked on the DEVICE, so it also holds
                                             under HIP-graph replay, where no host check can run.  Must be >= 1. */
  int32_t reserved0;
} paro_experts_t;
int paro_w4a16_gemv_experts(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                            int64_t workspace_bytes, const paro_fusion_t* fusion, const paro_experts_t* experts,
                            void* stream);

/* Prefill path (any rows): rotate pre-pass into the workspace, then an
 * LDS-staged MFMA GEMM with in-register INT4 dequant.
 * `variant` selects the kernel: 0 = auto (by rows / grid size / dtype);
 *   1 = 128x128 tile, 4 waves, per-group scale epilogue (f16 / bf16; the small-M fallback);
 *   2 = 256x128 tile, 4 waves, 16x16x32 MFMA, exact fp16 weights in registers (f16 only; K-split at small M);
 *   (3 = the round-1 256x256 kernel: removed in v11, the value is refused);
 *   4 = 256x256 tile, 8 waves as 1 x 8 (every wave owns 32 distinct columns over all 256 rows: each INT4
 *       word is dequantised exactly once per workgroup), 32x32x16 MFMA, f16 and native bf16.
 * Every variant computes the same function; tests force each one at small sizes through this knob.
 * (41..43: timing-only ablation builds of variant 4, wrong results.  44: EXPERIMENT -- the rotation fused into variant 4's LDS stage, one
 * launch, no rotated copy of x; needs L->rmat, fp16, group_size 128; same bits as the pre-pass + variant 4 and 2.4..3.3x slower at
 * M = 65536, profiles/NOTES.md 6.11 -- never selected automatically.) */
#define PARO_GEMM_AUTO 0
/* v19: the block shape GEMM variant 4 runs `rows` rows of this layer with -- rows per block (64 / 128 / 256) and K-splits (csrc/gemm.hip
 * g4_shape: one round of workgroups over the 256 CUs with as few fp32 partial tiles as possible) -- host-only, touches no device memory:
 * for tooling, logs and tests of the rule, as paro_gemv_launch_shape is for the GEMV. */
int paro_gemm_launch_shape(const paro_linear_t* L, int64_t rows, int* block_rows, int* ksplit);
int paro_w4a16_gemm(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                    int64_t workspace_bytes, int variant, void* stream);

/* Grouped prefill GEMM over expert segments (v11).  The reference's MoE export keeps ONE rotation per
 * projection shared by all experts (cli/convert.py:280-379) and its MLX back-end rotates the tokens once before the routed
 * experts (mlx/modules.py:159-212); here the caller rotates once (rotation::rotate), sorts the (token, expert) pairs by
 * expert ON THE DEVICE, pads every expert's segment to a multiple of block_rows (64, 128 or 256) and passes
 *   x_rot         act_dtype [padded_rows][K]   rotated rows in that order (padding rows: anything finite)
 *   block_expert  int32 [padded_rows / block_rows] in DEVICE memory: the expert whose packed weights row block b uses,
 *                 L->wq + e * wq_stride_bytes / L->sz + e * sz_stride_bytes; -1 = block unused; an id >= n_experts (v13) is
 *                 treated as unused on the device (its rows of y are not written): never an out-of-bounds weight read
 *   y             act_dtype [padded_rows][N]
 * One launch of GEMM variant 4 for all experts; no host synchronisation; HIP-graph capturable.  L: one rotation partition. */
int paro_w4a16_gemm_grouped(const paro_linear_t* L, const void* x_rot, void* y, int64_t padded_rows, int block_rows,
                            const int32_t* block_expert, int64_t wq_stride_bytes, int64_t sz_stride_bytes, int32_t n_experts,
                            void* stream);

/* Dispatcher used by the Python operator (RotateQuantizedLinear.forward, transformers/modules.py:57-71; ParoQuantLinearMethod.apply,
 * vllm/plugin.py:281-311): rows <= 16 the fused GEMV (mode auto); 17..32 rows -- and up to 64 rows on outputs below 1024 column tiles --
 * the schedule pre-pass + the GEMV on 2 / 4 MFMA row tiles (mode 1); everything else pre-pass + MFMA GEMM.  Environment (A/B runs):
 * PARO_SKINNY=0 (GEMM above 16 rows), PARO_SKINNY_MAX=<rows> (last row count of the GEMV route), PARO_PREROT_SCHED=0 (the stage-kernel
 * pre-pass of rounds 1..5 instead of the schedule pre-pass). */
int paro_w4a16_linear(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* Batch-1 decode attention of one layer in one launch (the decode harness; the reference leaves
 * attention to HF generate() / vLLM, transformers/generator.py:37-67): optional per-head q / k RMSNorm (Qwen3),
 * rotary embedding (rotate_half convention), KV-cache append at *pos, grouped-query attention over 0..*pos.
 *   qkv     act_dtype [(n_heads + 2 n_kv_heads) * head_dim]   (output of the merged qkv projection)
 *   kcache          act_dtype [n_kv_heads][max_positions][head_dim]
 *   vcache          act_dtype [n_kv_heads][head_dim][max_positions]  (position-contiguous: the P V product reads it as
 *                   MFMA fragments); max_positions a multiple of 8
 *   out     act_dtype [n_heads * head_dim]
 *   pos     int32 in DEVICE memory (a captured graph replays for every token)
 *   rope    fp32 [max_positions][head_dim]: cos then sin (head_dim / 2 each) of every position
 * head_dim in {64, 128}; n_heads / n_kv_heads <= 8.  The grid is (KV heads) x (chunks of 256 positions) over
 * max_positions, so that one captured graph serves every position; chunks beyond *pos exit at once.  With more than
 * one active chunk the partial results are merged in-launch by the last workgroup of a KV head to arrive
 * (ticket in `workspace`, agent-scope release / acquire, no spinning).
 *   workspace  paro_attn_decode_workspace_bytes(...) bytes, zero-filled ONCE by the caller (the tickets return to zero). */
int64_t paro_attn_decode_workspace_bytes(int n_heads, int n_kv_heads, int head_dim, int max_positions);
int paro_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                     const void* q_norm_w, const void* k_norm_w, float eps, float scale, int n_heads, int n_kv_heads,
                     int head_dim, int max_positions, int act_dtype, void* workspace, int64_t workspace_bytes,
                     void* stream);
/* v12: the same with q / k / v arriving as the partial sums a K-split qkv projection left (paro_fusion_t.parts_out):
 *   qkv_parts  fp32 [(n_heads + 2 n_kv_heads) * head_dim + 1][PARO_MAX_PARTIALS]
 *   element    = round((((p0 + p1) + p2) + p3) * rstd) in act_dtype -- what the projection's own epilogue would have stored --,
 *                rstd = rsqrt(sum(last row) / norm_dim + norm_eps) when norm_dim > 0 (the projection ran with the RMSNorm prologue
 *                over norm_dim = hidden channels), 1 when norm_dim == 0. */
int paro_attn_decode_parts(const float* qkv_parts, int64_t norm_dim, float norm_eps, void* kcache, void* vcache, void* out,
                           const int32_t* pos, const float* rope, const void* q_norm_w, const void* k_norm_w, float eps,
                           float scale, int n_heads, int n_kv_heads, int head_dim, int max_positions, int act_dtype,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* v14: split attention -- the merge over position chunks is left to the CONSUMER of the attention output.  The launch works in chunks of
 * 128 positions, groups the active chunks into at most FOUR slots and stores per slot the un-normalised triple
 *   attn_parts  fp32, paro_attn_parts_floats(n_heads, head_dim) elements, 16-byte aligned, zero-filled once by the caller:
 *               [n_heads * head_dim][4]  o_c[d] = sum_{p in slot c} 2^((s_p - m_c) log2 e) v_p[d]
 *               [n_heads][8]             m_c (slots 0..3), l_c (slots 0..3); a slot nobody filled: (-3e38, 0)
 * (one chunk per slot up to 512 positions: no ticket, no fence; beyond that the chunks of a slot meet at a per-slot ticket).  The output is
 *   out[j][d] = round(sum_c 2^((m_c - M) log2 e) o_c[d] / sum_c 2^((m_c - M) log2 e) l_c),  M = max_c m_c
 * completed either by the next fused GEMV (paro_fusion_t.attn_in: o_proj reads the slots as its x) or by paro_attn_finish; both compute
 * the same expression operation for operation (same bits).  Exactly one of qkv / qkv_parts (the two input forms above).  Why: at
 * short contexts the in-launch merge is ~2.5 us of an ~8 us launch and one workgroup per KV head ingests the whole cache
 * (profiles/r04_attn_stage_ablation.jsonl). */
int64_t paro_attn_parts_floats(int n_heads, int head_dim);
int paro_attn_decode_split(const void* qkv, const float* qkv_parts, int64_t norm_dim, float norm_eps, void* kcache, void* vcache,
                           float* attn_parts, const int32_t* pos, const float* rope, const void* q_norm_w, const void* k_norm_w,
                           float eps, float scale, int n_heads, int n_kv_heads, int head_dim, int max_positions, int act_dtype,
                           void* workspace, int64_t workspace_bytes, void* stream);
int paro_attn_finish(const float* attn_parts, int n_heads, int head_dim, void* out, int act_dtype, void* stream);

/* v19 -- the PROMPT pass of the decode harness: the element-wise work between a decoder layer's four fused linears at
 * `rows` prompt positions, three launches instead of ~45 framework operators per layer.  The reference's prompt pass is HF's modelling
 * code under generate() (transformers/generator.py:37-67); the harness folds the RMSNorm weights into the consumers' channel scales,
 * which leaves the per-row scalar.  csrc/prompt.hip states the rounding points (those of the framework expressions they replace).
 *   paro_prompt_row_rms    rs[t] = rsqrt(mean_k h[t][k]^2 + eps), fp32 [rows]; hidden a multiple of 8
 *   paro_prompt_qkv_post   qkv [rows][(n_heads + 2 n_kv_heads) head_dim] = the merged projection's raw output; * rs[t] (NULL: 1), q / k
 *                          per-head RMSNorm with q_norm_w / k_norm_w [head_dim] (NULL: none), rotary embedding of positions pos0 + t from
 *                          rope fp32 [max_positions][head_dim] (cos | sin halves) -> q_out [rows][n_heads][head_dim], k_out / v_out
 *                          [rows][n_kv_heads][head_dim] AND the decode caches kcache [n_kv_heads][max_positions][head_dim], vcache
 *                          [n_kv_heads][head_dim][max_positions] (paro_attn_decode's layouts); head_dim even, <= 128
 *   paro_prompt_silu_mul   out[t][i] = silu(g) * u,  g / u = gate_up[t][i | inter + i] * rs[t] (NULL: 1); inter a multiple of 8 */
int paro_prompt_row_rms(const void* h, float* rs, int64_t rows, int64_t hidden, float eps, int act_dtype, void* stream);
int paro_prompt_qkv_post(const void* qkv, const float* rs, const void* q_norm_w, const void* k_norm_w, const float* rope,
                         void* q_out, void* k_out, void* v_out, void* kcache, void* vcache, int64_t rows, int pos0,
                         int n_heads, int n_kv_heads, int head_dim, int max_positions, float eps, int act_dtype, void* stream);
int paro_prompt_silu_mul(const void* gate_up, const float* rs, void* out, int64_t rows, int64_t inter, int act_dtype, void* stream);


/* Qwen3.5 in the decode harness (v13; BASELINE configs 3 and 5): the hybrid decoder's token mixers at batch 1, following transformers'
 * models/qwen3_5 (the reference decodes any HF architecture through generate(), transformers/generator.py:37-67).
 *   paro_gdn_prep   gated delta net, in front of the recurrence: causal_conv1d_update + SiLU over the in_proj_qkv outputs
 *                   (conv_state: act_dtype [conv_dim][4], elements 1..3 = the last three inputs, updated in place; conv_w fp32
 *                   [conv_dim][4] = conv1d.weight), and the two dense projections in_proj_a / in_proj_b of the RMS-normalised hidden state
 *                   (w_ab fp32 [2 n_v_heads][hidden]: in_proj_a rows then in_proj_b rows with the input norm's (1 + w) folded in; x = the
 *                   un-normalised residual stream, the norm's scalar is applied here):
 *                   g_beta[h] = exp(-exp(A_log[h]) * softplus(a[h] + dt_bias[h])),  g_beta[n_v_heads + h] = sigmoid(b[h])
 *   paro_gdn_step   torch_recurrent_gated_delta_rule for one token per value head (key / value head dims 128): q, k l2-normalised (q scaled
 *                   by 128^-1/2), S *= decay, delta = (v - S^T k) beta, S += k delta^T, o = S^T q, then Qwen3_5RMSNormGated with z;
 *                   conv_out = [q heads | k heads | v heads] from paro_gdn_prep; state fp32 [n_v_heads][128][128], updated in place; four
 *                   workgroups per value head (32 value columns each), the last to arrive normalises the head (workspace: raw outputs + tickets)
 *   paro_attn_decode_gated   Qwen3_5Attention at one row: qkv = [n_heads][2][head_dim] (query | gate per head) then k, v heads; q / k
 *                   RMSNorm with weights w (norm_plus_one: 1 + w), rotary embedding on the first rotary_dim dimensions (rope fp32
 *                   [max_positions][rotary_dim]: cos then sin), KV append at *pos (kcache / vcache act_dtype [n_kv_heads][max_positions]
 *                   [head_dim]), attention over 0..*pos, output * sigmoid(gate).  head_dim 256.  *pos is read on the DEVICE: a value
 *                   outside [0, max_positions) appends nothing and the output rows are NaN (no host check is possible under graph replay).
 *                   paro_gdn_fused_step uses *pos only for the parity of its double-buffered convolution state (any value is in bounds). */
int paro_gdn_prep(const void* qkv, const void* x, const float* w_ab, float eps, void* conv_state, const float* conv_w, const float* A_log,
                  const float* dt_bias, void* conv_out, float* g_beta, int hidden, int conv_dim, int n_v_heads, int act_dtype, void* stream);
int64_t paro_gdn_workspace_bytes(int n_v_heads);   /* scratch of paro_gdn_step: zero-filled ONCE by the caller (the arrival tickets return to zero) */
int paro_gdn_step(const void* conv_out, const void* z, const float* g_beta, float* state, const void* norm_w, float eps, void* out,
                  int n_k_heads, int n_v_heads, int act_dtype, void* workspace, void* stream);
/* v14: paro_gdn_prep folded into paro_gdn_step -- one launch per gated-delta-net block and token.  qkvz = the in_proj output [conv_dim +
 * value_dim] (the convolution's inputs, then z); conv_state is DOUBLE-BUFFERED by the token's parity, act_dtype [2][conv_dim][4]: the launch
 * reads buffer (*pos & 1) and writes buffer ((*pos + 1) & 1) (workgroups that share a key head compute the same q / k channels: nobody reads
 * what another workgroup of the launch has written, duplicate writers store identical values); every other argument as in the two calls.
 * The same arithmetic, operation for operation. */
int paro_gdn_fused_step(const void* qkvz, const void* x, const float* w_ab, float eps_in, void* conv_state, const float* conv_w,
                        const float* A_log, const float* dt_bias, float* state, const void* norm_w, float eps, void* out,
                        const int32_t* pos, int hidden, int conv_dim, int n_k_heads, int n_v_heads, int act_dtype, void* workspace,
                        void* stream);
/* v14: the recurrence of paro_gdn_step over n_tokens tokens in ONE launch (the prompt pass of a gated-delta-net layer): the state is read
 * once and written once, conv_out [n_tokens][2 key_dim + value_dim] and g_beta [n_tokens][2 n_v_heads] are per-token rows as paro_gdn_prep
 * lays them out, out_raw fp32 [n_tokens][n_v_heads * 128] receives o = S^T q per token BEFORE the gated RMSNorm (which needs whole heads
 * and is row-parallel over the tokens: left to the caller, like the causal convolution in front). */
int paro_gdn_sequence(const void* conv_out, const float* g_beta, float* state, float* out_raw, int n_tokens, int n_k_heads,
                      int n_v_heads, int act_dtype, void* stream);
int paro_attn_decode_gated(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                           const void* q_norm_w, const void* k_norm_w, int norm_plus_one, float eps, float scale, int n_heads,
                           int n_kv_heads, int head_dim, int rotary_dim, int max_positions, int act_dtype, void* stream);

/* Tail of a decode step (decode harness): final RMSNorm + the unquantised lm_head as an HBM-bound
 * fp16 / bf16 matrix-vector product, then the greedy argmax.
 *   paro_lm_head         logits[v] = sum_k W[v][k] xn[k],  xn = rmsnorm(x; norm_weight, eps) with HF's rounding;
 *                        W act_dtype [vocab][hidden] row-major, hidden = 512 x 1..8; also leaves every
 *                        workgroup's (max logit, lowest index) in `workspace` (paro_lm_head_workspace_bytes).
 *   paro_argmax_advance  out_tokens[*pos] = *token (the token just consumed; skipped when out_tokens is NULL);
 *                        *token = argmax(logits) (lowest index on ties); *pos += 1.   All three live in device memory,
 *                        so a captured decode step replays without a host round trip. */
int64_t paro_lm_head_workspace_bytes(int64_t vocab);
int paro_lm_head(const void* x, const void* norm_weight, const void* W, void* logits, int64_t vocab, int64_t hidden,
                 float eps, int act_dtype, void* workspace, int64_t workspace_bytes, void* stream);
int paro_argmax_advance(const void* workspace, int64_t vocab, int64_t* token, int32_t* pos, int64_t* out_tokens,
                        int64_t out_len, void* stream);

/* ---------------------------------------------------------------------------
 * Decode chain (v11): the same linear on activations that ARRIVE rotated, with the NEXT linear's rotation applied by
 * the launch that produces its input.  The reference runs `rotate -> GEMM` per linear (transformers/modules.py:57-71,
 * vllm/plugin.py:281-311); the rotation is block-diagonal over 128 channels, so the workgroup that finishes a
 * 128-column block of linear i can apply linear i+1's pairs / theta / channel_scales to it once, instead of every
 * workgroup of linear i+1 rotating all of x again:
 *     y       = x_rot @ dequant(W) * rstd + bias + residual          (rstd from ssq_in, or 1)
 *     next_x_rot[p'] = rotate_{next, p'}(act(y[:, col0 : col0 + next.K]) * next.channel_scales[p'])
 *   x_rot        act_dtype [n_parts][rows][K]: partition p rotated with L's partition-p parameters
 *                (paro_rotate_parts, rotation::rotate per partition, or a producer's next_x_rot)
 *   y            act_dtype [rows][N], or NULL when only the consumer reads the result
 *   residual     act_dtype [rows][N] added before the one rounding, or NULL
 *   ssq_in       fp32 [rows][ssq_in_blocks]: partial sums of squares of the un-normalised vector that was rotated into
 *                x_rot (a producer's ssq_out): y is scaled by rsqrt(sum / norm_dim + eps) -- the RMSNorm in front of
 *                this linear, its weight folded into channel_scales (paro_fusion_t, PARO_PROLOGUE_RMSNORM); or NULL
 *   ssq_out      fp32 [rows][N / 128]: sum of squares of each 128-column block of the rounded y, or NULL
 *   next         the consuming linear (its rot / channel_scales / n_parts / K are read), or NULL
 *   next_x_rot   act_dtype [next.n_parts][rows][next.K]
 *   next_col0    first column of y the consumer reads (multiple of 128)
 *   next_act     PARO_CHAIN_ACT_NONE, or PARO_CHAIN_ACT_SILU_MUL: L is the merged gate|up projection (two partitions of
 *                next.K columns) and the consumer reads silu(gate) * up (mlx/modules.py:204-207 rotates the activation
 *                output before down_proj the same way)
 * rows 1..16; group_size 128; every partition a multiple of 128 columns; ksplit / waves: 0 = automatic.  The K-split of
 * this family tags its granules with (block, per-block epoch) -- the epochs live in the first PARO_WS_COUNTER_BYTES of
 * the workspace (zero-filled once by the caller, like the counters of paro_w4a16_gemv, with which the workspace can be
 * shared); workspace >= paro_chain_workspace_bytes(L, rows). */
#define PARO_CHAIN_ACT_NONE 0
#define PARO_CHAIN_ACT_SILU_MUL 1
#define PARO_CHAIN_ACT_GELU_TANH_MUL 2   /* gelu_tanh(gate) * up (Gemma) */
typedef struct paro_chain {
  const void* x_rot;
  void* y;
  const void* residual;
  const float* ssq_in;
  int32_t ssq_in_blocks;
  float eps;
  int64_t norm_dim;
  float* ssq_out;
  const struct paro_linear* next;
  void* next_x_rot;
  int64_t next_col0;
  int32_t next_act;
  int32_t reserved0;
} paro_chain_t;
int64_t paro_chain_workspace_bytes(const paro_linear_t* L, int64_t rows);
int paro_chain_launch_shape(const paro_linear_t* L, const paro_chain_t* C, int64_t rows, int* ksplit, int* waves);
int paro_w4a16_gemv_chain(const paro_linear_t* L, const paro_chain_t* C, int64_t rows, void* workspace,
                          int64_t workspace_bytes, int ksplit, int waves, void* stream);
/* Head of a chain: x [rows][K] rotated with every partition's parameters of L into x_rot [n_parts][rows][K] in ONE
 * launch (the stage kernel behind rotation::rotate, rotation.cu:10-43; with L->rmat and >= 256 rows the dense per-group product on
 * the matrix cores, i.e. exactly the pre-pass paro_w4a16_gemm runs in front of its GEMM kernel). */
int paro_rotate_parts(const paro_linear_t* L, const void* x, void* x_rot, int64_t rows, void* stream);

/* (v17: the two persistent decode engines -- paro_engine_* / paro_engine2_* of v13 / v15 -- lost to the per-call launches on every
 * measured shape (profiles/NOTES.md 4.2, 5.1) and left the default library: `make EXPERIMENTAL=1` builds them,
 * include/paro_abi_experimental.h declares them.) */

/* Dequantise packed weights back to a dense [K, N] matrix of act_dtype
 * (debug / verification aid; W[k,n] = (q - z) * s rounded once). */
int paro_dequant_packed(const paro_linear_t* L, void* out_w, void* stream);

/* ---------------------------------------------------------------------------
 * One-shot all-reduce(SUM) of a small activation vector across the ranks of one node: the collective after the
 * row-parallel linears (o_proj, down_proj) of tensor-parallel decode, which the reference leaves to vLLM's
 * RowParallelLinear (SURVEY section 8 row e).  One launch of one workgroup per rank, no host work, HIP-graph capturable:
 * every rank stores its vector straight into a slot of every peer's buffer (xGMI peer stores), raises a flag there, waits
 * (bounded) for the world's flags in its own buffer and sums the slots in rank order in fp32 -- bit-identical on all
 * ranks.
 *   buffer: paro_allreduce_buffer_bytes(world, max_elems) bytes per rank of fine-grained device memory
 *           (paro_allreduce_buffer_create), mapped into every peer (paro_allreduce_buffer_open on the 64-byte handle);
 *           peers = HOST array of the world's buffer addresses as mapped in the calling process, own buffer at [rank]
 *           (v9: copied into the launch arguments -- a pointer fetched from device memory per peer was a dependent round
 *           trip in front of every store).
 *   n: elements, a multiple of 8, <= max_elems (the value the buffer was sized with).  x and y may alias.
 *   residual: optional vector added to the sum before the one rounding (y = sum_r x_r + residual: the decoder's residual
 *             stream after o_proj / down_proj), or NULL.
 * A peer that never arrives makes the call give up after a bounded spin: word 1 of the buffer becomes
 * PARO_WS_STATUS_GIVEUP (sticky) and the sum is garbage -- the host checks that word at teardown / after warm-up; calls
 * on a buffer that has given up poll once instead of waiting again (ranks out of step cost the bound once, not per call). */
int64_t paro_allreduce_buffer_bytes(int world, int64_t max_elems);
/* Setup-time helpers (they allocate and synchronise): the buffer must be FINE-GRAINED device memory -- peers write into it
 * and the owner polls it within one kernel, and ordinary device memory is coherent across GPUs only at kernel boundaries.
 * create: hipExtMallocWithFlags(hipDeviceMallocFinegrained) + zero fill + hipIpcGetMemHandle (64 opaque bytes to hand to the
 * peers); open / close: hipIpcOpenMemHandle / hipIpcCloseMemHandle in a peer process; destroy: hipFree in the owner;
 * status: synchronises `stream`, PARO_ERR_LAUNCH if a call ever gave up. */
int paro_allreduce_buffer_create(int64_t bytes, void** out_ptr, void* out_handle64);
int paro_allreduce_buffer_open(const void* handle64, void** out_ptr);
int paro_allreduce_buffer_close(void* peer_ptr);
int paro_allreduce_buffer_destroy(void* own_ptr);
int paro_allreduce_status(const void* own_ptr, void* stream);
int paro_allreduce_oneshot(const void* x, const void* residual, void* y, int64_t n, int act_dtype, const void* const* peers,
                           int world, int rank, int64_t max_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARO_ABI_H */

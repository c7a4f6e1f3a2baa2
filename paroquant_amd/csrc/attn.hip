// Batch-1 decode attention for the decode harness: everything between the merged qkv projection
// and o_proj of one decoder layer in ONE launch -- per-head q/k RMSNorm (Qwen3), rotary embedding, KV-cache append,
// grouped-query attention over positions 0..pos, softmax, P V.  The reference leaves this to HF generate() / vLLM;
// here it exists so that a whole decode step is five launches per layer (qkv GEMV, this, o GEMV, gate_up GEMV,
// down GEMV) inside one HIP graph, and the end-to-end tokens/s of north_star can be measured.
//
// Grid = (KV heads, position chunks of 256 -- 128 for caches above 512 positions).  A workgroup handles one chunk of
// one KV head for all of that head's n_rep = Hq / Hkv query heads (they share every K / V byte): every global load of
// the chunk -- K and V as MFMA B fragments, 16 bytes per lane and load -- is requested before anything else; scores and
// P V run on the matrix cores, the soft-max of a wave's 64 / 32 positions in registers.  A CU ingests ~13 B / clock, so
// longer contexts are spread over many CUs rather than one workgroup per head.  The V cache is kept position-contiguous
// ([head][dim][position]) so that its fragments are 16-byte loads too.  With more than one active chunk the partial
// (max, sum, unnormalised output) triples go to a workspace and the LAST workgroup of a KV head to arrive (write-through
// stores -> ticket -> agent-scope loads, see st_agent below; no spinning, so no dependence on dispatch order) merges them, reading eight chunks'
// triples at a time.  The position comes from DEVICE memory, so one captured graph replays for every token: the grid
// always covers max_positions, chunks beyond `pos` exit at once.
//
// SPLIT builds (paro_attn_decode_split): the merge is NOT done here.  Chunks are 64 positions below 256 positions and 128 from there on
// (one kernel, both paths; the grid is sized for 64); the active chunks are grouped into at most FOUR slots (one chunk per slot up to
// 256 / 512 positions -- no ticket, the workgroup stores its (max, sum, un-normalised output) triple and is done; beyond that the
// chunks of a slot meet at a per-slot ticket and the last one stores the slot's triple)
// and the launch that consumes the attention output -- o_proj, `paro_fusion_t.attn_in` -- completes
//     out[j][d] = sum_c 2^(m_c - M) o_c[d] / sum_c 2^(m_c - M) l_c
// while it seeds its rotation: each element is read by the few workgroups whose K-slice holds it.  At short contexts that
// removes the whole in-launch merge (~2.5 us of an ~8 us launch) and puts the chunks of a head on different CUs.
#include "attn_impl.hpp"

namespace paro {

// The consumer's completion of a SPLIT launch, as its own launch (tests; callers whose next launch is not a fused GEMV): the same
// expression, operation for operation, as the attn_in prologue of gemv_kernel (attn_merge, common.hpp).
template <typename AT>
__global__ __launch_bounds__(256) void attn_finish_kernel(const f32x4* o, const f32x4* ml, unsigned short* out, int n, int hd_shift) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int j = e >> hd_shift;
  out[e] = Act<AT>::from_f32(attn_merge(o[e], ml[2 * j], ml[2 * j + 1]));
}

}  // namespace paro

extern "C" int64_t paro_attn_parts_floats(int n_heads, int head_dim) {
  if (n_heads < 1 || head_dim < 2) return -1;
  return (int64_t)n_heads * head_dim * 4 + (int64_t)n_heads * 8;
}

extern "C" int paro_attn_finish(const float* attn_parts, int n_heads, int head_dim, void* out, int act_dtype, void* stream) {
  using namespace paro;
  if (!attn_parts || !out) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_heads < 1 || (head_dim != 64 && head_dim != 128 && head_dim != 256)) return fail(PARO_ERR_INVALID, "head_dim must be 64, 128 or 256");
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  const int n = n_heads * head_dim, shift = head_dim == 64 ? 6 : (head_dim == 128 ? 7 : 8);
  const f32x4* o = (const f32x4*)attn_parts;
  const f32x4* ml = (const f32x4*)(attn_parts + (int64_t)n * 4);
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(attn_finish_kernel<f16>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, o, ml, (unsigned short*)out, n, shift);
  else hipLaunchKernelGGL(attn_finish_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, o, ml, (unsigned short*)out, n, shift);
  return check_launch("paro_attn_finish");
}

extern "C" int64_t paro_attn_decode_workspace_bytes(int n_heads, int n_kv_heads, int head_dim, int max_positions) {
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || head_dim < 2 || max_positions < 1) return -1;
  const int64_t n_rep = n_heads / n_kv_heads, chunks = (max_positions + paro::attn_chunk(max_positions) - 1) / paro::attn_chunk(max_positions);
  // tickets (zero-filled by the caller once: [0, 64) per KV head, [64, 320) per (KV head, slot) of the split launch) + partials, sized for
  // 64-position chunks (the split launch's)
  const int64_t chunks128 = (max_positions + 63) / 64;
  return paro::kAttnWsHeader + (int64_t)n_kv_heads * (chunks > chunks128 ? chunks : chunks128) * n_rep * (head_dim + 2) * 4;
}

namespace paro {
static int attn_decode_impl(const void* qkv, const float* qkv_parts, int64_t norm_dim, float norm_eps, void* kcache, void* vcache, void* out, float* split_out,
                            const int32_t* pos, const float* rope, const void* q_norm_w, const void* k_norm_w, float eps, float scale,
                            int n_heads, int n_kv_heads, int head_dim, int max_positions, int act_dtype, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  const bool parts = qkv_parts != nullptr;
  const bool split = split_out != nullptr;
  if ((!qkv && !parts) || !kcache || !vcache || (!out && !split) || !pos || !rope) return fail(PARO_ERR_INVALID, "null pointer");
  if (parts && norm_dim < 0) return fail(PARO_ERR_INVALID, "norm_dim must be >= 0");
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0) return fail(PARO_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (n_kv_heads > 64) return fail(PARO_ERR_UNSUPPORTED, "at most 64 KV heads");
  if (n_heads / n_kv_heads > 8) return fail(PARO_ERR_UNSUPPORTED, "at most 8 query heads per KV head (got %d)", n_heads / n_kv_heads);
  if (head_dim != 64 && head_dim != 128) return fail(PARO_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return fail(PARO_ERR_INVALID, "q / k norm weights come together");
  const int64_t need = paro_attn_decode_workspace_bytes(n_heads, n_kv_heads, head_dim, max_positions);
  if (max_positions < 8 || max_positions % 8 != 0 || max_positions > 65535 * 128)
    return fail(PARO_ERR_INVALID, "max_positions must be a multiple of 8 in 8..%d (got %d)", 65535 * 128, max_positions);
  static const int env_sc = getenv("PARO_ATTN_SPLIT_CHUNK") ? atoi(getenv("PARO_ATTN_SPLIT_CHUNK")) : 0;   // A/B knob
  const int kChunk = split ? (env_sc == 128 ? 128 : 64) : attn_chunk(max_positions);   // split: the grid is sized for 64-position chunks; the
                                                                                       // kernel itself moves to 128 from 256 positions on
  const bool split_dual = split && env_sc == 0;
  if (!workspace || workspace_bytes < need)
    return fail(PARO_ERR_INVALID, "attention workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
  AttnArgs a;
  a.qkv = (const unsigned short*)qkv;
  a.qkv_parts = (const f32x4*)qkv_parts;
  a.norm_dim = (float)norm_dim;
  a.norm_eps = norm_eps;
  a.kcache = (unsigned short*)kcache;
  a.vcache = (unsigned short*)vcache;
  a.out = (unsigned short*)out;
  a.pos = pos;
  a.rope = rope;
  a.qnw = (const unsigned short*)q_norm_w;
  a.knw = (const unsigned short*)k_norm_w;
  a.ticket = (unsigned*)workspace;
  a.split_o = split_out;
  a.split_ml = split ? split_out + (int64_t)n_heads * head_dim * 4 : nullptr;
  a.part = (float*)((char*)workspace + kAttnWsHeader);
  a.eps = eps;
  a.scale = scale;
  a.Hq = n_heads;
  a.Hkv = n_kv_heads;
  a.hd = head_dim;
  a.T_max = max_positions;
  a.chunks = (max_positions + kChunk - 1) / kChunk;
  static const int env_dbg = getenv("PARO_ATTN_DBG") ? atoi(getenv("PARO_ATTN_DBG")) : 0;
  a.dbg = env_dbg;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)n_kv_heads, (unsigned)a.chunks);
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  const bool h16 = act_dtype == PARO_DTYPE_F16;
  const int n_rep = n_heads / n_kv_heads;
  const int nr = n_rep <= 1 ? 1 : (n_rep <= 2 ? 2 : (n_rep <= 4 ? 4 : 8));
#define PARO_ATTN_LAUNCH(T, HD, NR) \
  do { \
    if (split) { \
      if (split_dual) { \
        if (parts) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 0, true, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 0, false, true>), grid, dim3(256), 0, st, a); \
      } else if (kChunk == 64) { \
        if (parts) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 64, true, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 64, false, true>), grid, dim3(256), 0, st, a); \
      } else if (parts) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 128, true, true>), grid, dim3(256), 0, st, a); \
      else hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 128, false, true>), grid, dim3(256), 0, st, a); \
    } else if (parts) { \
      if (kChunk == 256) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 256, true>), grid, dim3(256), 0, st, a); \
      else hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 128, true>), grid, dim3(256), 0, st, a); \
    } else if (kChunk == 256) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 256, false>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR, 128, false>), grid, dim3(256), 0, st, a); \
  } while (0)
#define PARO_ATTN_NR(T, HD) \
  do { \
    if (nr == 1) PARO_ATTN_LAUNCH(T, HD, 1); \
    else if (nr == 2) PARO_ATTN_LAUNCH(T, HD, 2); \
    else if (nr == 4) PARO_ATTN_LAUNCH(T, HD, 4); \
    else PARO_ATTN_LAUNCH(T, HD, 8); \
  } while (0)
  if (head_dim == 128) {
    if (h16) PARO_ATTN_NR(f16, 128); else PARO_ATTN_NR(bf16, 128);
  } else {
    if (h16) PARO_ATTN_NR(f16, 64); else PARO_ATTN_NR(bf16, 64);
  }
#undef PARO_ATTN_NR
#undef PARO_ATTN_LAUNCH
  return check_launch("paro_attn_decode");
}
}  // namespace paro

extern "C" int paro_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                                const void* q_norm_w, const void* k_norm_w, float eps, float scale, int n_heads,
                                int n_kv_heads, int head_dim, int max_positions, int act_dtype, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  return paro::attn_decode_impl(qkv, nullptr, 0, 0.f, kcache, vcache, out, nullptr, pos, rope, q_norm_w, k_norm_w, eps, scale, n_heads, n_kv_heads, head_dim,
                                max_positions, act_dtype, workspace, workspace_bytes, stream);
}

extern "C" int paro_attn_decode_parts(const float* qkv_parts, int64_t norm_dim, float norm_eps, void* kcache, void* vcache, void* out,
                                      const int32_t* pos, const float* rope, const void* q_norm_w, const void* k_norm_w, float eps,
                                      float scale, int n_heads, int n_kv_heads, int head_dim, int max_positions, int act_dtype,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  if (!qkv_parts) return paro::fail(PARO_ERR_INVALID, "null pointer");
  return paro::attn_decode_impl(nullptr, qkv_parts, norm_dim, norm_eps, kcache, vcache, out, nullptr, pos, rope, q_norm_w, k_norm_w, eps, scale, n_heads,
                                n_kv_heads, head_dim, max_positions, act_dtype, workspace, workspace_bytes, stream);
}

extern "C" int paro_attn_decode_split(const void* qkv, const float* qkv_parts, int64_t norm_dim, float norm_eps, void* kcache, void* vcache,
                                      float* attn_parts, const int32_t* pos, const float* rope, const void* q_norm_w, const void* k_norm_w,
                                      float eps, float scale, int n_heads, int n_kv_heads, int head_dim, int max_positions, int act_dtype,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  if (!attn_parts) return paro::fail(PARO_ERR_INVALID, "null pointer");
  if ((qkv == nullptr) == (qkv_parts == nullptr)) return paro::fail(PARO_ERR_INVALID, "exactly one of qkv / qkv_parts");
  return paro::attn_decode_impl(qkv, qkv_parts, norm_dim, norm_eps, kcache, vcache, nullptr, attn_parts, pos, rope, q_norm_w, k_norm_w, eps, scale,
                                n_heads, n_kv_heads, head_dim, max_positions, act_dtype, workspace, workspace_bytes, stream);
}

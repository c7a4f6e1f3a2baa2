// Prompt pass of the decode harness (ABI v19): the element-wise work BETWEEN the four fused linears of a decoder layer at
// T prompt rows, as three small kernels instead of ~45 framework operators per layer.  The reference's prompt pass is HF's own modelling code
// under generate() (transformers/generator.py:37-67): RMSNorm -> q/k/v projections -> per-head RMSNorm (Qwen3) -> rotary embedding -> KV
// cache -> attention -> o_proj -> residual -> RMSNorm -> gate / up -> SiLU * up -> down_proj -> residual.  The harness folds the two
// RMSNorm WEIGHTS into the consumers' channel scales at load time, which leaves the per-row scalar rsqrt(mean(x^2) + eps):
//   paro_prompt_row_rms    rs[t] = rsqrt(mean_k h[t][k]^2 + eps)                                   one 256-thread workgroup per row
//   paro_prompt_qkv_post   qkv_raw[t] * rs[t] -> q / k head RMSNorm (optional) -> rotary -> q_out[T][Hq][hd], k_out[T][Hkv][hd],
//                          v_out[T][Hkv][hd] (the attention's inputs) AND the decode caches kcache[Hkv][T_max][hd], vcache[Hkv][hd][T_max]
//                                                                                                    one wave per (row, head)
//   paro_prompt_silu_mul   act[t][i] = silu(gate) * up,  gate / up = gate_up_raw[t][i | I + i] * rs[t]
// All three are HBM-bound streams of a few MB (T x hidden elements); what they buy is launches: Qwen3-4B, 128-token prompt, ~1600 framework
// kernels -> ~110, time to first token 16.8 -> see profiles/NOTES.md 6.10.
// Rounding points follow the framework expression they replace -- (raw.float() * rs).to(T); (x.float() * rs_head).to(T) * w; rotary in
// fp32 with one rounding; silu(g).to(T) * u -- so the harness' HF-parity tests hold unchanged.
#include "common.hpp"

namespace paro {

template <typename AT>
__global__ __launch_bounds__(256) void prompt_row_rms_kernel(const unsigned short* __restrict__ h, float* __restrict__ rs, int hidden, float eps) {
  typedef Act<AT> A;
  __shared__ float part[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const unsigned short* row = h + (int64_t)t * hidden;
  float s = 0.f;
  for (int k = tid * 8; k < hidden; k += 256 * 8) {      // hidden is a multiple of 8 (host-checked)
    const u32x4 v = *(const u32x4*)(row + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = A::to_f32((unsigned short)(v[e] & 0xffffu)), b = A::to_f32((unsigned short)(v[e] >> 16));
      s = __builtin_fmaf(a, a, s);
      s = __builtin_fmaf(b, b, s);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) rs[t] = 1.0f / __builtin_sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)hidden + eps);
}

struct QkvPostArgs {
  const unsigned short* qkv;   // [T][(Hq + 2 Hkv) hd] raw projection output
  const float* rs;             // [T] or null (no row scale)
  const unsigned short* qnw;   // [hd] or null
  const unsigned short* knw;   // [hd] or null
  const float* rope;           // [T_max][hd]: cos (first half), sin (second half) of position t
  unsigned short* q_out;       // [T][Hq][hd]
  unsigned short* k_out;       // [T][Hkv][hd]
  unsigned short* v_out;       // [T][Hkv][hd]
  unsigned short* kcache;      // [Hkv][T_max][hd]
  unsigned short* vcache;      // [Hkv][hd][T_max]
  int T, Hq, Hkv, hd, T_max, pos0;
  float eps;
};

// one wave per (row t, head j of Hq + 2 Hkv); lane l < hd / 2 owns the rotary pair (l, l + hd / 2)
template <typename AT>
__global__ __launch_bounds__(256) void prompt_qkv_post_kernel(const QkvPostArgs a) {
  typedef Act<AT> A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.Hq + 2 * a.Hkv;
  const int j = blockIdx.x * 4 + wave, t = blockIdx.y;
  if (j >= H) return;
  const int half = a.hd >> 1;
  const bool on = lane < half;
  const unsigned short* src = a.qkv + ((int64_t)t * H + j) * a.hd;
  const float rs = a.rs ? a.rs[t] : 1.0f;
  // (raw.float() * rs).to(T)
  float x1 = 0.f, x2 = 0.f;
  if (on) {
    x1 = A::to_f32(A::from_f32(A::to_f32(src[lane]) * rs));
    x2 = A::to_f32(A::from_f32(A::to_f32(src[lane + half]) * rs));
  }
  const bool is_q = j < a.Hq, is_k = !is_q && j < a.Hq + a.Hkv;
  if (!is_q && !is_k) {   // value head: scaled copy + cache column
    if (on) {
      const int hv = j - a.Hq - a.Hkv;
      unsigned short* o = a.v_out + ((int64_t)t * a.Hkv + hv) * a.hd;
      const unsigned short b1 = A::from_f32(x1), b2 = A::from_f32(x2);
      o[lane] = b1;
      o[lane + half] = b2;
      unsigned short* vc = a.vcache + (int64_t)hv * a.hd * a.T_max + (a.pos0 + t);
      vc[(int64_t)lane * a.T_max] = b1;
      vc[(int64_t)(lane + half) * a.T_max] = b2;
    }
    return;
  }
  const unsigned short* w = is_q ? a.qnw : a.knw;
  if (w) {   // per-head RMSNorm: ((x.float() * rsqrt(mean(x^2) + eps)).to(T) * w)
    float s = x1 * x1 + x2 * x2;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float r = 1.0f / __builtin_sqrtf(s / (float)a.hd + a.eps);
    if (on) {
      x1 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(x1 * r)) * A::to_f32(w[lane])));
      x2 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(x2 * r)) * A::to_f32(w[lane + half])));
    }
  }
  if (!on) return;
  // rotary embedding of position pos0 + t (cos / sin in the activation type, as the framework expression casts them)
  const float* rp = a.rope + (int64_t)(a.pos0 + t) * a.hd;
  const float c = A::to_f32(A::from_f32(rp[lane])), sn = A::to_f32(A::from_f32(rp[lane + half]));
  const unsigned short o1 = A::from_f32(A::to_f32(A::from_f32(x1 * c)) - A::to_f32(A::from_f32(x2 * sn)));
  const unsigned short o2 = A::from_f32(A::to_f32(A::from_f32(x2 * c)) + A::to_f32(A::from_f32(x1 * sn)));
  if (is_q) {
    unsigned short* o = a.q_out + ((int64_t)t * a.Hq + j) * a.hd;
    o[lane] = o1;
    o[lane + half] = o2;
  } else {
    const int hk = j - a.Hq;
    unsigned short* o = a.k_out + ((int64_t)t * a.Hkv + hk) * a.hd;
    o[lane] = o1;
    o[lane + half] = o2;
    unsigned short* kc = a.kcache + ((int64_t)hk * a.T_max + (a.pos0 + t)) * a.hd;
    kc[lane] = o1;
    kc[lane + half] = o2;
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void prompt_silu_mul_kernel(const unsigned short* __restrict__ gu, const float* __restrict__ rs,
                                                             unsigned short* __restrict__ out, int inter) {
  typedef Act<AT> A;
  const int t = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= inter) return;
  const float r = rs ? rs[t] : 1.0f;
  const u32x4 g = *(const u32x4*)(gu + (int64_t)t * 2 * inter + i);
  const u32x4 u = *(const u32x4*)(gu + (int64_t)t * 2 * inter + inter + i);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned short r2[2];
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      const unsigned short gb = (unsigned short)(hsel ? g[e] >> 16 : g[e] & 0xffffu), ub = (unsigned short)(hsel ? u[e] >> 16 : u[e] & 0xffffu);
      const float gv = A::to_f32(A::from_f32(A::to_f32(gb) * r)), uv = A::to_f32(A::from_f32(A::to_f32(ub) * r));
      const float sv = A::to_f32(A::from_f32(gv / (1.0f + __expf(-gv))));
      r2[hsel] = A::from_f32(sv * uv);
    }
    o[e] = (unsigned)r2[0] | ((unsigned)r2[1] << 16);
  }
  *(u32x4*)(out + (int64_t)t * inter + i) = o;
}

}  // namespace paro

extern "C" int paro_prompt_row_rms(const void* h, float* rs, int64_t rows, int64_t hidden, float eps, int act_dtype, void* stream) {
  using namespace paro;
  if (rows == 0) return PARO_OK;
  if (!h || !rs || rows < 0 || rows > 0x7fffffff || hidden < 8 || hidden % 8 != 0 || hidden > (1 << 24))
    return fail(PARO_ERR_INVALID, "paro_prompt_row_rms: bad arguments (hidden must be a multiple of 8)");
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  const dim3 grid((unsigned)rows);
  if (act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(prompt_row_rms_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)h, rs, (int)hidden, eps);
  else
    hipLaunchKernelGGL(prompt_row_rms_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)h, rs, (int)hidden, eps);
  return check_launch("paro_prompt_row_rms");
}

extern "C" int paro_prompt_qkv_post(const void* qkv, const float* rs, const void* q_norm_w, const void* k_norm_w, const float* rope,
                                    void* q_out, void* k_out, void* v_out, void* kcache, void* vcache, int64_t rows, int pos0,
                                    int n_heads, int n_kv_heads, int head_dim, int max_positions, float eps, int act_dtype, void* stream) {
  using namespace paro;
  if (rows == 0) return PARO_OK;
  if (!qkv || !rope || !q_out || !k_out || !v_out || !kcache || !vcache) return fail(PARO_ERR_INVALID, "paro_prompt_qkv_post: null pointer");
  if (rows < 0 || rows > 65535 || pos0 < 0 || pos0 + rows > max_positions) return fail(PARO_ERR_INVALID, "paro_prompt_qkv_post: rows / positions out of range");
  if (n_heads < 1 || n_kv_heads < 1 || head_dim < 2 || head_dim > 128 || head_dim % 2 != 0)
    return fail(PARO_ERR_UNSUPPORTED, "paro_prompt_qkv_post: head_dim must be even and <= 128 (got %d)", head_dim);
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  QkvPostArgs a;
  a.qkv = (const unsigned short*)qkv; a.rs = rs; a.qnw = (const unsigned short*)q_norm_w; a.knw = (const unsigned short*)k_norm_w; a.rope = rope;
  a.q_out = (unsigned short*)q_out; a.k_out = (unsigned short*)k_out; a.v_out = (unsigned short*)v_out;
  a.kcache = (unsigned short*)kcache; a.vcache = (unsigned short*)vcache;
  a.T = (int)rows; a.Hq = n_heads; a.Hkv = n_kv_heads; a.hd = head_dim; a.T_max = max_positions; a.pos0 = pos0; a.eps = eps;
  const dim3 grid((unsigned)((n_heads + 2 * n_kv_heads + 3) / 4), (unsigned)rows);
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(prompt_qkv_post_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(prompt_qkv_post_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("paro_prompt_qkv_post");
}

extern "C" int paro_prompt_silu_mul(const void* gate_up, const float* rs, void* out, int64_t rows, int64_t inter, int act_dtype, void* stream) {
  using namespace paro;
  if (rows == 0) return PARO_OK;
  if (!gate_up || !out || rows < 0 || rows > 65535 || inter < 8 || inter % 8 != 0 || inter > (1 << 24))
    return fail(PARO_ERR_INVALID, "paro_prompt_silu_mul: bad arguments (intermediate size must be a multiple of 8, rows <= 65535)");
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  const dim3 grid((unsigned)((inter / 8 + 255) / 256), (unsigned)rows);
  if (act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(prompt_silu_mul_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gate_up, rs, (unsigned short*)out, (int)inter);
  else
    hipLaunchKernelGGL(prompt_silu_mul_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gate_up, rs, (unsigned short*)out, (int)inter);
  return check_launch("paro_prompt_silu_mul");
}

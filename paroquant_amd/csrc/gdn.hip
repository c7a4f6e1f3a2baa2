// Qwen3.5 in the decode harness (BASELINE configs 3 and 5 name Qwen3.5-4B / -27B): what sits between the quantised
// linears of a hybrid decoder layer at batch 1 --
//   * gated delta net (transformers models/qwen3_5, Qwen3_5GatedDeltaNet.forward, the single-token cached path):
//       gdn_prep_kernel   causal_conv1d_update + SiLU over the in_proj_qkv outputs (state = the last three inputs per channel), and the
//                         two DENSE projections in_proj_a / in_proj_b (the reference's optimiser leaves them unquantised,
//                         experiments/optimize/4bit.sh:17-20) with the input RMSNorm applied as one scalar:
//                         decay = exp(-exp(A_log) * softplus(a + dt_bias)),  beta = sigmoid(b)
//       gdn_step_kernel   torch_recurrent_gated_delta_rule for one token: q / k l2-normalised, S *= decay, delta = (v - S^T k) beta,
//                         S += k delta^T, o = S^T q (state 128 x 128 fp32 per value head, in registers: 64 floats per thread), then
//                         Qwen3_5RMSNormGated with z
//   * gated full attention with head_dim 256 (Qwen3_5Attention): q_proj holds [query | gate] per head, q / k RMSNorm with (1 + w)
//     weights, PARTIAL rotary embedding (rotate_half on the first rotary_dim dimensions), KV-cache append, soft-max attention over
//     0..pos, output * sigmoid(gate): attn_gated_decode_kernel (one workgroup per query head; head_dim 64 / 128 models use attn.hip).
// The reference leaves all of this to HF generate() (transformers/generator.py:37-67); the parity target is HF's own modelling code.
#include "common.hpp"

namespace paro {

template <typename T>
using GPc = const __attribute__((address_space(1))) T*;

__device__ __forceinline__ float wave_total(float v) {   // the wave's sum in every lane
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

struct GdnPrepArgs {
  const unsigned short* qkv;      // [conv_dim] in_proj_qkv output of this token (activation dtype)
  const unsigned short* x;        // [hidden] the residual stream in front of the layer (un-normalised)
  const float* w_ab;              // [2 nv][hidden]: in_proj_a rows, then in_proj_b rows, the input norm's (1 + w) folded in
  unsigned short* conv_state;     // [conv_dim][4] activation dtype: the last three inputs in [1..3] ([0] unused), updated in place
  const float* conv_w;            // [conv_dim][4]
  const float* A_log;             // [nv]
  const float* dt_bias;           // [nv]
  unsigned short* conv_out;       // [conv_dim] silu(conv) in the activation dtype
  float* g_beta;                  // [2 nv]: decay = exp(g), then beta
  float eps;
  int hidden, conv_dim, nv, conv_blocks;
};

template <typename AT>
__global__ __launch_bounds__(256) void gdn_prep_kernel(const GdnPrepArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.conv_blocks) {
    const int c = blockIdx.x * 256 + tid;
    if (c >= a.conv_dim) return;
    const u32x2 st = *(const u32x2*)(a.conv_state + 4 * c);
    const f32x4 w = *(const f32x4*)(a.conv_w + 4 * c);
    const unsigned short xn = a.qkv[c];
    const float s0 = A::to_f32(st[0] >> 16), s1 = A::to_f32(st[1] & 0xffffu), s2 = A::to_f32(st[1] >> 16);
    // (HF: F.conv1d over [state | x] in the weight dtype, SiLU, cast back -- fp32 accumulation here, one rounding)
    const float v = w[0] * s0 + w[1] * s1 + w[2] * s2 + w[3] * A::to_f32(xn);
    a.conv_out[c] = A::from_f32(v * sigmoidf_(v));
    *(u32x2*)(a.conv_state + 4 * c) = (u32x2){(st[1] & 0xffffu) << 16, (st[1] >> 16) | ((unsigned)xn << 16)};
    return;
  }
  // one workgroup per row of [in_proj_a ; in_proj_b]: dot with the RMS-normalised hidden state (the norm's scalar applied last)
  const int r = blockIdx.x - a.conv_blocks;
  __shared__ float red[8];
  float dot = 0.f, ssq = 0.f;
  for (int i = tid; i < a.hidden; i += 256) {
    const float xv = A::to_f32(a.x[i]);
    dot = __builtin_fmaf(a.w_ab[(int64_t)r * a.hidden + i], xv, dot);
    ssq = __builtin_fmaf(xv, xv, ssq);
  }
  dot = wave_total(dot);
  ssq = wave_total(ssq);
  if ((tid & 63) == 0) { red[tid >> 6] = dot; red[4 + (tid >> 6)] = ssq; }
  __syncthreads();
  if (tid == 0) {
    const float d = red[0] + red[1] + red[2] + red[3], s = red[4] + red[5] + red[6] + red[7];
    const float val = d * __builtin_amdgcn_rsqf(s / (float)a.hidden + a.eps);
    if (r < a.nv) {
      const float t = val + a.dt_bias[r];
      const float sp = t > 20.f ? t : log1pf(__expf(t));                   // softplus
      a.g_beta[r] = __expf(-__expf(a.A_log[r]) * sp);
    } else {
      a.g_beta[r] = sigmoidf_(val);
    }
  }
}

struct GdnStepArgs {
  const unsigned short* conv_out;  // [2 key_dim + value_dim]: q heads, k heads, v heads (after conv + SiLU)
  const unsigned short* z;         // [value_dim] in_proj_z output
  const float* g_beta;             // [2 nv]
  float* state;                    // [nv][128 k][128 v] fp32, updated in place
  const unsigned short* norm_w;    // [128] weight of the gated RMSNorm
  unsigned short* out;             // [value_dim]
  float eps;
  int nk, nv;
};

template <typename AT>
__global__ __launch_bounds__(256) void gdn_step_kernel(const GdnStepArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x, v = tid & 127, kh2 = tid >> 7;
  const int h = blockIdx.x, kh = h / (a.nv / a.nk);
  const int key_dim = a.nk * 128;
  __shared__ float qs[128], ks[128], part[2][128], nrm[4];
  float* S = a.state + ((int64_t)h * 128 + kh2 * 64) * 128 + v;
  float s[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) s[i] = S[i * 128];                           // in flight under the q / k normalisation
  // q and k of this head's key head, l2-normalised (eps 1e-6), q scaled by 1 / sqrt(128)
  {
    const float x = A::to_f32(a.conv_out[(kh2 ? key_dim : 0) + kh * 128 + v]);
    const float ss = wave_total(x * x);
    if ((tid & 63) == 0) nrm[tid >> 6] = ss;
    __syncthreads();
    const float inv = __builtin_amdgcn_rsqf(nrm[2 * kh2] + nrm[2 * kh2 + 1] + 1e-6f);
    if (kh2) ks[v] = x * inv; else qs[v] = x * inv * 0.08838834764831845f;
  }
  __syncthreads();
  const float decay = a.g_beta[h], beta = a.g_beta[a.nv + h];
  float kv = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    s[i] *= decay;
    kv = __builtin_fmaf(s[i], ks[kh2 * 64 + i], kv);
  }
  part[kh2][v] = kv;
  __syncthreads();
  const float vv = A::to_f32(a.conv_out[2 * key_dim + h * 128 + v]);
  const float delta = (vv - (part[0][v] + part[1][v])) * beta;
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    s[i] = __builtin_fmaf(ks[kh2 * 64 + i], delta, s[i]);
    o = __builtin_fmaf(s[i], qs[kh2 * 64 + i], o);
    S[i * 128] = s[i];
  }
  part[kh2][v] = o;
  __syncthreads();
  if (kh2 == 0) {
    // Qwen3_5RMSNormGated on the head's 128 outputs (HF rounds the core output to the activation dtype first, normalises in fp32,
    // multiplies by the weight in the activation dtype, then by silu(z) in fp32)
    const float of = A::to_f32(A::from_f32(part[0][v] + part[1][v]));
    const float ss = wave_total(of * of);
    if ((tid & 63) == 0) nrm[tid >> 6] = ss;
  }
  __syncthreads();
  if (kh2 == 0) {
    const float of = A::to_f32(A::from_f32(part[0][v] + part[1][v]));
    const float n = A::to_f32(A::from_f32(of * __builtin_amdgcn_rsqf((nrm[0] + nrm[1]) / 128.f + a.eps)));
    const float wn = A::to_f32(A::from_f32(A::to_f32(a.norm_w[v]) * n));
    const float zf = A::to_f32(a.z[h * 128 + v]);
    a.out[h * 128 + v] = A::from_f32(wn * (zf * sigmoidf_(zf)));
  }
}

struct AttnGatedArgs {
  const unsigned short* qkv;   // [Hq][2][hd] (query | gate per head), then [Hkv][hd] keys, [Hkv][hd] values
  unsigned short* kcache;      // [Hkv][T_max][hd]
  unsigned short* vcache;      // [Hkv][T_max][hd]
  unsigned short* out;         // [Hq][hd]
  const int* pos;
  const float* rope;           // [T_max][rotary_dim]: cos[0 .. rd/2) then sin[0 .. rd/2)
  const unsigned short* qnw;   // [hd] norm weights w (applied as 1 + w when plus_one)
  const unsigned short* knw;
  float eps, scale;
  int Hq, Hkv, hd, rd, T_max, plus_one;
};

// one workgroup (256 threads = 256 head dims) per QUERY head; thread d owns dimension d of q, of the new k / v and of the output
template <typename AT>
__global__ __launch_bounds__(256) void attn_gated_decode_kernel(const AttnGatedArgs a) {
  typedef Act<AT> A;
  constexpr int HD = 256;
  extern __shared__ __attribute__((aligned(16))) float sc[];            // [T_max] scores / probabilities
  __shared__ float qv[HD], kn[HD], red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, n_rep = a.Hq / a.Hkv, kvh = head / n_rep;
  const int pos = *a.pos;
  const int half = a.rd / 2;
  auto block_sum = [&](float v, int slot) {
    v = wave_total(v);
    if (lane == 0) red[slot * 4 + wave] = v;
    __syncthreads();
    const float r = red[slot * 4] + red[slot * 4 + 1] + red[slot * 4 + 2] + red[slot * 4 + 3];
    __syncthreads();
    return r;
  };
  const float q0 = A::to_f32(a.qkv[(head * 2) * HD + tid]);
  const float gate = A::to_f32(a.qkv[(head * 2 + 1) * HD + tid]);
  const float k0 = A::to_f32(a.qkv[a.Hq * 2 * HD + kvh * HD + tid]);
  const unsigned short vnew = a.qkv[a.Hq * 2 * HD + a.Hkv * HD + kvh * HD + tid];
  const float wq = (a.plus_one ? 1.f : 0.f) + A::to_f32(a.qnw[tid]), wk = (a.plus_one ? 1.f : 0.f) + A::to_f32(a.knw[tid]);
  // RMSNorm over the head (fp32, rounded to the activation dtype like HF's `output.type_as(x)`), then the partial rotary embedding
  const float qn = A::to_f32(A::from_f32(q0 * __builtin_amdgcn_rsqf(block_sum(q0 * q0, 0) / HD + a.eps) * wq));
  const float kk = A::to_f32(A::from_f32(k0 * __builtin_amdgcn_rsqf(block_sum(k0 * k0, 1) / HD + a.eps) * wk));
  qv[tid] = qn;
  kn[tid] = kk;
  __syncthreads();
  float qr = qn, kr = kk;
  if (tid < a.rd) {
    const float c = a.rope[(int64_t)pos * a.rd + (tid % half)], s = a.rope[(int64_t)pos * a.rd + half + (tid % half)];
    const float qo = tid < half ? -qv[tid + half] : qv[tid - half];        // rotate_half
    const float ko = tid < half ? -kn[tid + half] : kn[tid - half];
    qr = A::to_f32(A::from_f32(qn * c + qo * s));
    kr = A::to_f32(A::from_f32(kk * c + ko * s));
  }
  __syncthreads();
  qv[tid] = qr * a.scale;
  kn[tid] = kr;
  if (head % n_rep == 0) {                                                    // one query head of the group appends to the cache
    a.kcache[((int64_t)kvh * a.T_max + pos) * HD + tid] = A::from_f32(kr);
    a.vcache[((int64_t)kvh * a.T_max + pos) * HD + tid] = vnew;
  }
  __syncthreads();
  // scores of the cached positions: a wave per position, four dimensions per lane
  const unsigned short* kc = a.kcache + (int64_t)kvh * a.T_max * HD;
  const float q4[4] = {qv[4 * lane], qv[4 * lane + 1], qv[4 * lane + 2], qv[4 * lane + 3]};
  for (int p = wave; p < pos; p += 4) {
    const u32x2 kw = *(const u32x2*)(kc + (int64_t)p * HD + 4 * lane);
    float d = q4[0] * A::to_f32(kw[0] & 0xffffu) + q4[1] * A::to_f32(kw[0] >> 16) + q4[2] * A::to_f32(kw[1] & 0xffffu) + q4[3] * A::to_f32(kw[1] >> 16);
    d = wave_total(d);
    if (lane == 0) sc[p] = d;
  }
  {   // the new position from registers / LDS (whoever appends may not have written it yet)
    const float d = block_sum(qv[tid] * kn[tid], 0);
    if (tid == 0) sc[pos] = d;
  }
  __syncthreads();
  float m = -3.0e38f;
  for (int p = tid; p <= pos; p += 256) m = fmaxf(m, sc[p]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int p = tid; p <= pos; p += 256) {
    const float e = __expf(sc[p] - m);
    sc[p] = e;
    l += e;
  }
  l = block_sum(l, 0);
  // P V: thread d accumulates dimension d over the positions, eight cache rows in flight
  const unsigned short* vc = a.vcache + (int64_t)kvh * a.T_max * HD + tid;
  float acc = 0.f;
  int p = 0;
  for (; p + 8 <= pos; p += 8) {
    unsigned short vr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vr[j] = vc[(int64_t)(p + j) * HD];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(sc[p + j], A::to_f32(vr[j]), acc);
  }
  for (; p < pos; ++p) acc = __builtin_fmaf(sc[p], A::to_f32(vc[(int64_t)p * HD]), acc);
  acc = __builtin_fmaf(sc[pos], A::to_f32(vnew), acc);
  const float o = A::to_f32(A::from_f32(acc / l));
  a.out[head * HD + tid] = A::from_f32(o * sigmoidf_(gate));
}

}  // namespace paro

extern "C" int paro_gdn_prep(const void* qkv, const void* x, const float* w_ab, float eps, void* conv_state, const float* conv_w,
                             const float* A_log, const float* dt_bias, void* conv_out, float* g_beta, int hidden, int conv_dim,
                             int n_v_heads, int act_dtype, void* stream) {
  using namespace paro;
  if (!qkv || !x || !w_ab || !conv_state || !conv_w || !A_log || !dt_bias || !conv_out || !g_beta) return fail(PARO_ERR_INVALID, "null pointer");
  if (hidden < 1 || conv_dim < 1 || n_v_heads < 1) return fail(PARO_ERR_INVALID, "bad geometry");
  GdnPrepArgs a;
  a.qkv = (const unsigned short*)qkv; a.x = (const unsigned short*)x; a.w_ab = w_ab; a.conv_state = (unsigned short*)conv_state; a.conv_w = conv_w;
  a.A_log = A_log; a.dt_bias = dt_bias; a.conv_out = (unsigned short*)conv_out; a.g_beta = g_beta; a.eps = eps;
  a.hidden = hidden; a.conv_dim = conv_dim; a.nv = n_v_heads; a.conv_blocks = (conv_dim + 255) / 256;
  const dim3 grid((unsigned)(a.conv_blocks + 2 * n_v_heads));
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(gdn_prep_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(gdn_prep_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_gdn_prep");
}

extern "C" int paro_gdn_step(const void* conv_out, const void* z, const float* g_beta, float* state, const void* norm_w, float eps, void* out,
                             int n_k_heads, int n_v_heads, int act_dtype, void* stream) {
  using namespace paro;
  if (!conv_out || !z || !g_beta || !state || !norm_w || !out) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_k_heads < 1 || n_v_heads < n_k_heads || n_v_heads % n_k_heads) return fail(PARO_ERR_INVALID, "value heads must be a multiple of key heads");
  GdnStepArgs a;
  a.conv_out = (const unsigned short*)conv_out; a.z = (const unsigned short*)z; a.g_beta = g_beta; a.state = state;
  a.norm_w = (const unsigned short*)norm_w; a.out = (unsigned short*)out; a.eps = eps; a.nk = n_k_heads; a.nv = n_v_heads;
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(gdn_step_kernel<f16>, dim3((unsigned)n_v_heads), dim3(256), 0, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(gdn_step_kernel<bf16>, dim3((unsigned)n_v_heads), dim3(256), 0, (hipStream_t)stream, a);
  else return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_gdn_step");
}

extern "C" int paro_attn_decode_gated(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                                      const void* q_norm_w, const void* k_norm_w, int norm_plus_one, float eps, float scale, int n_heads,
                                      int n_kv_heads, int head_dim, int rotary_dim, int max_positions, int act_dtype, void* stream) {
  using namespace paro;
  if (!qkv || !kcache || !vcache || !out || !pos || !rope || !q_norm_w || !k_norm_w) return fail(PARO_ERR_INVALID, "null pointer");
  if (head_dim != 256) return fail(PARO_ERR_UNSUPPORTED, "gated decode attention is built for head_dim 256 (64 / 128: paro_attn_decode)");
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads) return fail(PARO_ERR_INVALID, "query heads must be a multiple of KV heads");
  if (rotary_dim < 2 || rotary_dim > head_dim || rotary_dim % 2) return fail(PARO_ERR_INVALID, "rotary_dim must be even and <= head_dim");
  if (max_positions < 1 || max_positions > 12288) return fail(PARO_ERR_UNSUPPORTED, "gated decode attention keeps one score per position in LDS: up to 12288 positions");
  AttnGatedArgs a;
  a.qkv = (const unsigned short*)qkv; a.kcache = (unsigned short*)kcache; a.vcache = (unsigned short*)vcache; a.out = (unsigned short*)out;
  a.pos = pos; a.rope = rope; a.qnw = (const unsigned short*)q_norm_w; a.knw = (const unsigned short*)k_norm_w;
  a.eps = eps; a.scale = scale; a.Hq = n_heads; a.Hkv = n_kv_heads; a.hd = head_dim; a.rd = rotary_dim; a.T_max = max_positions;
  a.plus_one = norm_plus_one;
  const size_t lds = (size_t)max_positions * 4;
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(attn_gated_decode_kernel<f16>, dim3((unsigned)n_heads), dim3(256), lds, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(attn_gated_decode_kernel<bf16>, dim3((unsigned)n_heads), dim3(256), lds, (hipStream_t)stream, a);
  else return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_attn_decode_gated");
}

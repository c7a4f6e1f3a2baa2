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

// the wave's sum in every lane: four row steps and two row broadcasts on the VALU's DPP path, the total read back from lane 63 (a
// __shfl_xor butterfly is six dependent ds_bpermute round trips of ~100 cycles each; these kernels are chains of such reductions)
template <int CTRL, int RMASK>
__device__ __forceinline__ float gdn_dpp(float a) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, RMASK, 0xf, false));
}
__device__ __forceinline__ float wave_total(float v) {
  v += gdn_dpp<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += gdn_dpp<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += gdn_dpp<0x141, 0xf>(v);   // row_half_mirror
  v += gdn_dpp<0x140, 0xf>(v);   // row_mirror: every lane of a 16-lane row holds the row's sum
  v += gdn_dpp<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += gdn_dpp<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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
  // one workgroup per row of [in_proj_a ; in_proj_b]: dot with the RMS-normalised hidden state (the norm's scalar applied last).
  // Every load of the row is requested before anything is consumed (first build: a 16-step loop of dependent round trips, 9 us per launch)
  const int r = blockIdx.x - a.conv_blocks;
  __shared__ float red[8];
  float dot = 0.f, ssq = 0.f;
  const float* wr = a.w_ab + (int64_t)r * a.hidden;
  for (int i0 = 0; i0 < a.hidden; i0 += 256 * 16) {                       // 16 channels per thread and pass (hidden <= 4096: one pass)
    const int i = i0 + tid * 16;
    f32x4 w4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    u32x4 x8[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (i + 16 <= a.hidden) {
#pragma unroll
      for (int q = 0; q < 4; ++q) w4[q] = *(const f32x4*)(wr + i + 4 * q);
      x8[0] = *(const u32x4*)(a.x + i);
      x8[1] = *(const u32x4*)(a.x + i + 8);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned xw = x8[q >> 2][q & 3];
      const float x0 = A::to_f32(xw & 0xffffu), x1 = A::to_f32(xw >> 16);
      dot = __builtin_fmaf(w4[q >> 1][(q & 1) * 2], x0, __builtin_fmaf(w4[q >> 1][(q & 1) * 2 + 1], x1, dot));
      ssq = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, ssq));
    }
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
  float* scratch;                  // [nv][128] raw outputs of a head's four workgroups + [nv] arrival tickets (zero between launches)
  float eps;
  int nk, nv;
};

// grid = value heads x 4: a workgroup owns 32 of the head's 128 value columns (the recurrence is independent per column) -- a CU ingests
// ~13 B / clock, so the 128 KiB a head's state moves per token (read + write) are spread over four CUs (first build: one workgroup per
// head, 5.9 us).  Thread = (column v, one of eight 16-row slices of k).  The gated RMSNorm needs the head's 128 outputs: the LAST of the
// four workgroups to arrive (write-through stores, agent-scope release, ticket, acquire; no spinning) normalises and stores them.
template <typename AT>
__global__ __launch_bounds__(256) void gdn_step_kernel(const GdnStepArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x, vl = tid & 31, ksl = tid >> 5;
  const int h = blockIdx.x >> 2, c4 = blockIdx.x & 3, v = c4 * 32 + vl;
  const int kh = h / (a.nv / a.nk);
  const int key_dim = a.nk * 128;
  __shared__ float qs[128], ks[128], part[8][32], nrm[4];
  __shared__ unsigned last;
  float* S = a.state + ((int64_t)h * 128 + ksl * 16) * 128 + v;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = S[i * 128];                           // in flight under the q / k normalisation
  {
    const int d = tid & 127, isk = tid >> 7;
    const float x = A::to_f32(a.conv_out[(isk ? key_dim : 0) + kh * 128 + d]);
    const float ss = wave_total(x * x);
    if ((tid & 63) == 0) nrm[tid >> 6] = ss;
    __syncthreads();
    const float inv = __builtin_amdgcn_rsqf(nrm[2 * isk] + nrm[2 * isk + 1] + 1e-6f);
    if (isk) ks[d] = x * inv; else qs[d] = x * inv * 0.08838834764831845f;
  }
  __syncthreads();
  const float decay = a.g_beta[h], beta = a.g_beta[a.nv + h];
  float kv = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s[i] *= decay;
    kv = __builtin_fmaf(s[i], ks[ksl * 16 + i], kv);
  }
  part[ksl][vl] = kv;
  __syncthreads();
  float kvm = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) kvm += part[q][vl];
  const float vv = A::to_f32(a.conv_out[2 * key_dim + h * 128 + v]);
  const float delta = (vv - kvm) * beta;
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s[i] = __builtin_fmaf(ks[ksl * 16 + i], delta, s[i]);
    o = __builtin_fmaf(s[i], qs[ksl * 16 + i], o);
    S[i * 128] = s[i];
  }
  part[ksl][vl] = o;
  __syncthreads();
  float* raw = a.scratch + (int64_t)h * 128;
  unsigned* ticket = (unsigned*)(a.scratch + (int64_t)a.nv * 128) + h;
  if (tid < 32) {
    float of = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) of += part[q][tid];
    __hip_atomic_store(raw + c4 * 32 + tid, of, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
  }
  __syncthreads();
  if (tid == 0) {
    // The raw outputs went out as write-through (agent-scope) stores of this wave and come back as agent-scope loads: once they have
    // completed (vmcnt) the ticket may follow -- no release fence (it would write back the whole L2, the state stores of every
    // workgroup included: the first four-workgroup build was SLOWER than one workgroup per head, 7.8 vs 5.9 us) and no acquire.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last || tid >= 128) return;
  // Qwen3_5RMSNormGated on the head's 128 outputs (HF rounds the core output to the activation dtype first, normalises in fp32,
  // multiplies by the weight in the activation dtype, then by silu(z) in fp32)
  const float of = A::to_f32(A::from_f32(__hip_atomic_load(raw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  const float ss = wave_total(of * of);
  if ((tid & 63) == 0) nrm[tid >> 6] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (only waves 0 and 1 are here)
  const float n = A::to_f32(A::from_f32(of * __builtin_amdgcn_rsqf((nrm[0] + nrm[1]) / 128.f + a.eps)));
  const float wn = A::to_f32(A::from_f32(A::to_f32(a.norm_w[tid]) * n));
  const float zf = A::to_f32(a.z[h * 128 + tid]);
  a.out[h * 128 + tid] = A::from_f32(wn * (zf * sigmoidf_(zf)));
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
  // *pos lives in device memory (a captured graph replays for every token): no host check can see it.  A position outside the cache
  // writes nothing to the KV cache or the score array and answers NaN (as paro_attn_decode's kernels leave the cache alone)
  if (pos < 0 || pos >= a.T_max) {
    a.out[head * HD + tid] = A::from_f32(__builtin_nanf(""));
    return;
  }
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
  // scores of the cached positions: sixteen positions per wave and step, FOUR lanes per position (64 dimensions = eight 16-byte loads
  // each), the four partial dots combined on the DPP path (first build: a wave per position and six ds_bpermute round trips per score,
  // 25 us per launch at ~200 positions)
  const unsigned short* kc = a.kcache + (int64_t)kvh * a.T_max * HD;
  {
    const int sub = lane & 3, pg = lane >> 2;
    float qd[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) qd[i] = qv[sub * 64 + i];
    for (int base = wave * 16; base < pos; base += 64) {
      const int p = base + pg;
      const int pc = p < pos ? p : 0;                                       // (clamped: the load count stays static)
      u32x4 kw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) kw[j] = *(const u32x4*)(kc + (int64_t)pc * HD + sub * 64 + 8 * j);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d = __builtin_fmaf(qd[8 * j + 2 * e], A::to_f32(kw[j][e] & 0xffffu), d);
          d = __builtin_fmaf(qd[8 * j + 2 * e + 1], A::to_f32(kw[j][e] >> 16), d);
        }
      d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
      d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
      if (sub == 0 && p < pos) sc[p] = d;
    }
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
  // P V: wave w takes the positions w, w + 4, ...; a lane owns four dimensions (one coalesced 512-byte row per wave and load, eight rows
  // in flight); the four waves' partial outputs meet in LDS
  const unsigned short* vc = a.vcache + (int64_t)kvh * a.T_max * HD + 4 * lane;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p0 = wave; p0 < pos; p0 += 32) {
    u32x2 vr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + 4 * j;
      vr[j] = *(const u32x2*)(vc + (int64_t)(p < pos ? p : 0) * HD);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + 4 * j;
      const float w = p < pos ? sc[p] : 0.f;
      acc[0] = __builtin_fmaf(w, A::to_f32(vr[j][0] & 0xffffu), acc[0]);
      acc[1] = __builtin_fmaf(w, A::to_f32(vr[j][0] >> 16), acc[1]);
      acc[2] = __builtin_fmaf(w, A::to_f32(vr[j][1] & 0xffffu), acc[2]);
      acc[3] = __builtin_fmaf(w, A::to_f32(vr[j][1] >> 16), acc[3]);
    }
  }
  __syncthreads();                                                          // (qv / kn are free now: the partial outputs take their place)
  float* pa = (wave & 1) ? kn : qv;                                          // waves 0 / 1 first, then 2 / 3 add
  if (wave < 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pa[4 * lane + e] = acc[e];
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pa[4 * lane + e] += acc[e];
  }
  __syncthreads();
  const float o = A::to_f32(A::from_f32((qv[tid] + kn[tid] + sc[pos] * A::to_f32(vnew)) / l));
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

namespace paro {
// Prefill of the gated delta net: the recurrence of gdn_step_kernel over T tokens in ONE launch -- the 128 x 128 state of a value head
// stays in registers from the first token to the last (read once, written once), which is what a recurrent layer's prompt pass should cost;
// the token-by-token route moved it through HBM per token (and was 2 launches per token and layer).  Same work split (value heads x 4
// workgroups of 32 columns), same arithmetic per token, token t + 1's inputs requested while token t is computed.  The gated RMSNorm
// needs a head's 128 outputs, i.e. all four workgroups: it is left to the caller (raw fp32 outputs [T][nv * 128]), as is the causal
// convolution in front (both are row-parallel over T: library GEMM / elementwise work, not a recurrence).
struct GdnSeqArgs {
  const unsigned short* conv_out;  // [T][2 key_dim + value_dim] after conv + SiLU
  const float* g_beta;             // [T][2 nv]: decay, beta
  float* state;                    // [nv][128][128], updated in place
  float* out_raw;                  // [T][nv * 128] fp32: o = S^T q per token (before the gated norm)
  int nk, nv, T;
};
template <typename AT>
__global__ __launch_bounds__(256) void gdn_seq_kernel(const GdnSeqArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x, vl = tid & 31, ksl = tid >> 5;
  const int h = blockIdx.x >> 2, c4 = blockIdx.x & 3, v = c4 * 32 + vl;
  const int kh = h / (a.nv / a.nk);
  const int key_dim = a.nk * 128, row = 2 * key_dim + a.nv * 128;
  __shared__ float qs[128], ks[128], part[8][32], nrm[4];
  float* S = a.state + ((int64_t)h * 128 + ksl * 16) * 128 + v;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = S[i * 128];
  const int d = tid & 127, isk = tid >> 7;
  auto fetch = [&](int t, unsigned short& xr, unsigned short& vr, float& dec, float& bet) {
    const unsigned short* r = a.conv_out + (int64_t)t * row;
    xr = r[(isk ? key_dim : 0) + kh * 128 + d];
    vr = r[2 * key_dim + h * 128 + v];
    dec = a.g_beta[(int64_t)t * 2 * a.nv + h];
    bet = a.g_beta[(int64_t)t * 2 * a.nv + a.nv + h];
  };
  unsigned short xr, vr;
  float decay, beta;
  fetch(0, xr, vr, decay, beta);
  for (int t = 0; t < a.T; ++t) {
    unsigned short xn = 0, vn = 0;
    float dn = 0.f, bn = 0.f;
    fetch(min(t + 1, a.T - 1), xn, vn, dn, bn);                      // the next token's inputs are in flight under this one's arithmetic
    {
      const float x = A::to_f32(xr);
      const float ss = wave_total(x * x);
      if ((tid & 63) == 0) nrm[tid >> 6] = ss;
      __syncthreads();
      const float inv = __builtin_amdgcn_rsqf(nrm[2 * isk] + nrm[2 * isk + 1] + 1e-6f);
      if (isk) ks[d] = x * inv; else qs[d] = x * inv * 0.08838834764831845f;
    }
    __syncthreads();
    float kv = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] *= decay;
      kv = __builtin_fmaf(s[i], ks[ksl * 16 + i], kv);
    }
    part[ksl][vl] = kv;
    __syncthreads();
    float kvm = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) kvm += part[q][vl];
    const float delta = (A::to_f32(vr) - kvm) * beta;
    __syncthreads();
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_fmaf(ks[ksl * 16 + i], delta, s[i]);
      o = __builtin_fmaf(s[i], qs[ksl * 16 + i], o);
    }
    part[ksl][vl] = o;
    __syncthreads();
    if (tid < 32) {
      float of = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) of += part[q][tid];
      a.out_raw[(int64_t)t * a.nv * 128 + h * 128 + c4 * 32 + tid] = of;
    }
    // (no barrier here: nrm / qs / ks / part are next written behind the next token's first, second and third barriers)
    xr = xn; vr = vn; decay = dn; beta = bn;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) S[i * 128] = s[i];
}
}  // namespace paro

extern "C" int paro_gdn_sequence(const void* conv_out, const float* g_beta, float* state, float* out_raw, int n_tokens, int n_k_heads,
                                 int n_v_heads, int act_dtype, void* stream) {
  using namespace paro;
  if (!conv_out || !g_beta || !state || !out_raw) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_tokens < 1) return fail(PARO_ERR_INVALID, "n_tokens must be >= 1");
  if (n_k_heads < 1 || n_v_heads < n_k_heads || n_v_heads % n_k_heads) return fail(PARO_ERR_INVALID, "value heads must be a multiple of key heads");
  GdnSeqArgs a;
  a.conv_out = (const unsigned short*)conv_out; a.g_beta = g_beta; a.state = state; a.out_raw = out_raw;
  a.nk = n_k_heads; a.nv = n_v_heads; a.T = n_tokens;
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(gdn_seq_kernel<f16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(gdn_seq_kernel<bf16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
  else return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_gdn_sequence");
}

namespace paro {
// gdn_prep_kernel folded into gdn_step_kernel: ONE launch per gated-delta-net block and token instead of two (a dependent launch costs
// ~2.4 us before it does anything; the block's two were 4.7 + 5.3 us).  Every workgroup (value head h, column quarter c4) computes what it
// needs itself, with the arithmetic of the two kernels operation for operation (same bits):
//   * the causal convolution of ITS channels -- q and k of the key head (one channel per thread), its 32 value columns -- and their next
//     state.  The workgroups that share a key head compute the same q / k channels: the convolution state is double-buffered by the
//     token's parity (read [pos & 1], write [(pos + 1) & 1]), so nobody reads what another workgroup of the launch has written, and the
//     duplicate writers store identical values;
//   * the head's two dense rows (in_proj_a, in_proj_b) with the input RMSNorm's scalar: 2 x hidden fp32 weights per workgroup, four times
//     per head (4 MB instead of 1 MB per layer on Qwen3.5-9B: L2-served after the first reader).
struct GdnFusedArgs {
  const unsigned short* qkvz;      // [conv_dim + value_dim]: the convolution's inputs, then z
  const unsigned short* x;         // [hidden] residual stream in front of the layer
  const float* w_ab;               // [2 nv][hidden]
  unsigned short* conv_state;      // [2][conv_dim][4]
  const float* conv_w;             // [conv_dim][4]
  const float* A_log;
  const float* dt_bias;
  float* state;                    // [nv][128][128]
  const unsigned short* norm_w;
  unsigned short* out;             // [value_dim]
  float* scratch;                  // paro_gdn_workspace_bytes
  const int* pos;
  float eps_in, eps;
  int hidden, conv_dim, nk, nv;
};
template <typename AT>
__global__ __launch_bounds__(256) void gdn_fused_kernel(const GdnFusedArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x, vl = tid & 31, ksl = tid >> 5;
  const int h = blockIdx.x >> 2, c4 = blockIdx.x & 3, v = c4 * 32 + vl;
  const int kh = h / (a.nv / a.nk);
  const int key_dim = a.nk * 128;
  __shared__ float qs[128], ks[128], part[8][32], nrm[4], red[16], vsh[32], gb[2];
  __shared__ unsigned last;
  int pos;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pos) : "s"(a.pos) : "memory");
  const unsigned short* cs_rd = a.conv_state + (int64_t)(pos & 1) * a.conv_dim * 4;
  unsigned short* cs_wr = a.conv_state + (int64_t)((pos + 1) & 1) * a.conv_dim * 4;
  float* S = a.state + ((int64_t)h * 128 + ksl * 16) * 128 + v;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = S[i * 128];                           // in flight under everything in front of the recurrence
  // ---- the convolution of this workgroup's channels (gdn_prep_kernel's expression): thread -> one q or k channel; threads 0..31 also a v column
  const int d = tid & 127, isk = tid >> 7;
  auto conv = [&](int c) -> float {
    const u32x2 st = *(const u32x2*)(cs_rd + 4 * c);
    const f32x4 w = *(const f32x4*)(a.conv_w + 4 * c);
    const unsigned short xn = a.qkvz[c];
    const float s0 = A::to_f32(st[0] >> 16), s1 = A::to_f32(st[1] & 0xffffu), s2 = A::to_f32(st[1] >> 16);
    const float cv = w[0] * s0 + w[1] * s1 + w[2] * s2 + w[3] * A::to_f32(xn);
    *(u32x2*)(cs_wr + 4 * c) = (u32x2){(st[1] & 0xffffu) << 16, (st[1] >> 16) | ((unsigned)xn << 16)};
    return A::to_f32(A::from_f32(cv * sigmoidf_(cv)));
  };
  const float xqk = conv((isk ? key_dim : 0) + kh * 128 + d);
  if (tid < 32) vsh[tid] = conv(2 * key_dim + h * 128 + c4 * 32 + tid);
  // ---- the head's dense rows a, b over the RMS-normalised hidden state (gdn_prep_kernel's row workgroups, two rows here)
  float dota = 0.f, dotb = 0.f, ssq = 0.f;
  {
    const float* wa = a.w_ab + (int64_t)h * a.hidden;
    const float* wb = a.w_ab + (int64_t)(a.nv + h) * a.hidden;
    for (int i0 = 0; i0 < a.hidden; i0 += 256 * 16) {
      const int i = i0 + tid * 16;
      f32x4 wa4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      f32x4 wb4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      u32x4 x8[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
      if (i + 16 <= a.hidden) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { wa4[q] = *(const f32x4*)(wa + i + 4 * q); wb4[q] = *(const f32x4*)(wb + i + 4 * q); }
        x8[0] = *(const u32x4*)(a.x + i);
        x8[1] = *(const u32x4*)(a.x + i + 8);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned xw = x8[q >> 2][q & 3];
        const float x0 = A::to_f32(xw & 0xffffu), x1 = A::to_f32(xw >> 16);
        dota = __builtin_fmaf(wa4[q >> 1][(q & 1) * 2], x0, __builtin_fmaf(wa4[q >> 1][(q & 1) * 2 + 1], x1, dota));
        dotb = __builtin_fmaf(wb4[q >> 1][(q & 1) * 2], x0, __builtin_fmaf(wb4[q >> 1][(q & 1) * 2 + 1], x1, dotb));
        ssq = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, ssq));
      }
    }
  }
  dota = wave_total(dota);
  dotb = wave_total(dotb);
  ssq = wave_total(ssq);
  {
    // q / k l2-norm statistics (gdn_step_kernel) ride on the same barrier
    const float ss = wave_total(xqk * xqk);
    if ((tid & 63) == 0) { red[tid >> 6] = dota; red[4 + (tid >> 6)] = dotb; red[8 + (tid >> 6)] = ssq; nrm[tid >> 6] = ss; }
  }
  __syncthreads();
  if (tid == 0) {
    const float sa = red[0] + red[1] + red[2] + red[3], sb = red[4] + red[5] + red[6] + red[7], sq = red[8] + red[9] + red[10] + red[11];
    const float rstd = __builtin_amdgcn_rsqf(sq / (float)a.hidden + a.eps_in);
    const float t = sa * rstd + a.dt_bias[h];
    const float sp = t > 20.f ? t : log1pf(__expf(t));
    gb[0] = __expf(-__expf(a.A_log[h]) * sp);
    gb[1] = sigmoidf_(sb * rstd);
  }
  {
    const float inv = __builtin_amdgcn_rsqf(nrm[2 * isk] + nrm[2 * isk + 1] + 1e-6f);
    if (isk) ks[d] = xqk * inv; else qs[d] = xqk * inv * 0.08838834764831845f;
  }
  __syncthreads();
  const float decay = gb[0], beta = gb[1];
  float kv = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s[i] *= decay;
    kv = __builtin_fmaf(s[i], ks[ksl * 16 + i], kv);
  }
  part[ksl][vl] = kv;
  __syncthreads();
  float kvm = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) kvm += part[q][vl];
  const float delta = (vsh[vl] - kvm) * beta;
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s[i] = __builtin_fmaf(ks[ksl * 16 + i], delta, s[i]);
    o = __builtin_fmaf(s[i], qs[ksl * 16 + i], o);
    S[i * 128] = s[i];
  }
  part[ksl][vl] = o;
  __syncthreads();
  float* raw = a.scratch + (int64_t)h * 128;
  unsigned* ticket = (unsigned*)(a.scratch + (int64_t)a.nv * 128) + h;
  if (tid < 32) {
    float of = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) of += part[q][tid];
    __hip_atomic_store(raw + c4 * 32 + tid, of, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last || tid >= 128) return;
  const float of = A::to_f32(A::from_f32(__hip_atomic_load(raw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  const float ss = wave_total(of * of);
  if ((tid & 63) == 0) nrm[tid >> 6] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const float n = A::to_f32(A::from_f32(of * __builtin_amdgcn_rsqf((nrm[0] + nrm[1]) / 128.f + a.eps)));
  const float wn = A::to_f32(A::from_f32(A::to_f32(a.norm_w[tid]) * n));
  const float zf = A::to_f32(a.qkvz[a.conv_dim + h * 128 + tid]);
  a.out[h * 128 + tid] = A::from_f32(wn * (zf * sigmoidf_(zf)));
}
}  // namespace paro

extern "C" int paro_gdn_fused_step(const void* qkvz, const void* x, const float* w_ab, float eps_in, void* conv_state, const float* conv_w,
                                   const float* A_log, const float* dt_bias, float* state, const void* norm_w, float eps, void* out,
                                   const int32_t* pos, int hidden, int conv_dim, int n_k_heads, int n_v_heads, int act_dtype, void* workspace,
                                   void* stream) {
  using namespace paro;
  if (!qkvz || !x || !w_ab || !conv_state || !conv_w || !A_log || !dt_bias || !state || !norm_w || !out || !pos || !workspace)
    return fail(PARO_ERR_INVALID, "null pointer");
  if (hidden < 16 || hidden % 16 || conv_dim < 1) return fail(PARO_ERR_INVALID, "bad geometry (hidden must be a multiple of 16)");
  if (n_k_heads < 1 || n_v_heads < n_k_heads || n_v_heads % n_k_heads) return fail(PARO_ERR_INVALID, "value heads must be a multiple of key heads");
  if (conv_dim != 2 * n_k_heads * 128 + n_v_heads * 128) return fail(PARO_ERR_INVALID, "conv_dim must be (2 key heads + value heads) x 128");
  GdnFusedArgs a;
  a.qkvz = (const unsigned short*)qkvz; a.x = (const unsigned short*)x; a.w_ab = w_ab; a.conv_state = (unsigned short*)conv_state; a.conv_w = conv_w;
  a.A_log = A_log; a.dt_bias = dt_bias; a.state = state; a.norm_w = (const unsigned short*)norm_w; a.out = (unsigned short*)out;
  a.scratch = (float*)workspace; a.pos = pos; a.eps_in = eps_in; a.eps = eps; a.hidden = hidden; a.conv_dim = conv_dim; a.nk = n_k_heads; a.nv = n_v_heads;
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(gdn_fused_kernel<f16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(gdn_fused_kernel<bf16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
  else return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_gdn_fused_step");
}

extern "C" int64_t paro_gdn_workspace_bytes(int n_v_heads) { return n_v_heads < 1 ? -1 : (int64_t)n_v_heads * (128 + 1) * 4; }

extern "C" int paro_gdn_step(const void* conv_out, const void* z, const float* g_beta, float* state, const void* norm_w, float eps, void* out,
                             int n_k_heads, int n_v_heads, int act_dtype, void* workspace, void* stream) {
  using namespace paro;
  if (!conv_out || !z || !g_beta || !state || !norm_w || !out || !workspace) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_k_heads < 1 || n_v_heads < n_k_heads || n_v_heads % n_k_heads) return fail(PARO_ERR_INVALID, "value heads must be a multiple of key heads");
  GdnStepArgs a;
  a.conv_out = (const unsigned short*)conv_out; a.z = (const unsigned short*)z; a.g_beta = g_beta; a.state = state;
  a.norm_w = (const unsigned short*)norm_w; a.out = (unsigned short*)out; a.eps = eps; a.nk = n_k_heads; a.nv = n_v_heads;
  a.scratch = (float*)workspace;
  if (act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL(gdn_step_kernel<f16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
  else if (act_dtype == PARO_DTYPE_BF16) hipLaunchKernelGGL(gdn_step_kernel<bf16>, dim3((unsigned)n_v_heads * 4), dim3(256), 0, (hipStream_t)stream, a);
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

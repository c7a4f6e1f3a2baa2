// Batch-1 decode attention for the decode harness: everything between the merged qkv projection
// and o_proj of one decoder layer in ONE launch -- per-head q/k RMSNorm (Qwen3), rotary embedding, KV-cache append,
// grouped-query attention over positions 0..pos, softmax, P V.  The reference leaves this to HF generate() / vLLM;
// here it exists so that a whole decode step is five launches per layer (qkv GEMV, this, o GEMV, gate_up GEMV,
// down GEMV) inside one HIP graph, and the end-to-end tokens/s of north_star can be measured.
//
// One 256-thread workgroup per KV head (its n_rep = Hq / Hkv query heads share every K / V byte it reads).
// The position comes from DEVICE memory (`pos`), so a captured graph replays for every token.
#include "common.hpp"

namespace paro {

struct AttnArgs {
  const unsigned short* qkv;   // [(Hq + 2 Hkv) * hd]: q heads, k heads, v heads of this token
  unsigned short* kcache;      // [Hkv][T_max][hd]
  unsigned short* vcache;
  unsigned short* out;         // [Hq * hd]
  const int* pos;              // device scalar: 0-based position of this token
  const float* rope;           // [T_max][hd]: cos[0 .. hd/2) then sin[0 .. hd/2) of every position
  const unsigned short* qnw;   // [hd] q-norm weight or null
  const unsigned short* knw;   // [hd] k-norm weight or null
  float eps, scale;
  int Hq, Hkv, hd, T_max;
};

template <typename AT>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs a) {
  typedef Act<AT> A;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x;
  const int hd = a.hd, half = hd >> 1;
  const int n_rep = a.Hq / a.Hkv;
  const int pos = *a.pos;
  const int T = pos + 1;
  // LDS: q [n_rep][hd] f32 | knew [hd] f32 | vnew [hd] f32 | red [n_rep][8] f32 | acc [parts][n_rep][hd] f32 | sc [n_rep][T] f32
  float* qs = (float*)smem;
  float* knew = qs + n_rep * hd;
  float* vnew = knew + hd;
  float* red = vnew + hd;
  const int parts = 256 / hd;
  float* accs = red + n_rep * 8;
  float* sc = accs + parts * n_rep * hd;

  // ---- step 1: per-head RMSNorm (optional) + rotary embedding of the n_rep query heads and the new key
  const float* rp = a.rope + (int64_t)pos * hd;
  for (int v = wave; v <= n_rep; v += 4) {          // vector v < n_rep: query head, v == n_rep: the key
    const bool isk = v == n_rep;
    const unsigned short* src = isk ? a.qkv + (int64_t)a.Hq * hd + (int64_t)h * hd : a.qkv + ((int64_t)h * n_rep + v) * hd;
    const unsigned short* nw = isk ? a.knw : a.qnw;
    const bool act = lane < half;
    float x0 = act ? A::to_f32(src[lane]) : 0.f, x1 = act ? A::to_f32(src[lane + half]) : 0.f;
    if (nw) {
      float ss = x0 * x0 + x1 * x1;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
      const float r = __builtin_amdgcn_rsqf(ss / (float)hd + a.eps);
      // HF: normalise in fp32, round to the activation dtype, then multiply by the weight
      if (act) {
        x0 = A::to_f32(A::from_f32(x0 * r)) * A::to_f32(nw[lane]);
        x1 = A::to_f32(A::from_f32(x1 * r)) * A::to_f32(nw[lane + half]);
        x0 = A::to_f32(A::from_f32(x0));
        x1 = A::to_f32(A::from_f32(x1));
      }
    }
    if (act) {
      // rotate_half convention, cos / sin rounded to the activation dtype like HF's rotary embedding does
      const float c = A::to_f32(A::from_f32(rp[lane])), s = A::to_f32(A::from_f32(rp[half + lane]));
      const float y0 = A::to_f32(A::from_f32(x0 * c - x1 * s)), y1 = A::to_f32(A::from_f32(x1 * c + x0 * s));
      if (isk) {
        knew[lane] = y0;
        knew[lane + half] = y1;
        unsigned short* kc = a.kcache + ((int64_t)h * a.T_max + pos) * hd;
        kc[lane] = A::from_f32(y0);
        kc[lane + half] = A::from_f32(y1);
      } else {
        qs[v * hd + lane] = y0 * a.scale;
        qs[v * hd + lane + half] = y1 * a.scale;
      }
    }
  }
  if (tid < hd) {
    const unsigned short vv = a.qkv[(int64_t)(a.Hq + a.Hkv) * hd + (int64_t)h * hd + tid];
    vnew[tid] = A::to_f32(vv);
    a.vcache[((int64_t)h * a.T_max + pos) * hd + tid] = vv;
  }
  __syncthreads();

  // ---- step 2: scores s[j][p] = q_j . K[p]; one position per thread and pass
  const unsigned short* kbase = a.kcache + (int64_t)h * a.T_max * hd;
  for (int p = tid; p < T; p += 256) {
    float dot[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dot[j] = 0.f;
    if (p == pos) {
      for (int d = 0; d < hd; ++d) {
        const float kv = knew[d];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < n_rep) dot[j] = __builtin_fmaf(qs[j * hd + d], kv, dot[j]);
      }
    } else {
      const u32x4* kr = (const u32x4*)(kbase + (int64_t)p * hd);
      for (int c = 0; c < hd / 8; ++c) {
        const u32x4 w = kr[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float k0 = A::to_f32(w[e] & 0xffffu), k1 = A::to_f32(w[e] >> 16);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < n_rep) {
              dot[j] = __builtin_fmaf(qs[j * hd + c * 8 + 2 * e], k0, dot[j]);
              dot[j] = __builtin_fmaf(qs[j * hd + c * 8 + 2 * e + 1], k1, dot[j]);
            }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < n_rep) sc[(int64_t)j * T + p] = dot[j];
  }
  __syncthreads();

  // ---- step 3: softmax over p for every query head (max, exp, sum through LDS)
  for (int j = 0; j < n_rep; ++j) {
    float m = -3.0e38f;
    for (int p = tid; p < T; p += 256) m = fmaxf(m, sc[(int64_t)j * T + p]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[j * 8 + wave] = m;
  }
  __syncthreads();
  for (int j = 0; j < n_rep; ++j) {
    const float m = fmaxf(fmaxf(red[j * 8 + 0], red[j * 8 + 1]), fmaxf(red[j * 8 + 2], red[j * 8 + 3]));
    float l = 0.f;
    for (int p = tid; p < T; p += 256) {
      const float e = __builtin_amdgcn_exp2f((sc[(int64_t)j * T + p] - m) * 1.4426950408889634f);
      sc[(int64_t)j * T + p] = e;
      l += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off, 64);
    if (lane == 0) red[j * 8 + 4 + wave] = l;
  }
  __syncthreads();

  // ---- step 4: o_j[d] = sum_p P_j[p] V[p][d] / l_j; thread = (part, d), parts interleave the positions
  {
    const int d = tid % hd, part = tid / hd;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    const unsigned short* vbase = a.vcache + (int64_t)h * a.T_max * hd;
    if (part < parts) {
      for (int p = part; p < T; p += parts) {
        const float vv = (p == pos) ? vnew[d] : A::to_f32(vbase[(int64_t)p * hd + d]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < n_rep) o[j] = __builtin_fmaf(sc[(int64_t)j * T + p], vv, o[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < n_rep) accs[(part * n_rep + j) * hd + d] = o[j];
    }
  }
  __syncthreads();
  for (int e = tid; e < n_rep * hd; e += 256) {
    const int j = e / hd, d = e % hd;
    float v = 0.f;
    for (int part = 0; part < parts; ++part) v += accs[(part * n_rep + j) * hd + d];
    const float l = red[j * 8 + 4] + red[j * 8 + 5] + red[j * 8 + 6] + red[j * 8 + 7];
    a.out[((int64_t)h * n_rep + j) * hd + d] = A::from_f32(v / l);
  }
}

}  // namespace paro

extern "C" int64_t paro_attn_decode_lds_bytes(int n_heads, int n_kv_heads, int head_dim, int max_positions) {
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || head_dim < 2) return -1;
  const int64_t n_rep = n_heads / n_kv_heads, parts = 256 / head_dim;
  return 4 * (n_rep * head_dim + 2 * head_dim + n_rep * 8 + parts * n_rep * head_dim + n_rep * (int64_t)max_positions);
}

extern "C" int paro_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                                const void* q_norm_w, const void* k_norm_w, float eps, float scale, int n_heads,
                                int n_kv_heads, int head_dim, int max_positions, int act_dtype, void* stream) {
  using namespace paro;
  if (!qkv || !kcache || !vcache || !out || !pos || !rope) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0) return fail(PARO_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (n_heads / n_kv_heads > 8) return fail(PARO_ERR_UNSUPPORTED, "at most 8 query heads per KV head (got %d)", n_heads / n_kv_heads);
  if (head_dim != 64 && head_dim != 128) return fail(PARO_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return fail(PARO_ERR_INVALID, "q / k norm weights come together");
  const int64_t lds = paro_attn_decode_lds_bytes(n_heads, n_kv_heads, head_dim, max_positions);
  if (max_positions < 1 || lds > 160 * 1024)
    return fail(PARO_ERR_UNSUPPORTED, "max_positions %d needs %lld bytes of LDS for the scores (limit 163840)", max_positions, (long long)lds);
  AttnArgs a;
  a.qkv = (const unsigned short*)qkv;
  a.kcache = (unsigned short*)kcache;
  a.vcache = (unsigned short*)vcache;
  a.out = (unsigned short*)out;
  a.pos = pos;
  a.rope = rope;
  a.qnw = (const unsigned short*)q_norm_w;
  a.knw = (const unsigned short*)k_norm_w;
  a.eps = eps;
  a.scale = scale;
  a.Hq = n_heads;
  a.Hkv = n_kv_heads;
  a.hd = head_dim;
  a.T_max = max_positions;
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == PARO_DTYPE_F16) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_decode_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_decode_kernel<f16>, dim3((unsigned)n_kv_heads), dim3(256), (size_t)lds, st, a);
  } else if (act_dtype == PARO_DTYPE_BF16) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_decode_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_decode_kernel<bf16>, dim3((unsigned)n_kv_heads), dim3(256), (size_t)lds, st, a);
  } else {
    return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  }
  return check_launch("paro_attn_decode");
}

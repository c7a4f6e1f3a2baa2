// Final RMSNorm + unquantised lm_head GEMV + greedy argmax for the decode harness.  The lm_head
// of a *-PARO checkpoint stays fp16 (only the decoder linears are quantised, cli/convert.py:408-464), so at batch 1
// it is a plain HBM-bound matrix-vector product (Qwen3-4B: 778 MB per token -- a third of the quantised weights):
//   logits[v] = sum_k W[v][k] * xn[k],   xn = fp16(fp16(x * rsqrt(mean(x^2) + eps)) * w_norm)   (HF RMSNorm rounding)
// One wave per 4 rows at a time, 16-byte non-temporal loads, fp32 accumulation; every workgroup also leaves its
// (max logit, lowest index) pair, and a one-workgroup second kernel reduces those, appends the consumed token to the
// output sequence and advances the position -- the three torch launches (index_copy, argmax, add) it replaces.
#include <type_traits>

#include "common.hpp"

namespace paro {

// acc + w.lo * x.lo + w.hi * x.hi on two packed activations (v_dot2c_f32_f16 / v_dot2c_f32_bf16: fp32 accumulation): the row products
// were a convert + an FMA per element -- ~50 us of VALU issue per Qwen3.5 token beside 208 us of streaming at one or two waves per SIMD
template <typename AT>
__device__ __forceinline__ float dot2_acc(unsigned w, unsigned x, float acc) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  if constexpr (std::is_same<AT, f16>::value) return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), __builtin_bit_cast(h2, x), acc, false);
  else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w), __builtin_bit_cast(b2, x), acc, false);
}

constexpr int kLmRowsPerWg = 64;   // 4 waves x 4 passes x 4 rows

struct LmHeadArgs {
  const unsigned short* x;       // [H]
  const unsigned short* nw;      // [H] final norm weight
  const unsigned short* W;       // [V][H]
  unsigned short* logits;        // [V]
  float* pmax;                   // [workgroups]
  int* pidx;                     // [workgroups]
  long long V;
  int H;
  float eps;
};

template <typename AT, int CH>   // CH = 16-byte chunks per lane = H / 512
__global__ __launch_bounds__(256) void lm_head_kernel(const LmHeadArgs a) {
  typedef Act<AT> A;
  __shared__ float wmax[4];
  __shared__ int widx[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row0 = (long long)blockIdx.x * kLmRowsPerWg;
  // first weight rows in flight before anything else
  const u32x4* Wv = (const u32x4*)a.W;
  const long long rs = a.H / 8;   // 16-byte chunks per row
  u32x4 w[4][CH];
  auto load_rows = [&](long long r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long rr = min(r + q, a.V - 1);
#pragma unroll
      for (int c = 0; c < CH; ++c) w[q][c] = __builtin_nontemporal_load(Wv + rr * rs + lane + 64 * c);
    }
  };
  load_rows(row0 + wave * 16);
  // x, its RMS statistic, the normalised vector of this lane's chunks (kept in registers as fp32)
  float xn[CH][8];
  unsigned xh[CH][4];            // the normalised vector as packed activations (what HF's RMSNorm returns: already rounded)
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const u32x4 xv = *(const u32x4*)(a.x + (lane + 64 * c) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xn[c][2 * e] = A::to_f32(xv[e] & 0xffffu);
      xn[c][2 * e + 1] = A::to_f32(xv[e] >> 16);
      ss = __builtin_fmaf(xn[c][2 * e], xn[c][2 * e], __builtin_fmaf(xn[c][2 * e + 1], xn[c][2 * e + 1], ss));
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);   // every wave holds all of x: no LDS needed
  const float r = __builtin_amdgcn_rsqf(ss / (float)a.H + a.eps);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const u32x4 wv = *(const u32x4*)(a.nw + (lane + 64 * c) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xn[c][2 * e] = A::to_f32(A::from_f32(A::to_f32(A::from_f32(xn[c][2 * e] * r)) * A::to_f32(wv[e] & 0xffffu)));
      xn[c][2 * e + 1] = A::to_f32(A::from_f32(A::to_f32(A::from_f32(xn[c][2 * e + 1] * r)) * A::to_f32(wv[e] >> 16)));
      xh[c][e] = (unsigned)A::from_f32(xn[c][2 * e]) | ((unsigned)A::from_f32(xn[c][2 * e + 1]) << 16);
    }
  }
  float best = -3.0e38f;
  int besti = 0x7fffffff;
  for (int pass = 0; pass < 4; ++pass) {
    const long long r0 = row0 + wave * 16 + pass * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q] = dot2_acc<AT>(w[q][c][e], xh[c][e], acc[q]);
    if (pass < 3) load_rows(r0 + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc[q] += __shfl_xor(acc[q], off, 64);
      const long long row = r0 + q;
      if (row < a.V) {
        const unsigned short h = A::from_f32(acc[q]);
        if (lane == 0) a.logits[row] = h;
        const float v = A::to_f32(h);             // argmax over the ROUNDED logits, like torch.argmax on the fp16 tensor
        if (v > best) { best = v; besti = (int)row; }
      }
    }
  }
  if (lane == 0) { wmax[wave] = best; widx[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    float m = wmax[0]; int mi = widx[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (wmax[k] > m || (wmax[k] == m && widx[k] < mi)) { m = wmax[k]; mi = widx[k]; }
    a.pmax[blockIdx.x] = m;
    a.pidx[blockIdx.x] = mi;
  }
}

// one workgroup: reduce the per-workgroup maxima (lowest index wins ties), store the consumed token into the
// output sequence at *pos, publish the next token, advance the position
__global__ __launch_bounds__(256) void argmax_advance_kernel(const float* pmax, const int* pidx, int n, long long* tok, int* pos,
                                                             long long* out_tokens, long long out_len) {
  __shared__ float sm[256];
  __shared__ int si[256];
  float m = -3.0e38f; int mi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = pmax[i]; const int ix = pidx[i];
    if (v > m || (v == m && ix < mi)) { m = v; mi = ix; }
  }
  sm[threadIdx.x] = m; si[threadIdx.x] = mi;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float v = sm[threadIdx.x + s]; const int ix = si[threadIdx.x + s];
      if (v > sm[threadIdx.x] || (v == sm[threadIdx.x] && ix < si[threadIdx.x])) { sm[threadIdx.x] = v; si[threadIdx.x] = ix; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int p = *pos;
    if (out_tokens && p >= 0 && p < out_len) out_tokens[p] = *tok;
    *tok = (long long)si[0];
    *pos = p + 1;
  }
}

}  // namespace paro

extern "C" int64_t paro_lm_head_workspace_bytes(int64_t vocab) {
  if (vocab < 1) return -1;
  return ((vocab + paro::kLmRowsPerWg - 1) / paro::kLmRowsPerWg) * 8;
}

extern "C" int paro_lm_head(const void* x, const void* norm_weight, const void* W, void* logits, int64_t vocab, int64_t hidden,
                            float eps, int act_dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace paro;
  if (!x || !norm_weight || !W || !logits || !workspace) return fail(PARO_ERR_INVALID, "null pointer");
  if (vocab < 1 || vocab > 0x7fffffff) return fail(PARO_ERR_INVALID, "vocab out of range");
  if (hidden < 512 || hidden % 512 != 0 || hidden > 4096) return fail(PARO_ERR_UNSUPPORTED, "hidden must be a multiple of 512 up to 4096 (got %lld)", (long long)hidden);
  if (workspace_bytes < paro_lm_head_workspace_bytes(vocab)) return fail(PARO_ERR_INVALID, "lm_head workspace too small");
  const int64_t wgs = (vocab + kLmRowsPerWg - 1) / kLmRowsPerWg;
  LmHeadArgs a;
  a.x = (const unsigned short*)x;
  a.nw = (const unsigned short*)norm_weight;
  a.W = (const unsigned short*)W;
  a.logits = (unsigned short*)logits;
  a.pmax = (float*)workspace;
  a.pidx = (int*)((char*)workspace + wgs * 4);
  a.V = vocab;
  a.H = (int)hidden;
  a.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  const int ch = (int)(hidden / 512);
  const bool h16 = act_dtype == PARO_DTYPE_F16;
  if (!h16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
#define PARO_LM(CH) \
  case CH: \
    if (h16) hipLaunchKernelGGL((lm_head_kernel<f16, CH>), dim3((unsigned)wgs), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((lm_head_kernel<bf16, CH>), dim3((unsigned)wgs), dim3(256), 0, st, a); \
    break;
  switch (ch) {
    PARO_LM(1) PARO_LM(2) PARO_LM(3) PARO_LM(4) PARO_LM(5) PARO_LM(6) PARO_LM(7) PARO_LM(8)
    default: return fail(PARO_ERR_UNSUPPORTED, "hidden = %lld is not built (512 x 1..8)", (long long)hidden);
  }
#undef PARO_LM
  return check_launch("paro_lm_head");
}

extern "C" int paro_argmax_advance(const void* workspace, int64_t vocab, int64_t* token, int32_t* pos, int64_t* out_tokens,
                                   int64_t out_len, void* stream) {
  using namespace paro;
  if (!workspace || !token || !pos) return fail(PARO_ERR_INVALID, "null pointer");
  const int64_t wgs = (vocab + kLmRowsPerWg - 1) / kLmRowsPerWg;
  hipLaunchKernelGGL(argmax_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     (const int*)((const char*)workspace + wgs * 4), (int)wgs, (long long*)token, pos, (long long*)out_tokens,
                     (long long)out_len);
  return check_launch("paro_argmax_advance");
}

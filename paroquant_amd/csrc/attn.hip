// Batch-1 decode attention for the decode harness: everything between the merged qkv projection
// and o_proj of one decoder layer in ONE launch -- per-head q/k RMSNorm (Qwen3), rotary embedding, KV-cache append,
// grouped-query attention over positions 0..pos, softmax, P V.  The reference leaves this to HF generate() / vLLM;
// here it exists so that a whole decode step is five launches per layer (qkv GEMV, this, o GEMV, gate_up GEMV,
// down GEMV) inside one HIP graph, and the end-to-end tokens/s of north_star can be measured.
//
// Grid = (KV heads, position chunks of 256).  A workgroup handles one chunk of one KV head for all of that head's
// n_rep = Hq / Hkv query heads (they share every K / V byte): every global load of the chunk -- one K row per thread,
// one channel pair of 64 V rows per thread -- is requested before anything else, scores one position per thread,
// soft-max statistics through LDS, P V from the registers -- a CU ingests ~10 B / clock, so long contexts are spread over many CUs rather than one
// workgroup per head.  With more than one active chunk the partial (max, sum, unnormalised output) triples go to
// a workspace and the LAST workgroup of a KV head to arrive (agent-scope release -> ticket -> acquire; no spinning,
// so no dependence on dispatch order) merges them.  The position comes from DEVICE memory, so one captured graph
// replays for every token: the grid always covers max_positions, chunks beyond `pos` exit at once.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace paro {

constexpr int kChunk = 256;

struct AttnArgs {
  const unsigned short* qkv;   // [(Hq + 2 Hkv) * hd]: q heads, k heads, v heads of this token
  unsigned short* kcache;      // [Hkv][T_max][hd]
  unsigned short* vcache;
  unsigned short* out;         // [Hq * hd]
  const int* pos;              // device scalar: 0-based position of this token
  const float* rope;           // [T_max][hd]: cos[0 .. hd/2) then sin[0 .. hd/2) of every position
  const unsigned short* qnw;   // [hd] q-norm weight or null
  const unsigned short* knw;   // [hd] k-norm weight or null
  float* part;                 // workspace: [Hkv][chunks][n_rep][hd + 2] partial results
  unsigned* ticket;            // workspace: [Hkv] arrival counters (zero between launches)
  float eps, scale;
  int Hq, Hkv, hd, T_max, chunks;
  int dbg;                     // PARO_ATTN_DBG: stop after phase N (timing ablation; wrong results)
};

// NREP = query heads per KV head rounded up to a power of two: a COMPILE-TIME bound, so that the score and P V
// loops unroll without a branch per iteration (with a run-time bound every iteration became its own basic block and
// paid the LDS latency of its probability reads: the P V phase alone took 18 us for 256 positions).  Padding heads
// have zero queries and are never stored.
template <typename AT, int HD, int NREP>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs a) {
  typedef Act<AT> A;
  constexpr int hd = HD, half = HD / 2;
  constexpr int GROUPS = 256 / half;           // thread = (group, channel pair) in the P V phase: 4 (hd 128) / 8 (hd 64)
  constexpr int VN = kChunk / GROUPS;          // cache rows per thread in the P V phase
  __shared__ __attribute__((aligned(16))) float qs[NREP * HD];        // [NREP][hd] roped queries * scale
  __shared__ __attribute__((aligned(16))) float sc[NREP * kChunk];    // [NREP][chunk] scores -> probabilities (0 past the chunk)
  __shared__ __attribute__((aligned(16))) float accs[GROUPS * NREP * HD];  // [groups][NREP][hd] partial outputs
  __shared__ float knew[HD], red[8 * 8];
  __shared__ __attribute__((aligned(16))) unsigned short q16[16 * HD];   // [16 MFMA rows][hd] roped queries (activation dtype), rows >= n_rep zero
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, s = blockIdx.y;
  const int n_rep = a.Hq / a.Hkv;
  const int p0 = s * kChunk;
  // ---- every global load of the chunk is requested up front (a dependent global access costs ~1-2 us at this
  // occupancy).  K in v_mfma_f32_16x16x32 B-fragment order: wave w owns positions 64 w .. 64 w + 63 of the chunk as four
  // tiles of 16; lane (kb = l >> 4, n = l & 15) holds dims 32 i + 8 kb .. + 7 of position 16 t + n for k-step i: kw[t][i].
  // V: this thread's channel pair of VN cache rows.  Chunk 0 always takes part, so ITS loads do not wait for the
  // position to arrive from device memory (rows clamped to the cache, masked once `pos` is known); later chunks first
  // learn from `pos` whether they run at all.
  constexpr int KS = HD / 32;
  u32x4 kw[4][KS];
  const int dq = tid % half, grp = tid / half;
  unsigned vv[VN];
  const unsigned vtok = *((const unsigned*)(a.qkv + (int64_t)(a.Hq + a.Hkv) * hd + (int64_t)h * hd) + dq);   // this token's v
  // the loads themselves are unconditional and clamped to the cache (a load inside a select becomes a branch per row:
  // 64 serialised round trips); rows past the chunk and the new token's row are masked where they are consumed
  auto issue_loads = [&]() {
    const int kb = lane >> 4, nn = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = min(p0 + wave * 64 + t * 16 + nn, a.T_max - 1);
      const u32x4* kr = (const u32x4*)(a.kcache + ((int64_t)h * a.T_max + row) * hd) + kb;
#pragma unroll
      for (int i = 0; i < KS; ++i) kw[t][i] = kr[4 * i];
    }
    const unsigned* vbase = (const unsigned*)(a.vcache + (int64_t)h * a.T_max * hd) + dq;
#pragma unroll
    for (int u = 0; u < VN; ++u) vv[u] = vbase[(int64_t)min(p0 + grp + u * GROUPS, a.T_max - 1) * half];
  };
  if (s == 0) issue_loads();
  const int pos = *a.pos;
  if (p0 > pos || pos >= a.T_max || pos < 0) return;  // chunk beyond the current position; a position outside the cache writes nothing
  const int n_act = pos / kChunk + 1;                 // chunks that take part
  const int cn = min(kChunk, pos + 1 - p0);           // positions of this chunk
  const bool own_new = (pos - p0) < kChunk;           // this chunk holds the new token's position
  if (s != 0) issue_loads();

  // padding query heads are zero, scores past the chunk are zero: the loops below need no bounds
  for (int e = tid; e < NREP * HD; e += 256) qs[e] = 0.f;
  for (int e = tid; e < 16 * HD / 2; e += 256) ((unsigned*)q16)[e] = 0u;
  for (int e = tid; e < NREP * kChunk; e += 256) sc[e] = 0.f;
  __syncthreads();

  // ---- step 1: per-head RMSNorm (optional) + rotary embedding of the n_rep query heads (and of the new key)
  const float* rp = a.rope + (int64_t)pos * hd;
  for (int v = wave; v <= n_rep; v += 4) {          // vector v < n_rep: query head, v == n_rep: the key
    const bool isk = v == n_rep;
    if (isk && !own_new) continue;
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
        x0 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(x0 * r)) * A::to_f32(nw[lane])));
        x1 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(x1 * r)) * A::to_f32(nw[lane + half])));
      }
    }
    if (act) {
      // rotate_half convention, cos / sin rounded to the activation dtype like HF's rotary embedding does
      const float c = A::to_f32(A::from_f32(rp[lane])), sn = A::to_f32(A::from_f32(rp[half + lane]));
      const float y0 = A::to_f32(A::from_f32(x0 * c - x1 * sn)), y1 = A::to_f32(A::from_f32(x1 * c + x0 * sn));
      if (isk) {
        knew[lane] = y0;
        knew[lane + half] = y1;
        unsigned short* kc = a.kcache + ((int64_t)h * a.T_max + pos) * hd;
        kc[lane] = A::from_f32(y0);
        kc[lane + half] = A::from_f32(y1);
      } else {
        qs[v * hd + lane] = y0 * a.scale;
        qs[v * hd + lane + half] = y1 * a.scale;
        q16[v * hd + lane] = A::from_f32(y0);            // exact: y0 / y1 are already rounded to the activation dtype
        q16[v * hd + lane + half] = A::from_f32(y1);
      }
    }
  }
  if (own_new && tid < hd)
    a.vcache[((int64_t)h * a.T_max + pos) * hd + tid] = a.qkv[(int64_t)(a.Hq + a.Hkv) * hd + (int64_t)h * hd + tid];
  __syncthreads();
  if (a.dbg == 1) return;

  // ---- step 2: scores s[j][p] = q_j . K[p] on the matrix cores: A = queries (row m = head, zero rows past n_rep),
  // B = the K fragments requested at the top; D[row 4 (l >> 4) + r][col l & 15] = (head, position)
  {
    typedef typename A::vec8 vec8;
    vec8 qa[KS];
    const int kb = lane >> 4, mm = lane & 15;
#pragma unroll
    for (int i = 0; i < KS; ++i) qa[i] = *(const vec8*)(q16 + mm * hd + 32 * i + 8 * kb);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KS; ++i) dacc = A::mfma(qa[i], __builtin_bit_cast(vec8, kw[t][i]), dacc);
      const int pp = wave * 64 + t * 16 + mm;
      if (pp < cn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 4 * kb + r;
          if (j < NREP) sc[j * kChunk + pp] = dacc[r] * a.scale;
        }
      }
    }
  }
  __syncthreads();
  // the new token's key is not in the cache yet (it was written above by this workgroup): its column is recomputed
  // from the LDS copy, one wave per query head
  if (own_new) {
    for (int j = wave; j < n_rep; j += 4) {
      float d = 0.f;
      for (int e = lane; e < hd; e += 64) d = __builtin_fmaf(qs[j * hd + e], knew[e], d);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
      if (lane == 0) sc[j * kChunk + (pos - p0)] = d;
    }
  }
  __syncthreads();
  if (a.dbg == 2) return;

  // ---- step 3: chunk-local soft-max statistics: m_j = max_p s, e = exp(s - m), l_j = sum e
  for (int j = wave; j < n_rep; j += 4) {             // one wave per query head
    float m = -3.0e38f;
    for (int p = lane; p < cn; p += 64) m = fmaxf(m, sc[j * kChunk + p]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float l = 0.f;
    for (int p = lane; p < cn; p += 64) {
      const float e = __builtin_amdgcn_exp2f((sc[j * kChunk + p] - m) * 1.4426950408889634f);
      sc[j * kChunk + p] = e;
      l += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off, 64);
    if (lane == 0) {
      red[j * 8] = m;
      red[j * 8 + 1] = l;
    }
  }
  __syncthreads();
  if (a.dbg == 3) return;

  // ---- step 4: o_j[d] = sum_p e_j[p] V[p][d] from the rows requested at the top (probabilities past the chunk are 0)
  {
    float o0[NREP], o1[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) o0[j] = o1[j] = 0.f;
    const float* scg = sc + grp;
#pragma unroll
    for (int u = 0; u < VN; ++u) {
      const int p = grp + u * GROUPS;                 // rows past the chunk -> 0, the new row -> this token's v
      const unsigned vw = (p < cn) ? ((p0 + p) != pos ? vv[u] : vtok) : 0u;
      const float v0 = A::to_f32(vw & 0xffffu), v1 = A::to_f32(vw >> 16);
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        const float e = scg[j * kChunk + u * GROUPS];
        o0[j] = __builtin_fmaf(e, v0, o0[j]);
        o1[j] = __builtin_fmaf(e, v1, o1[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      accs[(grp * NREP + j) * hd + 2 * dq] = o0[j];
      accs[(grp * NREP + j) * hd + 2 * dq + 1] = o1[j];
    }
  }
  __syncthreads();
  if (n_act == 1) {
    // the only chunk: normalise and write the output
    for (int e = tid; e < n_rep * hd; e += 256) {
      const int j = e / hd, d = e % hd;
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) v += accs[(g * NREP + j) * hd + d];
      a.out[((int64_t)h * n_rep + j) * hd + d] = A::from_f32(v / red[j * 8 + 1]);
    }
    return;
  }
  // ---- several chunks: publish this chunk's (o, m, l), the last arriver of the KV head merges
  float* mine = a.part + (((int64_t)h * a.chunks + s) * n_rep) * (hd + 2);
  for (int e = tid; e < n_rep * hd; e += 256) {
    const int j = e / hd, d = e % hd;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) v += accs[(g * NREP + j) * hd + d];
    mine[j * (hd + 2) + d] = v;
  }
  if (tid < n_rep) {
    mine[tid * (hd + 2) + hd] = red[tid * 8];
    mine[tid * (hd + 2) + hd + 1] = red[tid * 8 + 1];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the release's write-back has drained before the ticket is taken
    const unsigned t = __hip_atomic_fetch_add(a.ticket + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (t == (unsigned)(n_act - 1)) ? 1u : 0u;
    if (last_flag) {
      __hip_atomic_store(a.ticket + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  if (!last_flag) return;
  const float* base = a.part + ((int64_t)h * a.chunks) * n_rep * (hd + 2);
  for (int e = tid; e < n_rep * hd; e += 256) {
    const int j = e / hd, d = e % hd;
    float M = -3.0e38f;
    for (int c = 0; c < n_act; ++c) M = fmaxf(M, base[((int64_t)c * n_rep + j) * (hd + 2) + hd]);
    float num = 0.f, den = 0.f;
    for (int c = 0; c < n_act; ++c) {
      const float* pc = base + ((int64_t)c * n_rep + j) * (hd + 2);
      const float w = __builtin_amdgcn_exp2f((pc[hd] - M) * 1.4426950408889634f);
      num = __builtin_fmaf(w, pc[d], num);
      den = __builtin_fmaf(w, pc[hd + 1], den);
    }
    a.out[((int64_t)h * n_rep + j) * hd + d] = A::from_f32(num / den);
  }
}

}  // namespace paro

extern "C" int64_t paro_attn_decode_workspace_bytes(int n_heads, int n_kv_heads, int head_dim, int max_positions) {
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || head_dim < 2 || max_positions < 1) return -1;
  const int64_t n_rep = n_heads / n_kv_heads, chunks = (max_positions + paro::kChunk - 1) / paro::kChunk;
  return 256 + (int64_t)n_kv_heads * chunks * n_rep * (head_dim + 2) * 4;   // tickets (zero-filled by the caller once) + partials
}

extern "C" int paro_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, const int32_t* pos, const float* rope,
                                const void* q_norm_w, const void* k_norm_w, float eps, float scale, int n_heads,
                                int n_kv_heads, int head_dim, int max_positions, int act_dtype, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  using namespace paro;
  if (!qkv || !kcache || !vcache || !out || !pos || !rope) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0) return fail(PARO_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (n_kv_heads > 64) return fail(PARO_ERR_UNSUPPORTED, "at most 64 KV heads");
  if (n_heads / n_kv_heads > 8) return fail(PARO_ERR_UNSUPPORTED, "at most 8 query heads per KV head (got %d)", n_heads / n_kv_heads);
  if (head_dim != 64 && head_dim != 128) return fail(PARO_ERR_UNSUPPORTED, "head_dim must be 64 or 128 (got %d)", head_dim);
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return fail(PARO_ERR_INVALID, "q / k norm weights come together");
  const int64_t need = paro_attn_decode_workspace_bytes(n_heads, n_kv_heads, head_dim, max_positions);
  if (max_positions < 1 || max_positions > 65535 * kChunk) return fail(PARO_ERR_INVALID, "max_positions out of range");
  if (!workspace || workspace_bytes < need)
    return fail(PARO_ERR_INVALID, "attention workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
  AttnArgs a;
  a.qkv = (const unsigned short*)qkv;
  a.kcache = (unsigned short*)kcache;
  a.vcache = (unsigned short*)vcache;
  a.out = (unsigned short*)out;
  a.pos = pos;
  a.rope = rope;
  a.qnw = (const unsigned short*)q_norm_w;
  a.knw = (const unsigned short*)k_norm_w;
  a.ticket = (unsigned*)workspace;
  a.part = (float*)((char*)workspace + 256);
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
#define PARO_ATTN_LAUNCH(T, HD, NR) hipLaunchKernelGGL((attn_decode_kernel<T, HD, NR>), grid, dim3(256), 0, st, a)
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

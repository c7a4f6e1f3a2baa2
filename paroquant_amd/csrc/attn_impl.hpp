// Batch-1 decode attention, device side (see attn.hip for the design notes and the host entry points): AttnArgs, the kernel body as a
// device function -- shared by attn_decode_kernel (attn.hip) and the attention row of a fused qkv launch (gemv_impl.hpp) -- and the
// kernel itself.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace paro {


// positions per workgroup: 256 for short caches (one chunk = no merge up to 256 positions), 128 above that (a CU
// ingests ~13 B / clock: smaller chunks spread a context over more CUs; each extra chunk costs the in-launch merge)
constexpr int attn_chunk(int max_positions) { return max_positions <= 512 ? 256 : 128; }
constexpr int kAttnWsHeader = 2048;   // bytes of arrival counters in front of the partial results

struct AttnArgs {
  const unsigned short* qkv;   // [(Hq + 2 Hkv) * hd]: q heads, k heads, v heads of this token
  // PARTS builds: the same vector as the fp32 partial sums a K-split qkv projection left (paro_fusion_t.parts_out): [N + 1][4], N = (Hq + 2 Hkv) hd;
  // value = round(((p0 + p1) + p2) + p3) * rstd), rstd = rsqrt(sum(row N) / norm_dim + norm_eps) when norm_dim > 0 (the projection's RMSNorm
  // prologue, whose scalar a K-split launch cannot apply itself), else 1
  const f32x4* qkv_parts;
  float norm_dim, norm_eps;
  unsigned short* kcache;      // [Hkv][T_max][hd]
  unsigned short* vcache;      // [Hkv][hd][T_max]  (position-contiguous: the P V product's MFMA B fragments are 16-byte loads)
  unsigned short* out;         // [Hq * hd]
  const int* pos;              // device scalar: 0-based position of this token
  const float* rope;           // [T_max][hd]: cos[0 .. hd/2) then sin[0 .. hd/2) of every position
  const unsigned short* qnw;   // [hd] q-norm weight or null
  const unsigned short* knw;   // [hd] k-norm weight or null
  float* part;                 // workspace: [Hkv][chunks][n_rep][hd + 2] partial results
  unsigned* ticket;            // workspace: [Hkv] arrival counters (zero between launches); SPLIT: [64 + 4 Hkv + slot]
  float* split_o;              // SPLIT: [Hq * hd][4] un-normalised outputs per slot
  float* split_ml;             // SPLIT: [Hq][8]: the slots' maxima [0..3] and sums [4..7]; a slot nobody filled holds (-3e38, 0)
  float eps, scale;
  int Hq, Hkv, hd, T_max, chunks;
  int dbg;                     // PARO_ATTN_DBG: stop after phase N (timing ablation; wrong results)
};

// Cross-lane reductions on the VALU's DPP paths (a dependent step costs ~8 cycles; a __shfl_xor step is a ds_bpermute
// round trip of ~100): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror leave the reduction of
// every 16-lane row in all of its lanes; row_bcast:15 / row_bcast:31 carry on to the whole wave (total in lane 63).
template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_f(float a) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, RMASK, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1, 0xf>(v);
  v += dpp_f<0x4E, 0xf>(v);
  v += dpp_f<0x141, 0xf>(v);
  v += dpp_f<0x140, 0xf>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xf>(v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {   // the same value in every lane
  v = row16_sum(v);
  v += dpp_f<0x142, 0xa>(v);
  v += dpp_f<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Hand-over of partial results between workgroups of ONE launch without cache-wide fences: the producer's stores are agent-scope
// relaxed atomics (write-through), it waits for their completion (s_waitcnt vmcnt(0)) and takes a ticket; the last arriver reads with
// agent-scope relaxed atomic loads.  An agent-scope release / acquire pair instead writes back and invalidates the whole L2 of the
// XCD: ~2 us per launch here (and 7.8 -> 5.3 us in gdn_step_kernel, which uses the same scheme).
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NREP = query heads per KV head rounded up to a power of two: a COMPILE-TIME bound, so that every loop over heads
// unrolls without a branch per iteration.  Padding heads have zero queries and are never stored.
//
// Latency chain of a workgroup (what the kernel is bound by at short contexts: 8 workgroups, ~10 us of dependent
// steps): [K / V of the chunk + this token's q / k / v requested] -> q / k norm, RoPE, KV append -> ONE barrier ->
// scores on the matrix cores (the new key patched into the K fragments) -> soft-max of each wave's 64 positions in
// registers (DPP row reductions, no LDS, no barrier) -> P V of the wave's own rows -> ONE barrier -> the four waves'
// (max, sum, partial output) triples merged like chunks are.
// The body is a device function of (KV head h, position chunk s): the kernel of its own launch below, and the attention row of a
// fused qkv launch (gemv_impl.hpp, FUSED | 128; TAGGED: q / k / v arrive as {fp32, launch tag} granules from the same launch).
template <typename AT, int HD, int NREP, int CH, bool PARTS, bool SPLIT = false, bool TAGGED = false>
__device__ __forceinline__ void attn_decode_body(const AttnArgs& a, const int h, const int s, const unsigned qtag) {
  // CH = 0 (SPLIT only): the chunk size follows the POSITION -- 64 positions per workgroup while the context fits four such slots
  // (no ticket up to 256 positions, twice the CUs per head), 128 beyond; the grid is sized for 64
  constexpr int CHM = CH ? CH : 128;           // largest chunk this instantiation can run (LDS sizing)
  static_assert(CH != 0 || SPLIT, "position-dependent chunks are a split-launch feature");
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int hd = HD, half = HD / 2;
  constexpr int DT = HD / 16;                  // 16-column output tiles of the P V product
  // SPLIT builds cut the two products differently: the scores by POSITIONS (wave w: positions WP w .. of the chunk), P V by
  // DIMENSIONS (wave w: dims (HD / 4) w .. over all positions of the chunk, after one barrier) -- every output element is then complete in one
  // wave's registers and goes straight to memory: no partial outputs through LDS, no merge of the waves
  constexpr int VT = SPLIT ? HD / 64 : DT;     // V tiles (16 dims) a wave holds
  static_assert(SPLIT ? (CH == 128 || CH == 64 || CH == 0) : CH >= 128, "split launches work on 64- or 128-position chunks, the others on 128 / 256");
  constexpr float kLog2e = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) float pw[4 * NREP * (CHM / 4)];  // [wave][head][WP] unnormalised probabilities of the wave's positions
  __shared__ __attribute__((aligned(16))) float accs[SPLIT ? 4 : 4 * NREP * HD];      // [wave][head][hd] partial outputs
  __shared__ float st[4 * NREP * 2];                                       // [wave][head] (max, sum) of the wave's positions
  __shared__ __attribute__((aligned(16))) unsigned short q16[16 * HD];     // [16 MFMA rows][hd] roped queries (activation dtype), rows >= n_rep zero
  __shared__ __attribute__((aligned(16))) unsigned short k16[HD];          // the new token's roped key (activation dtype)
  __shared__ __attribute__((aligned(16))) unsigned short v16[HD];          // the new token's value
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_rep = a.Hq / a.Hkv;
  const int kb = lane >> 4, mm = lane & 15;
  // The position first, through the scalar cache (one ~300-cycle round trip, nothing queued in front of it): chunks
  // beyond it leave at once, and the rotary row -- the only load that depends on it -- goes out ahead of the K / V stream.
  constexpr int KS = HD / 32;                  // MFMA k-steps of a score (32 head dims each)
  // this token's query heads and key (vector v < n_rep: query head, v == n_rep: the key; wave w takes v = w, w + 4, ..),
  // the norm weights, this token's value: none of it depends on the position, so it is in flight while the position
  // makes its round trip
  constexpr int ITER = (NREP + 1 + 3) / 4;
  const bool act = lane < half;
  const int li = act ? lane : 0;
  float x0[ITER], x1[ITER], w0[ITER], w1[ITER];
  f32x4 pn = {0.f, 0.f, 0.f, 0.f};            // PARTS: the projection's partial sums of squares
  if constexpr (PARTS && !TAGGED) pn = a.qkv_parts[(a.Hq + 2 * a.Hkv) * hd];
  f32x4 pq0[PARTS ? ITER : 1], pq1[PARTS ? ITER : 1], pv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int v = min(wave + 4 * it, n_rep);
    const bool isk = v == n_rep;
    const int col0 = isk ? a.Hq * hd + h * hd : (h * n_rep + v) * hd;
    if constexpr (PARTS) {
      if constexpr (!TAGGED) {
        pq0[it] = a.qkv_parts[col0 + li];
        pq1[it] = a.qkv_parts[col0 + li + half];
      }
    } else {
      x0[it] = A::to_f32(a.qkv[col0 + li]);
      x1[it] = A::to_f32(a.qkv[col0 + li + half]);
    }
    const unsigned short* nw = isk ? a.knw : a.qnw;
    w0[it] = nw ? A::to_f32(nw[li]) : 1.f;
    w1[it] = nw ? A::to_f32(nw[li + half]) : 1.f;
  }
  unsigned short vnew = 0;                                                                                    // this token's v[tid]
  if constexpr (PARTS && !TAGGED) pv = a.qkv_parts[(a.Hq + a.Hkv) * hd + h * hd + (tid < hd ? tid : 0)];
  else vnew = a.qkv[(int64_t)(a.Hq + a.Hkv) * hd + (int64_t)h * hd + (tid < hd ? tid : 0)];
  // The position, through the scalar cache (one ~300-cycle round trip): chunks beyond it leave at once; nothing past
  // the chunk's last position is requested below (a CU ingests ~13 B / clock: the 128 KiB of a full chunk are ~4.5 us,
  // the floor of this kernel).
  int pos;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pos) : "s"(a.pos) : "memory");
  if (pos >= a.T_max || pos < 0) return;              // a position outside the cache writes nothing
  // ---- everything below depends on the chunk size: a generic lambda, instantiated once (CH != 0) or for both sizes (CH == 0)
  auto rest = [&](auto ch_tag) {
  constexpr int kChunk = decltype(ch_tag)::value;    // positions of a workgroup
  constexpr int WP = kChunk / 4;                     // positions of a wave (64 / 32 / 16)
  constexpr int KT = WP / 16;                        // K tiles (16 positions) of a wave
  constexpr int VS = WP >= 32 ? WP / 32 : 1;         // V k-steps (32 positions) of a wave
  constexpr int VK = SPLIT ? kChunk / 32 : VS;       // V k-steps (32 positions) a wave holds
  const int p0 = s * kChunk;
  if (p0 > pos) return;                               // chunk beyond the current position
  const int n_act = pos / kChunk + 1;                 // chunks that take part
  const int cn = min(kChunk, pos + 1 - p0);           // positions of this chunk
  const bool own_new = (pos - p0) < kChunk;           // this chunk holds the new token's position
  const float* rp = a.rope + (int64_t)pos * hd;
  const float rope_c = rp[li], rope_s = rp[half + li];
  // complete q / k / v from the projection's partial sums: the reducer's summation order, the projection's norm scalar, ONE rounding to
  // the activation type (what the projection's own epilogue would have stored)
  auto finish_qkv = [&]() {
    if constexpr (PARTS) {
      if constexpr (TAGGED) {
        // The partial sums come from THIS launch's projection workgroups (gemv_impl.hpp, FUSED | 128) as 8-byte {fp32, launch tag}
        // granules, four slots per element (slot order as above; the last K-slice also writes the unused slots): element e at
        // byte 32 e.  Write-through loads, every slot's tag checked, all of a wave's elements polled again together (bounded).
        // Give-up: NaN (it reaches the attention output and every logit).
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv_parts, 0, (((a.Hq + 2 * a.Hkv) * hd) + 1) * 32, 0x00020000);
        constexpr int NE = 2 * ITER + 2;
        unsigned eo[NE];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int v = min(wave + 4 * it, n_rep);
          const int col0 = v == n_rep ? a.Hq * hd + h * hd : (h * n_rep + v) * hd;
          eo[2 * it] = (unsigned)(col0 + li) * 32u;
          eo[2 * it + 1] = (unsigned)(col0 + li + half) * 32u;
        }
        eo[2 * ITER] = (unsigned)((a.Hq + a.Hkv) * hd + h * hd + (tid < hd ? tid : 0)) * 32u;
        eo[2 * ITER + 1] = (unsigned)((a.Hq + 2 * a.Hkv) * hd) * 32u;
        u32x4 glo[NE], ghi[NE];
        bool wave_bad = true;
        for (int spin = 0; spin < (1 << 16); ++spin) {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            glo[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, eo[e], 0, 16 /* sc1 */));
            ghi[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, eo[e] + 16u, 0, 16 /* sc1 */));
          }
          bool bad = false;
#pragma unroll
          for (int e = 0; e < NE; ++e) bad = bad || glo[e][1] != qtag || glo[e][3] != qtag || ghi[e][1] != qtag || ghi[e][3] != qtag;
          wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
          if (!wave_bad) break;
          __builtin_amdgcn_s_sleep(2);     // (0 .. 16 measured the same end to end: the poll is a ~1 us round trip either way)
        }
        auto val = [&](int e) {
          // (element -> scalar -> bit_cast: __builtin_bit_cast on a vector ELEMENT reads element 0 with hipcc 7.2, NOTES 5.2)
          const unsigned u0 = glo[e][0], u1 = glo[e][2], u2 = ghi[e][0], u3 = ghi[e][2];
          f32x4 r = {__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), __builtin_bit_cast(float, u2), __builtin_bit_cast(float, u3)};
          if (wave_bad) r = (f32x4){__builtin_nanf(""), 0.f, 0.f, 0.f};
          return r;
        };
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          pq0[it] = val(2 * it);
          pq1[it] = val(2 * it + 1);
        }
        pv = val(2 * ITER);
        pn = val(2 * ITER + 1);
      }
      const float rstd = a.norm_dim > 0.f ? __builtin_amdgcn_rsqf((((pn[0] + pn[1]) + pn[2]) + pn[3]) / a.norm_dim + a.norm_eps) : 1.f;
      auto fin = [&](const f32x4& p) { return A::from_f32((((p[0] + p[1]) + p[2]) + p[3]) * rstd); };
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        x0[it] = A::to_f32(fin(pq0[it]));
        x1[it] = A::to_f32(fin(pq1[it]));
      }
      vnew = fin(pv);
    }
  };
  if constexpr (!TAGGED) finish_qkv();
  // ---- the chunk's K and V, all requested here (a dependent global access costs ~1-2 us at this occupancy), as MFMA B
  // fragments: wave w owns positions 64 w .. 64 w + 63 of the chunk.
  //   K [pos][dim]: tile t = positions 16 t .. + 15; lane (kb, mm) holds dims 32 i + 8 kb .. + 7 of position 16 t + mm: kw[t][i]
  //   V [dim][pos]: tile t = dims 16 t .. + 15; lane (kb, mm) holds positions 32 i + 8 kb .. + 7 of dim 16 t + mm: vf[t][i]
  // Every load is issued unconditionally -- a load inside a branch makes the compiler's vmcnt bookkeeping give up and the
  // first use of K then waits for V as well -- but tiles past the chunk's end all read ONE address (the chunk's first
  // row: a single cache line for the whole wave), so they cost no bandwidth.  Rows past the end and the new token's row
  // (not in the cache yet) are masked / patched where they are consumed.
  u32x4 kw[KT][KS];
  u32x4 vf[VT][VK];
  const int vdim0 = SPLIT ? wave * (HD / 4) : 0;     // first dim / first chunk position of this wave's V fragments
  const int vpos0 = SPLIT ? 0 : wave * WP;
  {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const bool need = wave * WP + t * 16 < cn;
      const int row = need ? min(p0 + wave * WP + t * 16 + mm, a.T_max - 1) : p0;
      const u32x4* kr = (const u32x4*)(a.kcache + ((int64_t)h * a.T_max + row) * hd) + (need ? kb : 0);
#pragma unroll
      for (int i = 0; i < KS; ++i) kw[t][i] = kr[need ? 4 * i : 0];
    }
#pragma unroll
    for (int i = 0; i < VK; ++i) {
      const bool need = vpos0 + 32 * i < cn;
#pragma unroll
      for (int t = 0; t < VT; ++t) {
        const unsigned short* vr = a.vcache + ((int64_t)h * hd + (need ? vdim0 + 16 * t + mm : 0)) * a.T_max;
        vf[t][i] = *(const u32x4*)(vr + (need ? min(p0 + vpos0 + 32 * i + 8 * kb, a.T_max - 8) : p0));
      }
    }
  }
  if constexpr (TAGGED) finish_qkv();   // (behind the K / V requests: they are in flight while the projection's workgroups finish)
  // padding query heads are zero (rows >= n_rep of the MFMA A operand)
  for (int e = tid; e < 16 * HD / 2; e += 256) ((unsigned*)q16)[e] = 0u;
  if (tid < hd) v16[tid] = vnew;
  __syncthreads();                                    // q16 is zero before the query heads are written into it

  // ---- step 1: per-head RMSNorm (optional) + rotary embedding of the n_rep query heads (and of the new key)
  {
    // rotate_half convention, cos / sin rounded to the activation dtype like HF's rotary embedding does
    const float c = A::to_f32(A::from_f32(rope_c)), sn = A::to_f32(A::from_f32(rope_s));
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int v = wave + 4 * it;
      const bool isk = v == n_rep;
      float a0 = act ? x0[it] : 0.f, a1 = act ? x1[it] : 0.f;
      if (a.qnw) {
        const float ss = wave_sum(a0 * a0 + a1 * a1);
        const float r = __builtin_amdgcn_rsqf(ss / (float)hd + a.eps);
        // HF: normalise in fp32, round to the activation dtype, then multiply by the weight
        a0 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(a0 * r)) * w0[it]));
        a1 = A::to_f32(A::from_f32(A::to_f32(A::from_f32(a1 * r)) * w1[it]));
      }
      const unsigned short y0 = A::from_f32(a0 * c - a1 * sn), y1 = A::from_f32(a1 * c + a0 * sn);
      if (act && v <= n_rep) {
        if (isk) {
          if (own_new) {
            k16[lane] = y0;
            k16[lane + half] = y1;
            unsigned short* kc = a.kcache + ((int64_t)h * a.T_max + pos) * hd;
            kc[lane] = y0;
            kc[lane + half] = y1;
          }
        } else {
          q16[v * hd + lane] = y0;
          q16[v * hd + lane + half] = y1;
        }
      }
    }
  }
  if (own_new && tid < hd) a.vcache[((int64_t)h * hd + tid) * a.T_max + pos] = vnew;
  __syncthreads();
  if (a.dbg == 1) return;

  // ---- step 2: scores s[j][p] = q_j . K[p] on the matrix cores: A = queries (row m = head, zero rows past n_rep),
  // B = the K fragments requested at the top, the new token's key (not in the cache when they were requested)
  // patched into the fragment rows of its position; D[row 4 kb + r][col mm] = (head, position 16 t + mm of the wave)
  float sc[KT][4];   // [t][r]
  {
    vec8 qa[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) qa[i] = *(const vec8*)(q16 + mm * hd + 32 * i + 8 * kb);
    const int lp = pos - p0;                          // local position of the new token (when own_new)
    const bool patch = own_new && wave == lp / WP && mm == (lp & 15);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (patch && t == ((lp % WP) >> 4)) {
#pragma unroll
        for (int i = 0; i < KS; ++i) kw[t][i] = *(const u32x4*)(k16 + 32 * i + 8 * kb);
      }
      f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KS; ++i) dacc = A::mfma(qa[i], __builtin_bit_cast(vec8, kw[t][i]), dacc);
      const bool inb = wave * WP + t * 16 + mm < cn;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[t][r] = inb ? dacc[r] * a.scale : -3.0e38f;
    }
  }
  if (a.dbg == 2) return;

  // ---- step 3: soft-max of THIS WAVE's 64 positions, in registers: head j = 4 kb + r lives in lane row kb, its 64
  // positions in 16 lanes x 4 tiles: m = max, e = exp(s - m), l = sum e.  (max, sum) and the unnormalised
  // probabilities go to the wave's own LDS block; the four waves are merged at the end like chunks are.
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float mloc = sc[0][r];
#pragma unroll
    for (int t = 1; t < KT; ++t) mloc = fmaxf(mloc, sc[t][r]);
    const float m = row16_max(mloc);
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const bool inb = wave * WP + t * 16 + mm < cn;
      sc[t][r] = inb ? __builtin_amdgcn_exp2f((sc[t][r] - m) * kLog2e) : 0.f;
      l += sc[t][r];
    }
    l = row16_sum(l);
    const int j = 4 * kb + r;
    if (j < NREP) {
#pragma unroll
      for (int t = 0; t < KT; ++t) pw[(wave * NREP + j) * WP + t * 16 + mm] = sc[t][r];
      if (mm == 0) {
        st[(wave * NREP + j) * 2] = m;
        st[(wave * NREP + j) * 2 + 1] = l;
      }
    }
  }
  if constexpr (SPLIT) __syncthreads();   // every wave reads every wave's probabilities below
  else __builtin_amdgcn_wave_barrier();   // the wave reads back what its own lanes wrote (LDS is in order within a wave)
  if (a.dbg == 3) return;

  // ---- step 4: O[j][d] = sum_p e_j[p] V[p][d] over the wave's 64 positions on the matrix cores: A = probabilities
  // (row = head, rounded to the activation dtype as HF's attn_weights.to(dtype) does), B = the V fragments requested at
  // the top with positions past the chunk zeroed (stale cache memory times zero must not make a NaN) and the new
  // token's value patched into its position
  {
    const int lp = pos - p0;
    const bool own_blk = own_new && lp >= vpos0 && lp < vpos0 + 32 * VK;   // the new token's position is among this wave's fragments
    const bool vpatch = own_blk && kb == ((lp >> 3) & 3);
#pragma unroll
    for (int i = 0; i < VK; ++i) {
      // only the 32 positions that hold the chunk's end or the new token need any of this (wave-uniform test)
      const bool pi32 = own_blk && i == ((lp - vpos0) >> 5);
      if (vpos0 + 32 * i + 32 <= cn && !pi32) continue;
      const int nv = cn - (vpos0 + 32 * i + 8 * kb);          // valid positions among this lane's eight
      unsigned mk[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) mk[c] = nv >= 2 * c + 2 ? 0xffffffffu : (nv == 2 * c + 1 ? 0x0000ffffu : 0u);
      const bool pi = vpatch && i == ((lp - vpos0) >> 5);
      // the new token's value replaces one 16-bit slot of one word: keep-mask / insert-shift per word, computed once
      unsigned keep[4], sh[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool hit = pi && c == ((lp >> 1) & 3);
        keep[c] = hit ? ((lp & 1) ? 0x0000ffffu : 0xffff0000u) : 0xffffffffu;
        sh[c] = hit ? ((lp & 1) ? 16u : 0u) : 32u;            // 32: nothing inserted
      }
#pragma unroll
      for (int t = 0; t < VT; ++t) {
        const unsigned nv16 = v16[vdim0 + 16 * t + mm];
#pragma unroll
        for (int c = 0; c < 4; ++c) vf[t][i][c] = (vf[t][i][c] & mk[c] & keep[c]) | (sh[c] < 32u ? nv16 << sh[c] : 0u);
      }
    }
    if constexpr (SPLIT) {
      // the chunk's maximum and sum of head mm from the four waves' (max, sum); a wave's probabilities are rescaled by 2^(m_w - M) as
      // they become the A operand (WP = 32: wave i's positions ARE k-step i; WP = 16: k-step i = waves 2 i, 2 i + 1)
      // (four scalars, not an array: a select between two array elements became a dynamically indexed private array, which the
      // compiler moved to LDS -- and addressing it by work-item id made every wave read the workgroup size from the AQL dispatch
      // packet: one scalar load from the queue's memory, measured at 4 .. 7 us per launch)
      float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, M = 0.f, den = 0.f;
      if (mm < NREP) {
        const float m0 = st[(0 * NREP + mm) * 2], m1 = st[(1 * NREP + mm) * 2], m2 = st[(2 * NREP + mm) * 2], m3 = st[(3 * NREP + mm) * 2];
        M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        f0 = __builtin_amdgcn_exp2f((m0 - M) * kLog2e);
        f1 = __builtin_amdgcn_exp2f((m1 - M) * kLog2e);
        f2 = __builtin_amdgcn_exp2f((m2 - M) * kLog2e);
        f3 = __builtin_amdgcn_exp2f((m3 - M) * kLog2e);
        den = __builtin_fmaf(f0, st[(0 * NREP + mm) * 2 + 1], den);
        den = __builtin_fmaf(f1, st[(1 * NREP + mm) * 2 + 1], den);
        den = __builtin_fmaf(f2, st[(2 * NREP + mm) * 2 + 1], den);
        den = __builtin_fmaf(f3, st[(3 * NREP + mm) * 2 + 1], den);
      }
      const float fk[4] = {f0, f1, f2, f3};                                         // WP = 32: k-step i's factor (compile-time index)
      const float fh[2] = {(kb >> 1) ? f1 : f0, (kb >> 1) ? f3 : f2};               // WP = 16: this lane's half of k-step i
      vec8 pa[VK];
#pragma unroll
      for (int i = 0; i < VK; ++i) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (mm < NREP) {
          const int sw = WP == 32 ? i : 2 * i + (kb >> 1);                      // the wave whose positions these eight are
          const int so = WP == 32 ? 8 * kb : 8 * (kb & 1);
          const float fi = WP == 32 ? fk[i] : fh[i & 1];
          const f32x4 e0 = *(const f32x4*)(pw + (sw * NREP + mm) * WP + so);
          const f32x4 e1 = *(const f32x4*)(pw + (sw * NREP + mm) * WP + so + 4);
          w[0] = (unsigned)A::from_f32(e0[0] * fi) | ((unsigned)A::from_f32(e0[1] * fi) << 16);
          w[1] = (unsigned)A::from_f32(e0[2] * fi) | ((unsigned)A::from_f32(e0[3] * fi) << 16);
          w[2] = (unsigned)A::from_f32(e1[0] * fi) | ((unsigned)A::from_f32(e1[1] * fi) << 16);
          w[3] = (unsigned)A::from_f32(e1[2] * fi) | ((unsigned)A::from_f32(e1[3] * fi) << 16);
        }
        pa[i] = __builtin_bit_cast(vec8, w);
      }
      // where this chunk's triple goes: its slot of the caller's buffer (one chunk per slot), or the workspace (several chunks per
      // slot: merged by the slot's last arriver below)
      const int per = (n_act + 3) >> 2, slot = s / per, n_slots = (n_act + per - 1) / per;
      float* ob;             // element (j, d) at ob[j * ohs + d * oes]
      float* mb;             // (M, den) of head j at mb[j * mhs], mb[j * mhs + mds]
      int ohs, oes, mhs, mds;
      if (per == 1) {
        ob = a.split_o + (int64_t)h * n_rep * hd * 4 + slot; ohs = hd * 4; oes = 4;
        mb = a.split_ml + (int64_t)h * n_rep * 8 + slot; mhs = 8; mds = 4;
      } else {
        ob = a.part + (((int64_t)h * a.chunks + s) * n_rep) * (hd + 2); ohs = hd + 2; oes = 1;
        mb = ob + hd; mhs = hd + 2; mds = 1;
      }
#pragma unroll
      for (int t = 0; t < VT; ++t) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < VK; ++i) o = A::mfma(pa[i], __builtin_bit_cast(vec8, vf[t][i]), o);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 4 * kb + r;
          if (j < n_rep) {
            // (one chunk per slot: the consumer is the NEXT launch, ordinary stores do -- 0.25 us cheaper than write-through ones at the
            // kernel's tail; several: the slot's last arriver reads them in this launch)
            if (per == 1) ob[j * ohs + (vdim0 + 16 * t + mm) * oes] = o[r];
            else st_agent(ob + j * ohs + (vdim0 + 16 * t + mm) * oes, o[r]);
          }
        }
      }
      if (wave == 0 && kb == 0 && mm < n_rep) {
        if (per == 1) {
          mb[mm * mhs] = M;
          mb[mm * mhs + mds] = den;
        } else {
          st_agent(mb + mm * mhs, M);
          st_agent(mb + mm * mhs + mds, den);
        }
      }
      // chunk 0 (always active) marks the slots nobody fills: (max, sum) = (-3e38, 0) -- the consumer skips their outputs
      if (s == 0 && tid < n_rep * 4) {
        const int j = tid >> 2, q = tid & 3;
        if (q >= n_slots) {
          a.split_ml[((int64_t)h * n_rep + j) * 8 + q] = -3.0e38f;
          a.split_ml[((int64_t)h * n_rep + j) * 8 + 4 + q] = 0.f;
        }
      }
      if (per == 1) return;
      // several chunks per slot: the slot's last arriver merges its chunks into the slot's triple
      const int c_first = slot * per, c_count = min(per, n_act - c_first);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        unsigned* tk = a.ticket + 64 + h * 4 + slot;
        const unsigned tkt = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (tkt == (unsigned)(c_count - 1)) ? 1u : 0u;
        if (last_flag) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!last_flag) return;
      const float* base = a.part + ((int64_t)h * a.chunks + c_first) * n_rep * (hd + 2);
      for (int e = tid; e < n_rep * hd; e += 256) {
        const int j = e / hd, d = e % hd;
        const float* pj = base + (int64_t)j * (hd + 2);
        const int64_t cstride = (int64_t)n_rep * (hd + 2);
        float Ms = -3.0e38f;
        for (int c0 = 0; c0 < c_count; c0 += 8) {
          float mv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) mv[q] = ld_agent(pj + (int64_t)min(c0 + q, c_count - 1) * cstride + hd);
#pragma unroll
          for (int q = 0; q < 8; ++q) Ms = fmaxf(Ms, mv[q]);
        }
        float num = 0.f, dn = 0.f;
        for (int c0 = 0; c0 < c_count; c0 += 8) {
          float mv[8], lv[8], ov[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float* pc = pj + (int64_t)min(c0 + q, c_count - 1) * cstride;
            mv[q] = ld_agent(pc + hd);
            lv[q] = ld_agent(pc + hd + 1);
            ov[q] = ld_agent(pc + d);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float wq = (c0 + q < c_count) ? __builtin_amdgcn_exp2f((mv[q] - Ms) * kLog2e) : 0.f;
            num = __builtin_fmaf(wq, ov[q], num);
            dn = __builtin_fmaf(wq, lv[q], dn);
          }
        }
        a.split_o[(((int64_t)h * n_rep + j) * hd + d) * 4 + slot] = num;
        if (d == 0) {
          a.split_ml[((int64_t)h * n_rep + j) * 8 + slot] = Ms;
          a.split_ml[((int64_t)h * n_rep + j) * 8 + 4 + slot] = dn;
        }
      }
      return;
    } else {
    vec8 pa[VS];
#pragma unroll
    for (int i = 0; i < VS; ++i) {
      u32x4 w = {0u, 0u, 0u, 0u};
      if (mm < NREP) {
        const f32x4 e0 = *(const f32x4*)(pw + (wave * NREP + mm) * WP + 32 * i + 8 * kb);
        const f32x4 e1 = *(const f32x4*)(pw + (wave * NREP + mm) * WP + 32 * i + 8 * kb + 4);
        w[0] = (unsigned)A::from_f32(e0[0]) | ((unsigned)A::from_f32(e0[1]) << 16);
        w[1] = (unsigned)A::from_f32(e0[2]) | ((unsigned)A::from_f32(e0[3]) << 16);
        w[2] = (unsigned)A::from_f32(e1[0]) | ((unsigned)A::from_f32(e1[1]) << 16);
        w[3] = (unsigned)A::from_f32(e1[2]) | ((unsigned)A::from_f32(e1[3]) << 16);
      }
      pa[i] = __builtin_bit_cast(vec8, w);
    }
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < VS; ++i) o = A::mfma(pa[i], __builtin_bit_cast(vec8, vf[t][i]), o);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * kb + r;
        if (j < NREP) accs[(wave * NREP + j) * hd + 16 * t + mm] = o[r];
      }
    }
    }
  }
  if constexpr (!SPLIT) {
  if (a.dbg == 4) return;
  __syncthreads();
  // ---- the four waves' (max, sum, partial output) -> the chunk's: M = max m_w, num = sum 2^(m_w - M) o_w, den likewise
  auto chunk_value = [&](int j, int d, float& M, float& den) {
    M = fmaxf(fmaxf(st[(0 * NREP + j) * 2], st[(1 * NREP + j) * 2]), fmaxf(st[(2 * NREP + j) * 2], st[(3 * NREP + j) * 2]));
    float num = 0.f;
    den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __builtin_amdgcn_exp2f((st[(w * NREP + j) * 2] - M) * kLog2e);
      den = __builtin_fmaf(f, st[(w * NREP + j) * 2 + 1], den);
      num = __builtin_fmaf(f, accs[(w * NREP + j) * hd + d], num);
    }
    return num;
  };
  if (n_act == 1) {
    // the only chunk: normalise and write the output
    for (int e = tid; e < n_rep * hd; e += 256) {
      const int j = e / hd, d = e % hd;
      float M, den;
      const float num = chunk_value(j, d, M, den);
      a.out[((int64_t)h * n_rep + j) * hd + d] = A::from_f32(num / den);
    }
    return;
  }
  // ---- several chunks: publish this chunk's (o, m, l), the last arriver of the KV head merges
  float* mine = a.part + (((int64_t)h * a.chunks + s) * n_rep) * (hd + 2);
  for (int e = tid; e < n_rep * hd; e += 256) {
    const int j = e / hd, d = e % hd;
    float M, den;
    st_agent(mine + j * (hd + 2) + d, chunk_value(j, d, M, den));
    if (d == 0) {
      st_agent(mine + j * (hd + 2) + hd, M);
      st_agent(mine + j * (hd + 2) + hd + 1, den);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's write-through stores have completed ...
  __syncthreads();                                     // ... every thread's have
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(a.ticket + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (t == (unsigned)(n_act - 1)) ? 1u : 0u;
    if (last_flag) __hip_atomic_store(a.ticket + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
  }
  __syncthreads();
  if (!last_flag) return;
  // the last arriver merges the chunks: (max, sum, partial output) triples of up to hundreds of chunks, read eight at a
  // time (a chunk at a time is one dependent ~0.3 us access after the other: 16 chunks took ~9 us)
  const float* base = a.part + ((int64_t)h * a.chunks) * n_rep * (hd + 2);
  for (int e = tid; e < n_rep * hd; e += 256) {
    const int j = e / hd, d = e % hd;
    const float* pj = base + (int64_t)j * (hd + 2);
    const int64_t cstride = (int64_t)n_rep * (hd + 2);
    float M = -3.0e38f;
    for (int c0 = 0; c0 < n_act; c0 += 8) {
      float mv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) mv[q] = ld_agent(pj + (int64_t)min(c0 + q, n_act - 1) * cstride + hd);
#pragma unroll
      for (int q = 0; q < 8; ++q) M = fmaxf(M, mv[q]);
    }
    float num = 0.f, den = 0.f;
    for (int c0 = 0; c0 < n_act; c0 += 8) {
      float mv[8], lv[8], ov[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float* pc = pj + (int64_t)min(c0 + q, n_act - 1) * cstride;
        mv[q] = ld_agent(pc + hd);
        lv[q] = ld_agent(pc + hd + 1);
        ov[q] = ld_agent(pc + d);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float w = (c0 + q < n_act) ? __builtin_amdgcn_exp2f((mv[q] - M) * kLog2e) : 0.f;
        num = __builtin_fmaf(w, ov[q], num);
        den = __builtin_fmaf(w, lv[q], den);
      }
    }
    a.out[((int64_t)h * n_rep + j) * hd + d] = A::from_f32(num / den);
  }
  }
  };   // rest
  if constexpr (CH == 0) {
    if (pos < 256) rest(std::integral_constant<int, 64>{});
    else rest(std::integral_constant<int, 128>{});
  } else {
    rest(std::integral_constant<int, CH>{});
  }
}

template <typename AT, int HD, int NREP, int CH, bool PARTS, bool SPLIT = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs a) {
  attn_decode_body<AT, HD, NREP, CH, PARTS, SPLIT, false>(a, (int)blockIdx.x, (int)blockIdx.y, 0u);
}

}  // namespace paro

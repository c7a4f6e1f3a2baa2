// Fused pairwise-rotation + INT4 dequant + GEMV/small-batch GEMM for gfx950 (decode path).
//
// One launch computes  y[b, :] = rotate_p(x[b, :] * cs_p) @ dequant(W)  for every merged
// partition p (qkv = 3 rotations, gate_up = 2) -- the work the reference spreads over
// `rotate` + AWQ/Marlin GEMM launches per partition + torch.cat + bias
// (transformers/modules.py:57-71, vllm/plugin.py:281-311).
//
// Mapping (CDNA4-first, not a warp-tiled port):
//   * work unit = (128-channel quantisation group g) x (TPW column tiles of 16 outputs).
//     A 64-lane wavefront owns one unit at a time: 64 lanes == the 64 Givens pairs of the
//     group, so the wave rotates ITS OWN slice of x in wave-private LDS with no workgroup
//     barrier.  All 8 stages' coefficients arrive as two 16-byte loads per lane
//     (paro_pack_rotation), requested BEFORE the unit's INT4 tiles so that waiting for them
//     does not wait for the tiles (vmcnt retires in order); the rotation then runs while
//     the tiles (non-temporal, 1 KiB per wave-load, straight to VGPRs) are in flight.
//   * the group loop is software-pipelined: unit n+1's coefficients and tiles are requested
//     before unit n's tiles are consumed.
//   * the INT4 tile is stored in MFMA B-fragment order (paro_repack_awq), so a lane's
//     16-byte load IS its four v_mfma_f32_16x16x32 B operands after a shift/and/or unpack
//     to (16 + q) halves; scale and zero point come as one packed word per (group, column)
//     and are applied on the fp32 MFMA result:  acc += s * (D - (16 + z) * sum_k x_k),
//     with sum_k x_k from one extra MFMA against a ones fragment.
//   * batch rows (<= 16) ride in the MFMA M dimension: rows <= 4 occupy MFMA rows 0,4,8,12
//     so a single accumulator register per tile suffices.
//   * the WAVES (4..16) waves of a workgroup take different groups of the same columns and
//     reduce through LDS; by default one workgroup covers ALL of K, so no cross-workgroup
//     reduction exists.  An optional K-split (grid.y) is combined in-launch with data-tagged
//     granules: the first ksplit-1 splits write {tag, partial} 8-byte granules with ONE
//     write-through (sc1) store per output and exit; the last split polls them (sc1 loads, bounded
//     spin), re-arms them, and writes y exactly once -- no fences, no tickets, one round trip.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace paro {

struct GemvArgs {
  const u32x4* wq;
  const unsigned* sz;
  const unsigned* rot;
  const unsigned short* cs;
  const unsigned short* bias;
  const unsigned short* x;  // [rows][K], or pre-rotated [nparts][rows][K] when PREROT
  unsigned short* y;
  unsigned long long* slabs;  // K-split granules {tag << 32 | fp32 bits}: [ksplit - 1][rows][N]
  unsigned* counters;
  int K, N, G, rows, krot, ksplit, gps;  // gps = groups per K-split
  int tstride, gstride;                  // 1-KiB chunk index of tile (t, g) = t * tstride + g * gstride
  int flags;                             // debug (PARO_GEMV_FLAGS): 16 = return at kernel entry (launch floor)
  int pd;                                // software-pipeline distance in units (1 or 2)
  PartTable pt;
};

constexpr int kXhStride = 136;  // halves per fragment row in LDS (128 + 8 pad: 16 rows' b128 reads spread over banks)

template <typename AT, int TPW, int MB, int WAVES, bool PREROT, int PD>  // PD = prefetch distance in units (1 or 2)
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(const GemvArgs a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int PDIST = PD % 10;   // pipeline distance variant
  constexpr int DIAG = PD / 10;    // diagnostics (tools/ablate): 1 = no rotation stages, 2 = also no unpack/MFMA (pure stream)
  constexpr int MR = MB <= 4 ? 1 : (MB <= 8 ? 2 : 4);  // accumulator registers kept per tile
  constexpr int VW = MB >= 4 ? 4 : MB;                  // LDS vector width of the rotation state
  constexpr int NCH = MB / VW;
  typedef float V __attribute__((ext_vector_type(VW)));
  constexpr int XR_FLOATS = PREROT ? 0 : MB * 128;
  constexpr int XH_HALVES = PREROT ? 0 : (MB + 1) * kXhStride;  // + one all-zero row for unused MFMA rows
  constexpr int REGION_BYTES = XR_FLOATS * 4 + ((XH_HALVES * 2 + 15) / 16) * 16;  // rotation state + fragment rows of one unit
  constexpr int WAVE_BYTES = (PDIST == 4 ? 2 : 1) * REGION_BYTES;
  constexpr int RED_FLOATS = WAVES * TPW * MR * 64;
  constexpr int LDS_BYTES = (WAVES * WAVE_BYTES > RED_FLOATS * 4 ? WAVES * WAVE_BYTES : RED_FLOATS * 4) + 16;
  constexpr int NSZ = TPW <= 4 ? 1 : TPW / 4;  // 16-byte scale/zero vectors per unit
  constexpr int SZW = TPW < 4 ? TPW : 4;
  typedef unsigned SZV __attribute__((ext_vector_type(SZW)));
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.x, ks = blockIdx.y;
  if (a.flags & 16) return;  // ablation: launch floor

  const int p = a.pt.part_of_cb(cb);
  const int ltile0 = (cb - a.pt.cb_start[p]) * TPW;
  const int tile0 = a.pt.tile_start[p] + ltile0;
  const int nt = min(TPW, a.pt.tile_start[p + 1] - tile0);
  const int ts0 = a.pt.szt_start[p] + ltile0;
  const int g_begin = ks * a.gps;
  const int g_end = min(a.G, g_begin + a.gps);

  float* xr = (float*)(lds + wave * WAVE_BYTES);
  unsigned short* xh = (unsigned short*)(lds + wave * WAVE_BYTES + XR_FLOATS * 4);
  if constexpr (!PREROT) {
    for (int c = lane; c < kXhStride; c += 64) xh[MB * kXhStride + c] = 0;  // zero row
    if constexpr (PDIST == 4)
      for (int c = lane; c < kXhStride; c += 64) xh[REGION_BYTES / 2 + MB * kXhStride + c] = 0;
  }

  const int n = lane & 15, mq = lane >> 4;
  // A-fragment source row of this lane: MFMA row m' = lane & 15 carries batch row (m'>>2)*MR + (m'&3)
  const int mrow = lane & 15;
  const int brow = (mrow >> 2) * MR + (mrow & 3);
  const bool avalid = ((mrow & 3) < MR) && (brow < a.rows);

  float acc[TPW][MR];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[j][r] = 0.f;

  struct PBuf {
    unsigned xv[PREROT ? 1 : MB];
    unsigned csv;
    u32x4 r0, r1;
    u32x4 xa[PREROT ? 4 : 1];
  };
  struct TBuf {
    u32x4 q[TPW];
    SZV sz[NSZ];
  };

  const unsigned short* xrot_p = a.x + (PREROT ? (int64_t)p * a.rows * a.K : 0);
  const int64_t szrow = (int64_t)(a.pt.tsz >> 2) * 64;  // words per group row of the scale/zero array

  auto load_p = [&](PBuf& b, int g) {
    if constexpr (PREROT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b.xa[i] = (u32x4){0u, 0u, 0u, 0u};
        if (avalid) b.xa[i] = *(const u32x4*)(xrot_p + (int64_t)brow * a.K + g * 128 + 32 * i + 8 * mq);
      }
    } else {
      const u32x4* rp = (const u32x4*)(a.rot + (((int64_t)p * a.G + g) * 64 + lane) * 8);
      b.r0 = rp[0];
      b.r1 = rp[1];
      b.csv = *(const unsigned*)(a.cs + (int64_t)p * a.K + g * 128 + 2 * lane);
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int rr = r < a.rows ? r : 0;  // clamp instead of branching: keeps the load count static
        b.xv[r] = *(const unsigned*)(a.x + (int64_t)rr * a.K + g * 128 + 2 * lane);
      }
    }
  };
  // Every load below is unconditional (ragged column blocks re-read their last tile and mask it
  // later) so that the compiler's vmcnt bookkeeping is exact: the wait in front of the rotation
  // covers only the coefficient loads and leaves the tile loads in flight.
  auto load_t = [&](TBuf& b, int g) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int jj = j < nt ? j : nt - 1;
      b.q[j] = __builtin_nontemporal_load(a.wq + ((int64_t)(tile0 + jj) * a.tstride + (int64_t)g * a.gstride) * 64 + lane);
    }
    const unsigned* sp = a.sz + (int64_t)g * szrow + ((int64_t)(ts0 >> 2) * 16 + n) * 4 + (ts0 & 3);
#pragma unroll
    for (int v = 0; v < NSZ; ++v) b.sz[v] = *(const SZV*)(sp + v * 64);
  };

  PBuf pc, pn, pn2;
  TBuf tc, tn, tn2;

  // One work unit: rotate group g's slice of x (coefficients in pc), then consume its tiles (tc).
  // Software pipeline.  PFP: request coefficients of unit gp (before this unit's rotation); PFT:
  // request tiles of unit gt (before this unit's tiles are consumed).  Coefficients are always
  // requested before the tiles of the same unit (in-order vmcnt: a wait for coefficients never waits
  // for younger tile loads).  PD selects the distances (measured, MI355X, same box, Llama-3-8B shapes):
  //   PD 1: coefficients +1, tiles +1            gate_up 15.5 us  o_proj 6.9 us
  //   PD 2: coefficients +2, tiles +2            gate_up 16.9 us  o_proj 7.8 us  (more bulk loads in
  //         flight only lengthen the queue the small coefficient loads wait in)
  //   PD 3: coefficients +2, tiles +1
  auto step = [&](auto pfp_tag, auto pft_tag, int gp, int gt) {
    constexpr bool PFP = decltype(pfp_tag)::value;
    constexpr bool PFT = decltype(pft_tag)::value;
    if constexpr (PFP) load_p(pn2, gp);

    // ---- A fragments of this group (4 x K=32)
    vec8 af[4];
    if constexpr (PREROT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(vec8, pc.xa[i]);
    } else {
      const float c0 = f16_bits_to_f32(pc.csv & 0xffffu), c1 = f16_bits_to_f32(pc.csv >> 16);
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int ch = r / VW, v = r % VW;
        const unsigned xv = r < a.rows ? pc.xv[r] : 0u;
        xr[(ch * 128 + 2 * lane) * VW + v] = A::to_f32(xv & 0xffffu) * c0;
        xr[(ch * 128 + 2 * lane + 1) * VW + v] = A::to_f32(xv >> 16) * c1;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (DIAG == 0 && r < a.krot) {
          const unsigned w = r < 4 ? pc.r0[r & 3] : pc.r1[r & 3];
          const int i = (int)(w & 0xffu), j = (int)((w >> 8) & 0xffu);
          float s, c;
          fast_sincos(f16_bits_to_f32(w >> 16), s, c);
          V va[NCH], vb[NCH];
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            va[ch] = *(const V*)(xr + (ch * 128 + i) * VW);
            vb[ch] = *(const V*)(xr + (ch * 128 + j) * VW);
          }
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            *(V*)(xr + (ch * 128 + i) * VW) = va[ch] * c + vb[ch] * s;
            *(V*)(xr + (ch * 128 + j) * VW) = vb[ch] * c - va[ch] * s;
          }
          // a wave's DS operations execute in issue order, so the next stage's reads see these
          // writes; the barrier only stops the compiler from reordering across stages
          __builtin_amdgcn_wave_barrier();
        }
      }
      // fp32 state -> activation-dtype fragment rows (one rounding), lane l converts channels 2l, 2l+1
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int ch = r / VW, v = r % VW;
        const float v0 = xr[(ch * 128 + 2 * lane) * VW + v];
        const float v1 = xr[(ch * 128 + 2 * lane + 1) * VW + v];
        *(unsigned*)(xh + r * kXhStride + 2 * lane) = (unsigned)A::from_f32(v0) | ((unsigned)A::from_f32(v1) << 16);
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned short* afrag = xh + (avalid ? brow : MB) * kXhStride + 8 * mq;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const vec8*)(afrag + 32 * i);
    }
    // sx = sum_k x_k and so = sum_k x_k off_k (off_k = the per-element offset unpack_fast leaves in)
    f32x4 sx = {0.f, 0.f, 0.f, 0.f}, so = {0.f, 0.f, 0.f, 0.f};
    {
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const u32x4 offs = {A::kOffFrag0, A::kOffFrag1, A::kOffFrag0, A::kOffFrag1};
      const vec8 ob = __builtin_bit_cast(vec8, ones);
      const vec8 fb = __builtin_bit_cast(vec8, offs);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sx = A::mfma(af[i], ob, sx);
        so = A::mfma(af[i], fb, so);
      }
    }
    if constexpr (PFT) load_t(tn2, gt);

    // ---- per tile: unpack -> 4 MFMA -> scale / zero point on the fp32 result
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      if constexpr (DIAG == 2) {
        d[0] = __builtin_bit_cast(float, (tc.q[j][0] ^ tc.q[j][1] ^ tc.q[j][2] ^ tc.q[j][3]) & 0x3fffffffu);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned w4[4];
          A::unpack_fast(tc.q[j][i], w4);
          const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
          d = A::mfma(af[i], __builtin_bit_cast(vec8, wv), d);
        }
      }
      const unsigned szw = tc.sz[j / 4][j % 4];
      const float s = f16_bits_to_f32(szw & 0xffffu);
      const float zf = f16_bits_to_f32(szw >> 16);
#pragma unroll
      for (int r = 0; r < MR; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-zf, sx[r], d[r] - so[r]), acc[j][r]);
    }
    if constexpr (PDIST == 1) {
      if constexpr (PFP) pc = pn2;
      if constexpr (PFT) tc = tn2;
    } else {
      pc = pn;
      if constexpr (PFP) pn = pn2;
      if constexpr (PDIST == 2) {
        tc = tn;
        if constexpr (PFT) tn = tn2;
      } else if constexpr (PFT) {
        tc = tn2;
      }
    }
  };

  // ---- PDIST 4: units are processed in PAIRS whose rotations are interleaved stage by stage.  With 8-wave
  // workgroups the 8-stage rotation is a dependent LDS chain (~150-200 cycles per stage) rather than LDS
  // throughput, so two chains in flight per wave take the time of one.
  if constexpr (PDIST == 4 && !PREROT) {
    float* xrB = (float*)(lds + wave * WAVE_BYTES + REGION_BYTES);
    unsigned short* xhB = (unsigned short*)(lds + wave * WAVE_BYTES + REGION_BYTES + XR_FLOATS * 4);
    auto seed = [&](float* xs, const PBuf& pb) {
      const float c0 = f16_bits_to_f32(pb.csv & 0xffffu), c1 = f16_bits_to_f32(pb.csv >> 16);
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int ch = r / VW, v = r % VW;
        const unsigned xv = r < a.rows ? pb.xv[r] : 0u;
        xs[(ch * 128 + 2 * lane) * VW + v] = A::to_f32(xv & 0xffffu) * c0;
        xs[(ch * 128 + 2 * lane + 1) * VW + v] = A::to_f32(xv >> 16) * c1;
      }
    };
    auto stage = [&](float* xs, unsigned w) {
      const int i = (int)(w & 0xffu), j = (int)((w >> 8) & 0xffu);
      float s, c;
      fast_sincos(f16_bits_to_f32(w >> 16), s, c);
      V va[NCH], vb[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        va[ch] = *(const V*)(xs + (ch * 128 + i) * VW);
        vb[ch] = *(const V*)(xs + (ch * 128 + j) * VW);
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        *(V*)(xs + (ch * 128 + i) * VW) = va[ch] * c + vb[ch] * s;
        *(V*)(xs + (ch * 128 + j) * VW) = vb[ch] * c - va[ch] * s;
      }
    };
    auto finish = [&](const float* xs, unsigned short* hs, vec8 (&af)[4], f32x4& sx, f32x4& so) {
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int ch = r / VW, v = r % VW;
        const float v0 = xs[(ch * 128 + 2 * lane) * VW + v];
        const float v1 = xs[(ch * 128 + 2 * lane + 1) * VW + v];
        *(unsigned*)(hs + r * kXhStride + 2 * lane) = (unsigned)A::from_f32(v0) | ((unsigned)A::from_f32(v1) << 16);
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned short* afrag = hs + (avalid ? brow : MB) * kXhStride + 8 * mq;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const vec8*)(afrag + 32 * i);
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const u32x4 offs = {A::kOffFrag0, A::kOffFrag1, A::kOffFrag0, A::kOffFrag1};
      sx = (f32x4){0.f, 0.f, 0.f, 0.f};
      so = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sx = A::mfma(af[i], __builtin_bit_cast(vec8, ones), sx);
        so = A::mfma(af[i], __builtin_bit_cast(vec8, offs), so);
      }
    };
    auto consume = [&](const TBuf& t, const vec8 (&af)[4], const f32x4& sx, const f32x4& so) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned w4[4];
          A::unpack_fast(t.q[j][i], w4);
          const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
          d = A::mfma(af[i], __builtin_bit_cast(vec8, wv), d);
        }
        const unsigned szw = t.sz[j / 4][j % 4];
        const float s = f16_bits_to_f32(szw & 0xffffu);
        const float zf = f16_bits_to_f32(szw >> 16);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-zf, sx[r], d[r] - so[r]), acc[j][r]);
      }
    };
    PBuf pA, pB, pA2, pB2;
    TBuf tA, tB, tA2, tB2;
    // one pair; HASB: the pair has a second unit; NEXT: 0 = nothing follows, 1 = one more unit, 2 = a full pair
    auto pair = [&](auto hasb_tag, auto next_tag, int gnA, int gnB) {
      constexpr bool HASB = decltype(hasb_tag)::value;
      constexpr int NEXT = decltype(next_tag)::value;
      if constexpr (NEXT >= 1) load_p(pA2, gnA);
      if constexpr (NEXT >= 2) load_p(pB2, gnB);
      seed(xr, pA);
      if constexpr (HASB) seed(xrB, pB);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < a.krot) {
          stage(xr, r < 4 ? pA.r0[r & 3] : pA.r1[r & 3]);
          if constexpr (HASB) stage(xrB, r < 4 ? pB.r0[r & 3] : pB.r1[r & 3]);
          __builtin_amdgcn_wave_barrier();
        }
      }
      vec8 afA[4], afB[4];
      f32x4 sxA, soA, sxB, soB;
      finish(xr, xh, afA, sxA, soA);
      if constexpr (HASB) finish(xrB, xhB, afB, sxB, soB);
      if constexpr (NEXT >= 1) load_t(tA2, gnA);
      consume(tA, afA, sxA, soA);
      if constexpr (NEXT >= 2) load_t(tB2, gnB);
      if constexpr (HASB) consume(tB, afB, sxB, soB);
      if constexpr (NEXT >= 1) {
        pA = pA2;
        tA = tA2;
      }
      if constexpr (NEXT >= 2) {
        pB = pB2;
        tB = tB2;
      }
    };
    const int g0 = g_begin + wave;
    const std::true_type yes{};
    const std::false_type no{};
    const std::integral_constant<int, 0> n0{};
    const std::integral_constant<int, 1> n1{};
    const std::integral_constant<int, 2> n2{};
    if (g0 + WAVES < g_end) {        // at least one full pair
      load_p(pA, g0);
      load_p(pB, g0 + WAVES);
      load_t(tA, g0);
      load_t(tB, g0 + WAVES);
      int g = g0;
      for (; g + 3 * WAVES < g_end; g += 2 * WAVES) pair(yes, n2, g + 2 * WAVES, g + 3 * WAVES);
      if (g + 2 * WAVES < g_end) {   // a single unit follows the current pair
        pair(yes, n1, g + 2 * WAVES, 0);
        pair(no, n0, 0, 0);
      } else {
        pair(yes, n0, 0, 0);
      }
    } else if (g0 < g_end) {
      load_p(pA, g0);
      load_t(tA, g0);
      pair(no, n0, 0, 0);
    }
  } else
  {
    const int g0 = g_begin + wave;
    const std::true_type yes{};
    const std::false_type no{};
    if (PDIST == 2 && g0 + WAVES < g_end) {  // >= 2 units: both units' coefficients, then both units' tiles, up front
      load_p(pc, g0);
      load_p(pn, g0 + WAVES);
      load_t(tc, g0);
      load_t(tn, g0 + WAVES);
      for (int g = g0; g + 2 * WAVES < g_end; g += WAVES) step(yes, yes, g + 2 * WAVES, g + 2 * WAVES);
      step(no, no, 0, 0);
      step(no, no, 0, 0);
    } else if (PDIST == 3 && g0 + WAVES < g_end) {
      load_p(pc, g0);
      load_p(pn, g0 + WAVES);
      load_t(tc, g0);
      int g = g0;
      for (; g + 2 * WAVES < g_end; g += WAVES) step(yes, yes, g + 2 * WAVES, g + WAVES);
      step(no, yes, 0, g + WAVES);
      step(no, no, 0, 0);
    } else if (PDIST == 1 && g0 < g_end) {  // distance 1: the next unit is requested while this one is processed
      load_p(pc, g0);
      load_t(tc, g0);
      int g = g0;
      for (; g + WAVES < g_end; g += WAVES) step(yes, yes, g + WAVES, g + WAVES);
      step(no, no, 0, 0);
    } else if (g0 < g_end) {
      load_p(pc, g0);
      load_t(tc, g0);
      step(no, no, 0, 0);
    }
  }

  // ---- reduce the workgroup's waves (different groups, same columns) through LDS
  __syncthreads();
  float* red = (float*)lds;
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) red[((wave * TPW + j) * MR + r) * 64 + lane] = acc[j][r];
  __syncthreads();

  const bool direct = (a.ksplit == 1);
  for (int e = tid; e < TPW * MR * 64; e += WAVES * 64) {
    const int el = e & 63, r = (e >> 6) % MR, j = e / (MR * 64);
    const int b = (el >> 4) * MR + r;
    if (j >= nt || b >= a.rows) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[e + w * TPW * MR * 64];
    const int col = (tile0 + j) * 16 + (el & 15);
    if (direct) {
      if (a.bias) v += A::to_f32(a.bias[col]);
      a.y[(int64_t)b * a.N + col] = A::from_f32(v);
    } else if (ks != a.ksplit - 1) {
      // producer: ONE 8-byte {tag = 1, fp32 partial} granule per output, written through (sc1); no
      // drain, no flag, no fence -- the data IS the flag (cdna guide G16 recipe R2); then exit.
      const unsigned long long gv = (1ull << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
      __hip_atomic_store(a.slabs + ((int64_t)ks * a.rows + b) * a.N + col, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // reducer (the last K-split of this column block; dispatched after the others): keep the own
      // partial in registers, poll the other splits' granules until their tags appear (bounded),
      // re-arm them to zero for the next launch, write y once.
      for (int s = 0; s < a.ksplit - 1; ++s) {
        unsigned long long* gp = a.slabs + ((int64_t)s * a.rows + b) * a.N + col;
        unsigned long long gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; (gv >> 32) != 1ull && spin < (1 << 17); ++spin) {
          __builtin_amdgcn_s_sleep(2);
          gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((gv >> 32) != 1ull) a.counters[PARO_WS_COUNTER_BYTES / 4 - 1] = 0xDEADu;  // give-up code, checked by tests
        v += __builtin_bit_cast(float, (unsigned)gv);
        __hip_atomic_store(gp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (a.bias) v += A::to_f32(a.bias[col]);
      a.y[(int64_t)b * a.N + col] = A::from_f32(v);
    }
  }
}

// ---- per-translation-unit launch tables (one TU per activation type x PREROT, built in parallel)
template <typename AT, int TPW, int MB, bool PREROT, int PD>
int launch_waves_pd(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if constexpr (MB <= 4 && TPW <= 4) {
    if (waves == 16) {
      hipLaunchKernelGGL((gemv_kernel<AT, TPW, MB, 16, PREROT, PD>), grid, dim3(1024), 0, st, a);
      return PARO_OK;
    }
  }
  if constexpr (MB <= 4 && !PREROT) {  // odd group counts (K = 2560 -> 20 groups, K = 9728 -> 76): 5 / 10 waves divide them evenly
    if (waves == 10 && TPW <= 4) {
      hipLaunchKernelGGL((gemv_kernel<AT, TPW, MB, 10, PREROT, PD>), grid, dim3(640), 0, st, a);
      return PARO_OK;
    }
    if (waves == 5) {
      hipLaunchKernelGGL((gemv_kernel<AT, TPW, MB, 5, PREROT, PD>), grid, dim3(320), 0, st, a);
      return PARO_OK;
    }
  }
  if (waves == 8) {
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, MB, 8, PREROT, PD>), grid, dim3(512), 0, st, a);
    return PARO_OK;
  }
  if (waves == 4) {
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, MB, 4, PREROT, PD>), grid, dim3(256), 0, st, a);
    return PARO_OK;
  }
  return fail(PARO_ERR_UNSUPPORTED, "waves per workgroup = %d not built for %d batch rows", waves, MB);
}

template <typename AT, int TPW, int MB, bool PREROT>
int launch_waves(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  // PD 2 / 3 (deeper prefetch) measured slower on MI355X (see the table above); only PD 1 is built,
  // plus two diagnostic variants of the M = 1 kernel (PARO_GEMV_PD = 11 / 12, tools/ablate_gemv.py).
  // PD 4 (pairs of units with interleaved rotation chains) is correct but measured slower than PD 1 on
  // every Llama-3-8B / Qwen3-4B shape (down_proj 12.2 vs 10.9 us, qkv 8.5 vs 7.7 us): the staggered issue
  // of PD 1 (tiles of unit n+1 requested only after unit n's rotation) is what overlaps rotation with the
  // HBM burst.  Not instantiated; build with -DPARO_GEMV_PAIRED to A/B it again.
#ifdef PARO_GEMV_PAIRED
  if constexpr (MB <= 4 && !PREROT && TPW <= 4) {
    if (a.pd == 4) return launch_waves_pd<AT, TPW, MB, PREROT, 4>(a, waves, grid, st);
  }
#endif
  if constexpr (MB == 1 && !PREROT) {
    if (a.pd == 11) return launch_waves_pd<AT, TPW, MB, PREROT, 11>(a, waves, grid, st);
    if (a.pd == 12) return launch_waves_pd<AT, TPW, MB, PREROT, 12>(a, waves, grid, st);
  }
  return launch_waves_pd<AT, TPW, MB, PREROT, 1>(a, waves, grid, st);
}

template <typename AT, int TPW, bool PREROT>
int launch_rows(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if (a.rows <= 1) return launch_waves<AT, TPW, 1, PREROT>(a, waves, grid, st);
  if (a.rows <= 4) return launch_waves<AT, TPW, 4, PREROT>(a, waves, grid, st);
  if (a.rows <= 8) return launch_waves<AT, TPW, 8, PREROT>(a, waves, grid, st);
  if constexpr (TPW <= 4) return launch_waves<AT, TPW, 16, PREROT>(a, waves, grid, st);
  return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave = 8 is not built for more than 8 batch rows");
}

template <typename AT, bool PREROT>
int launch_gemv_variant(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st) {
  switch (tpw) {
    case 1: return launch_rows<AT, 1, PREROT>(a, waves, grid, st);
    case 2: return launch_rows<AT, 2, PREROT>(a, waves, grid, st);
    case 4: return launch_rows<AT, 4, PREROT>(a, waves, grid, st);
    case 8: return launch_rows<AT, 8, PREROT>(a, waves, grid, st);
  }
  return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave must be 1, 2, 4 or 8 (got %d)", tpw);
}

// defined in gemv_f16.hip / gemv_f16_pre.hip / gemv_bf16.hip / gemv_bf16_pre.hip
int launch_gemv_f16(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st);
int launch_gemv_f16_pre(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st);
int launch_gemv_bf16(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st);
int launch_gemv_bf16_pre(const GemvArgs& a, int tpw, int waves, dim3 grid, hipStream_t st);

}  // namespace paro

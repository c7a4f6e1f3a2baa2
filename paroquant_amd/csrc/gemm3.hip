// W4A16 MFMA GEMM, variant 4 (prefill, >= 256 rows):  y = x_rot @ dequant(W) (+ bias), f16 and native bf16.
//
// 256 x 256 output tile per 512-thread workgroup; the 8 waves sit side by side along N (1 x 8): wave w owns the
// 32 columns [32 w, 32 w + 32) over ALL 256 rows, so every INT4 word of the workgroup's weight tile is loaded
// and dequantised by exactly one wave (the 2 x 4 arrangement of variant 3 does that work twice, in both row
// waves: its VALU stream -- 2 ops per MFMA -- is what keeps a wave from issuing an MFMA every 16 cycles).
//   * v_mfma_f32_32x32x16: the WEIGHTS are the MFMA "A" operand (row i = output column n, lane l holds
//     n = l & 31, k = 8 (l >> 5) .. + 7 -- one dequantised INT4 word), the ACTIVATIONS the "B" operand (column
//     j = row m of x).  One MFMA = 32 cycles of the SIMD's matrix pipe; per MFMA the wave issues one
//     ds_read_b128 (the fragment it has just consumed, re-filled for the next k-step) and <= 3 dequant ops,
//     so a wave ALONE on its SIMD still issues back to back (the two waves of a SIMD drift apart, the older
//     one wins arbitration and waits at the barrier -- the other must not slow down while it runs alone).
//   * accumulator D[i][j]: lane l holds column j = l & 31 (= row m) and rows i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
//     (= output column): four consecutive output columns per register quad -> 8-byte stores, 32 per lane.
//   * A side (activations): the 256-row x 128-k slab of the current quantisation group goes global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4), double buffered (2 x 64 KiB), 16-slot XOR swizzle on the SOURCE
//     address (the LDS image of an LDS-DMA is lane-linear), conflict-free ds_read_b128.  The next group's
//     eight DMA pieces of a wave are issued two per k-step, between the MFMAs of the first four k-steps.
//   * B side (weights): two 16-byte loads per lane per group straight from the packed tiles
//     (paro_repack_awq: tile (t, g) = 1 KiB, lane' = (kb, n) owns words i = 0..3 with k = 32 i + 8 kb + e):
//     lane (n32, kh) takes lane' (kh, n) and (2 + kh, n) of tile n32 >> 4 -- word i of the first is k-step 2 i,
//     of the second k-step 2 i + 1.  fp16: exact (q - z) * s in packed fp16 (13 ops per word), bit-for-bit the
//     reference's fp16 dequant.  bf16: q -> fp32 (v_cvt_f32_ubyte), fma with (s, -z s) -- the exact product --
//     one rounding to bf16 (v_cvt_pk_bf16_f32): 23 ops per word, still < 3 per MFMA.
#include <type_traits>

#include "common.hpp"
#include "gemm_args.hpp"

namespace paro {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <typename AT>
struct Mma32;
template <>
struct Mma32<f16> {
  __device__ static __forceinline__ f32x16 run(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma32<bf16> {
  __device__ static __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// Per-(group, column) dequant state and the slice-wise dequant of one INT4 word (8 weights of one column,
// consecutive k) into an MFMA operand; `part(r, w)` is slot r of 8 -- one slot follows each MFMA of a k-step.
template <typename AT>
struct Dequant;

template <>
struct Dequant<f16> {
  Act<f16>::Unpack u;
  f16x2 s2, c_hi, c_lo;
  unsigned t, o0, o1, o2, o3;
  f16x2 a0, a1, a2, a3;
  u32x4 out;
  __device__ __forceinline__ void init() { u = Act<f16>::unpack_consts(); }
  __device__ __forceinline__ void set_group(unsigned szw) {
    const f16 sh = __builtin_bit_cast(f16, (unsigned short)(szw & 0xffffu));
    const f16 zh = __builtin_bit_cast(f16, (unsigned short)(szw >> 16));
    s2 = (f16x2){sh, sh};
    const f16 ch = (f16)(-1024.f) - zh, cl = (f16)(-64.f) - zh;   // exact: integers of magnitude <= 1039
    c_hi = (f16x2){ch, ch};
    c_lo = (f16x2){cl, cl};
  }
  __device__ __forceinline__ void part(int r, unsigned w) {
    if (r == 0) {
      t = w >> 8;
      o0 = (w & u.m0) | u.k0;
    } else if (r == 1) {
      o1 = (w & u.m1) | u.k1;
      o2 = (t & u.m0) | u.k0;
    } else if (r == 2) {
      o3 = (t & u.m1) | u.k1;
      a0 = __builtin_bit_cast(f16x2, o0) + c_hi;
    } else if (r == 3) {
      out[0] = __builtin_bit_cast(unsigned, a0 * s2);
      a1 = __builtin_bit_cast(f16x2, o1) + c_lo;
    } else if (r == 4) {
      out[1] = __builtin_bit_cast(unsigned, a1 * s2);
      a2 = __builtin_bit_cast(f16x2, o2) + c_hi;
    } else if (r == 5) {
      out[2] = __builtin_bit_cast(unsigned, a2 * s2);
      a3 = __builtin_bit_cast(f16x2, o3) + c_lo;
    } else if (r == 6) {
      out[3] = __builtin_bit_cast(unsigned, a3 * s2);
    }
  }
};

template <>
struct Dequant<bf16> {
  float s, nzs;          // scale, -(zero * scale): both exact in fp32
  unsigned t0, t1;       // nibbles 0,2,4,6 / 1,3,5,7 of the word, one per byte
  float f0, f1;
  u32x4 out;
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void set_group(unsigned szw) {
    s = f16_bits_to_f32(szw & 0xffffu);
    nzs = -(f16_bits_to_f32(szw >> 16) * s);
  }
  // element e of the word sits in nibble (e >> 1) + 4 (e & 1): pair v = (e 2v, e 2v+1) = bytes (v >> 1, (v >> 1) + 2)
  // of t0 (v even) / t1 (v odd)
  __device__ __forceinline__ float elem(unsigned tt, int byte) const {
    return __builtin_fmaf((float)((tt >> (8 * byte)) & 0xffu), s, nzs);
  }
  __device__ __forceinline__ unsigned pack(float lo, float hi) const {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  }
  __device__ __forceinline__ void part(int r, unsigned w) {
    if (r == 0) {
      t0 = w & 0x0F0F0F0Fu;
      t1 = (w >> 4) & 0x0F0F0F0Fu;
    } else if (r == 1) {
      f0 = elem(t0, 0);
      f1 = elem(t0, 2);
    } else if (r == 2) {
      out[0] = pack(f0, f1);
      f0 = elem(t1, 0);
    } else if (r == 3) {
      f1 = elem(t1, 2);
      out[1] = pack(f0, f1);
    } else if (r == 4) {
      f0 = elem(t0, 1);
      f1 = elem(t0, 3);
    } else if (r == 5) {
      out[2] = pack(f0, f1);
      f0 = elem(t1, 1);
    } else if (r == 6) {
      f1 = elem(t1, 3);
      out[3] = pack(f0, f1);
    }
  }
};


// DIAG (energy / issue ablation on the power-limited chip, tools/bench_gemm.py --variants 41,42,43; wrong results):
// 1 = no dequant (raw INT4 words as the weight operand), 2 = no fragment re-reads inside a group (the first
// k-step's fragments are reused), 3 = both.  0 = the shipping kernel.
// DIAG 4 (variant 44; CORRECT results, bit-identical to pre-pass + variant 4): the north star's FUSED form -- "stages tiles in LDS, applies
// the pairwise rotation ..., then feeds MFMA" (BASELINE.json) -- built to be MEASURED against the two launches that
// ship.  The LDS-DMA slab of group g lands UN-rotated; wave w rotates row tile w of it in place as one dense product per group on the
// matrix cores, x_rot = x R'^T with the matrices of the pre-pass (rmat[p][g][n][k]: 4 channel tiles x 8 k-steps = 32 MFMAs of 32x32x16
// per wave and group beside the 64 of the GEMM itself -- the rotation is per (partition, group), and inside the GEMM it is repeated in
// every 256-column block), rounds once to the activation type as the pre-pass does, writes it back through the slab's swizzle; one more
// barrier; then the unchanged k-loop.  No rotated copy of x in HBM, one launch -- and 2.5 .. 3.4x the time of the two launches at
// M = 65536 (a second cut with the matrix brought by LDS-DMA into the CU's last 32 KB of LDS and two interleaved MFMA chains: 3.3 ..
// 3.7x): profiles/NOTES.md 6.11, profiles/r06_fused_rot_gemm*.  Kept as an experiment variant, never selected.
// QS: quantisation groups per 128-channel slab (1: group_size 128; 2: group_size 64 -- k-steps 0..3 and 4..7 of a
// slab are dequantised with different (scale, zero) words).
// RT: 32-row tiles of the workgroup's row block (8 = 256 rows, the prefill shape; 1 / 2 / 4 = 32 / 64 / 128 rows for
// batched decode and short prefill, where a 256-row block would stage and multiply mostly padding).  With a K-split
// (grid.z) the fp32 tile goes to `partial` and gemm_reduce_kernel sums the splits.
template <typename AT, int DIAG = 0, int QS = 1, int RT = 8>
__global__ __launch_bounds__(512) void gemm3_kernel(const GemmArgs a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int BMR = RT * 32;                 // rows of the workgroup's block
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BMR * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.x + a.cb0;
  const int row0 = blockIdx.y * BMR;

  const u32x4* wq_base = a.wq;
  const unsigned* sz_base = a.sz;
  if (a.block_expert) {   // grouped launch: this row block belongs to one expert (or to the padding behind the last one)
    const int e = a.block_expert[blockIdx.y];
    if (e < 0 || e >= a.n_experts) return;   // unused block, or an id outside the expert table (never an out-of-bounds weight read)
    wq_base += (int64_t)e * a.wq_estride;
    sz_base += (int64_t)e * a.sz_estride;
  }
  const int p = a.pt.part_of_cb(cb);
  const int ltile0 = (cb - a.pt.cb_start[p]) * 16 + wave * 2;   // the wave's two 16-column tiles
  const int tile0 = a.pt.tile_start[p] + ltile0;
  const int nt = max(0, min(2, a.pt.tile_start[p + 1] - tile0));
  constexpr bool FROT = DIAG == 4;
  const unsigned short* xp = FROT ? a.xrot : a.xrot + (int64_t)p * a.rows * a.K;   // (fused rotation: x itself, shared by the partitions)

  const int n32 = lane & 31, kh = lane >> 5;
  const int jt = n32 >> 4, n = n32 & 15;
  // ragged partitions: a lane whose tile does not exist computes on a valid tile and is never stored
  const int my_tile = min(tile0 + (jt < nt ? jt : 0), a.pt.tiles - 1);
  const int my_ts = min(a.pt.szt_start[p] + ltile0 + (jt < nt ? jt : 0), a.pt.tsz - 1);
  const int64_t szrow = (int64_t)(a.pt.tsz >> 2) * 64;
  const unsigned* szp = sz_base + ((int64_t)(my_ts >> 2) * 16 + n) * 4 + (my_ts & 3);
  const u32x4* wq0 = wq_base + (int64_t)my_tile * a.tstride * 64 + (kh * 16 + n);   // lane' (kb = kh, n): k-steps 0, 2, 4, 6
  const int64_t wq_gstride = (int64_t)a.gstride * 64;

  // --- activation staging: wave w fills LDS rows 32 w .. 32 w + 31 (8 LDS-DMA pieces of 4 rows).  Lane l lands on
  // (row = 4 c + l / 16, physical slot l % 16) and therefore FETCHES logical slot phys ^ (row & 15).
  const int srow_in = lane >> 4, sphys = lane & 15;
  const unsigned short* asrc[RT];
#pragma unroll
  for (int c = 0; c < RT; ++c) {
    const int row = (wave * RT + c) * 4 + srow_in;
    const int grow = min(row0 + row, a.rows - 1);   // tail rows re-read the last valid row (never stored)
    asrc[c] = xp + (int64_t)grow * a.K + ((sphys ^ (row & 15)) << 3);
  }
  // Every piece is issued unconditionally (rows past the matrix re-read the last valid row), and through inline
  // asm: hipcc orders a builtin LDS-DMA against the ds_reads around it with `s_waitcnt lgkmcnt(0)` (it cannot see
  // that the DMA fills the OTHER buffer), which stalls every k-step on the fragment reads it has just issued.
  // Hidden from the compiler, the fragment waits stay counted (lgkmcnt(7): the oldest of eight reads in flight).
  // The price is doing the DMA's bookkeeping by hand: `s_waitcnt vmcnt(0)` in front of the barrier that
  // publishes a slab (cdna guide section 5.7 item 1; M0 is saved and restored inside the statement).
  const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) unsigned char*)lds);
  auto issue_a_piece = [&](int c, int g, int buf) {
    const unsigned short* src = asrc[c] + g * 128;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * (BMR * 256) + (wave * RT + c) * 1024));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
  };

  u32x4 qn0, qn1;
  unsigned szn[QS];
  auto load_b = [&](int g) {
    const u32x4* q = wq0 + (int64_t)g * wq_gstride;
    qn0 = q[0];
    qn1 = q[32];     // lane' (2 + kh, n): k-steps 1, 3, 5, 7
#pragma unroll
    for (int hq = 0; hq < QS; ++hq) szn[hq] = szp[(int64_t)(g * QS + hq) * szrow];
  };

  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

  // fragment read address: LDS row = 32 rt + (lane & 31), logical slot 2 s + kh -> physical (2 s) ^ kh ^ (row & 15);
  // rows are 256 bytes, so the k-step only XORs bits 5..7 of the byte address
  const unsigned a0 = (unsigned)((lane & 31) * 256 + (((kh ^ (lane & 15)) & 15) << 4));

  Dequant<AT> d;
  d.init();

  const int ks = blockIdx.z;
  const int g0 = ks * a.gps, g1 = min(a.G, g0 + a.gps);
#pragma unroll
  for (int c = 0; c < RT; ++c) issue_a_piece(c, g0, 0);
  load_b(g0);
  // one quantisation group; MORE = another group follows (its slab and weights are requested here).  The last
  // group is a second instantiation of the body, so that nothing in the loop is conditional.
  auto group = [&](int g, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const u32x4 qc0 = qn0, qc1 = qn1;
    unsigned szc[QS];
#pragma unroll
    for (int hq = 0; hq < QS; ++hq) szc[hq] = szn[hq];
    d.set_group(szc[0]);
    auto word = [&](int s) -> unsigned { return (s & 1) ? qc1[(s >> 1) & 3] : qc0[(s >> 1) & 3]; };
    // the first k-step's weights are dequantised before the barrier (they need nothing from LDS)
    if constexpr (DIAG == 1 || DIAG == 3) {
      d.out = (u32x4){qc0[0], qc0[1], qc1[0], qc1[1]};
    } else {
#pragma unroll
      for (int r = 0; r < 7; ++r) d.part(r, word(0));
    }
    u32x4 bcur = d.out;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab g have landed ...
    __syncthreads();                                   // ... and so have everyone else's; all have left the other buffer
    const int nbuf = (g + 1 - g0) & 1;
    if constexpr (MORE) load_b(g + 1);
    const unsigned char* abuf = lds + ((g - g0) & 1) * (BMR * 256);
    if constexpr (FROT) {
      static_assert(!FROT || RT == 8, "the fused rotation is built for 256-row blocks (wave w rotates row tile w)");
      // B operand: this wave's 32 rows of the un-rotated slab, all eight k-steps (the same fragment reads as the k-loop's)
      vec8 xb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xb[s] = *(const vec8*)(abuf + (a0 ^ (unsigned)(s << 5)) + wave * 8192);
      const unsigned short* rm = a.rmat + ((int64_t)(p * a.G + g) * 128) * 128;
      unsigned char* wbuf = lds + ((g - g0) & 1) * (BMR * 256) + wave * 8192 + n32 * 256;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {          // 32 rotated channels at a time: A operand = 32 rows of R' (n) x 16 k
        f32x16 r;
#pragma unroll
        for (int e = 0; e < 16; ++e) r[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const vec8 ra = *(const vec8*)(rm + (ct * 32 + n32) * 128 + 16 * s + 8 * kh);
          r = Mma32<AT>::run(ra, xb[s], r);
        }
        // D[i][j]: this lane holds row m = n32 (column j) and channels ct*32 + 8 q + 4 kh + (0..3): one 8-byte piece per q of the
        // 16-byte slot ct*4 + q, physical slot = logical ^ (row & 15)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 o;
          o[0] = (unsigned)A::from_f32(r[4 * q + 0]) | ((unsigned)A::from_f32(r[4 * q + 1]) << 16);
          o[1] = (unsigned)A::from_f32(r[4 * q + 2]) | ((unsigned)A::from_f32(r[4 * q + 3]) << 16);
          *(u32x2*)(wbuf + ((((ct * 4 + q) ^ (n32 & 15)) & 15) << 4) + 8 * kh) = o;
        }
      }
      __syncthreads();     // every wave's row tile is rotated before anyone multiplies by it
    }
    vec8 af[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) af[rt] = *(const vec8*)(abuf + a0 + rt * 8192);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const vec8 wf = __builtin_bit_cast(vec8, bcur);
      // eight slots per k-step: slot k carries the MFMA of row tile k (when there is one) and part k of the NEXT
      // k-step's dequant (always: the word has eight parts whatever the number of row tiles)
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        if constexpr (RT == 8) {
          acc[rt] = Mma32<AT>::run(wf, af[rt], acc[rt]);
        } else {
          if (rt < RT) acc[rt] = Mma32<AT>::run(wf, af[rt < RT ? rt : 0], acc[rt < RT ? rt : 0]);
        }
        if (s < 7) {
          if constexpr (DIAG == 1 || DIAG == 3) {
            if (rt == 0) d.out = (u32x4){word(s + 1), qc0[s & 3], qc1[s & 3], qc0[(s + 1) & 3]};
          } else {
            // group_size 64: k-step 4 opens the slab's second quantisation group (its weights are dequantised
            // during k-step 3, all eight parts of a word with the same scale / zero)
            if constexpr (QS == 2) {
              if (s == 3 && rt == 0) d.set_group(szc[1]);
            }
            d.part(rt, word(s + 1));
          }
          if constexpr (DIAG != 2 && DIAG != 3) {
            if (rt < RT) af[rt < RT ? rt : 0] = *(const vec8*)(abuf + (a0 ^ (unsigned)((s + 1) << 5)) + (rt < RT ? rt : 0) * 8192);
          }
        }
        // the next slab's DMA pieces go out early in the group (RT = 8: two per k-step over the first four k-steps;
        // fewer row tiles: one per k-step), so that the vmcnt drain in front of the next group's first weight use
        // finds them long landed
        if constexpr (MORE) {
          if constexpr (RT == 8) {
            if (s < 4 && rt == 1) issue_a_piece(2 * s, g + 1, nbuf);
            if (s < 4 && rt == 5) issue_a_piece(2 * s + 1, g + 1, nbuf);
          } else {
            if (s < RT && rt == 1) issue_a_piece(s < RT ? s : 0, g + 1, nbuf);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s < 7) bcur = d.out;
    }
  };
  for (int g = g0; g + 1 < g1; ++g) group(g, std::true_type{});
  group(g1 - 1, std::false_type{});

  // epilogue: register quad q of row tile rt = output columns 8 q + 4 kh .. + 3 of the wave's 32, row 32 rt + n32
  const int col_base = tile0 * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((q >> 1) >= nt) continue;
    const int col = col_base + 8 * q + 4 * kh;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = A::to_f32(a.bias[col + e]);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = row0 + rt * 32 + n32;
      if (row >= a.rows) continue;
      if (a.ksplit > 1) {   // K-split: the fp32 tile of this split, summed (+ bias) by gemm_reduce_kernel
        const f32x4 pv = {acc[rt][4 * q + 0], acc[rt][4 * q + 1], acc[rt][4 * q + 2], acc[rt][4 * q + 3]};
        *(f32x4*)(a.partial + ((int64_t)ks * a.rows + row) * a.N + col) = pv;
        continue;
      }
      u32x2 o;
      o[0] = (unsigned)A::from_f32(acc[rt][4 * q + 0] + bv[0]) | ((unsigned)A::from_f32(acc[rt][4 * q + 1] + bv[1]) << 16);
      o[1] = (unsigned)A::from_f32(acc[rt][4 * q + 2] + bv[2]) | ((unsigned)A::from_f32(acc[rt][4 * q + 3] + bv[3]) << 16);
      *(u32x2*)(a.y + (int64_t)row * a.N + col) = o;
    }
  }
}

template <typename AT, int QS>
static void launch_gemm3_rt(const GemmArgs& a, dim3 grid, hipStream_t st, int rt) {
  switch (rt) {
    case 2: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 2>), grid, dim3(512), 0, st, a); break;
    case 3: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 3>), grid, dim3(512), 0, st, a); break;
    case 4: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 4>), grid, dim3(512), 0, st, a); break;
    case 5: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 5>), grid, dim3(512), 0, st, a); break;
    case 6: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 6>), grid, dim3(512), 0, st, a); break;
    case 7: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 7>), grid, dim3(512), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm3_kernel<AT, 0, QS, 8>), grid, dim3(512), 0, st, a); break;
  }
}

// rt: 32-row tiles per row block (2..8 -- grid.y counts blocks of 32 rt rows)
int launch_gemm3(const GemmArgs& a, int act_dtype, dim3 grid, hipStream_t st, int diag, int qs, int rt) {
  if (rt < 2 || rt > 8) return fail(PARO_ERR_INVALID, "GEMM variant 4 is built for 2..8 row tiles (got %d)", rt);
  if (diag != 0) {
    if (qs == 2 || rt != 8 || act_dtype != PARO_DTYPE_F16 || a.ksplit > 1)
      return fail(PARO_ERR_UNSUPPORTED, "the ablation builds of GEMM variant 4 exist for fp16, group_size 128, 256-row blocks, no K-split only");
    if (diag == 4) {
      if (!a.rmat) return fail(PARO_ERR_INVALID, "GEMM variant 44 (fused rotation) needs paro_linear_t.rmat");
      hipLaunchKernelGGL((gemm3_kernel<f16, 4>), grid, dim3(512), 0, st, a);
    } else if (diag == 1)
      hipLaunchKernelGGL((gemm3_kernel<f16, 1>), grid, dim3(512), 0, st, a);
    else if (diag == 2)
      hipLaunchKernelGGL((gemm3_kernel<f16, 2>), grid, dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL((gemm3_kernel<f16, 3>), grid, dim3(512), 0, st, a);
    return PARO_OK;
  }
  if (act_dtype == PARO_DTYPE_F16) {
    if (qs == 2) launch_gemm3_rt<f16, 2>(a, grid, st, rt); else launch_gemm3_rt<f16, 1>(a, grid, st, rt);
  } else {
    if (qs == 2) launch_gemm3_rt<bf16, 2>(a, grid, st, rt); else launch_gemm3_rt<bf16, 1>(a, grid, st, rt);
  }
  return PARO_OK;
}

}  // namespace paro

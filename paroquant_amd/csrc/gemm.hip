// W4A16 MFMA GEMM for gfx950 (prefill / large-batch path):  y = rotate(x) @ dequant(W) (+ bias).
//
// Two steps behind one ABI call (paro_w4a16_gemm):
//   1. rotate pre-pass (rotate.hip) writes the rotated activations once per merged partition into the
//      caller's workspace  xrot[p][rows][K]  (the reference re-launches `rotate` per partition too --
//      vllm/plugin.py:288-290 -- and then hands fp16 x to a separate Marlin GEMM);
//   2. one of three GEMM kernels (the `variant` knob of the ABI; 0 = auto):
//        1  gemm_kernel<T>          128 x 128 tile, 4 waves, register-staged A, per-group scale epilogue
//                                   (f16 / bf16; what bf16 runs below 256 rows)
//        2  gemm2_f16_kernel<2>     256 x 128 tile, 4 waves (2 x 2), A by LDS-DMA, exact fp16 weights in
//                                   registers, optional K-split over grid.z for small M
//        (3, the round-1 256 x 256 kernel with 2 x 4 waves, was kept for A/B in round 2 and removed in ABI v11)
//        4  gemm3_kernel<T>         256 x 256 tile, 8 waves (1 x 8), 32x32x16 MFMA (gemm3.hip), f16 / bf16
//      In all of them the INT4 B tiles are NOT staged through LDS: a lane's 16-byte load of the packed
//      tile (paro_repack_awq) is its MFMA operand after the in-register dequant.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_args.hpp"

namespace paro {

int launch_rotate(const void* x, void* out, const int16_t* idx, const void* theta, const void* scales,
                  int64_t rows, int64_t hidden, int krot, int gs, int x_dt, int p_dt, hipStream_t st, int nparts);
int validate_linear(const paro_linear_t* L);
int launch_prerot_sched(const void* x, void* out, const void* rot, const void* cs, int64_t rows, int64_t K, int krot, int nparts,
                        int dt, int frag_row_tiles, hipStream_t st);   // rotate.hip
int launch_rotate_mfma(const void* x, void* out, const void* rmat, int64_t rows, int64_t K, int nparts, int dt,
                       hipStream_t st);

constexpr int BM = 128;
constexpr int BN_TILES = 8;  // 128 columns per workgroup

template <typename AT, int QS = 1>   // QS: quantisation groups per 128-channel span (2 = group_size 64)
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  __shared__ __attribute__((aligned(16))) unsigned char lds[BM * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int cb = blockIdx.x;
  const int row0 = blockIdx.y * BM;

  const int p = a.pt.part_of_cb(cb);
  const int ltile0 = (cb - a.pt.cb_start[p]) * BN_TILES + wc * 4;
  const int tile0 = a.pt.tile_start[p] + ltile0;
  const int nt = max(0, min(4, a.pt.tile_start[p + 1] - tile0));
  const int ts0 = a.pt.szt_start[p] + ltile0;  // multiple of 4
  const unsigned short* xp = a.xrot + (int64_t)p * a.rows * a.K;

  const int n = lane & 15, mq = lane >> 4;
  const int64_t szrow = (int64_t)(a.pt.tsz >> 2) * 64;
  const unsigned* szp = a.sz + ((int64_t)(ts0 >> 2) * 16 + n) * 4;

  // staging map: chunk c -> row (tid >> 4) + 16 c, 16-byte slot tid & 15
  const int srow = tid >> 4, sslot = tid & 15;

  f32x4 acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[rt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 stg[8];
  u32x4 qv[4];
  u32x4 szv[QS];
  constexpr int SPH = 4 / QS;   // MFMA k-steps per quantisation group

  auto issue_loads = [&](int g) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int row = row0 + srow + 16 * c;
      stg[c] = (u32x4){0u, 0u, 0u, 0u};
      if (row < a.rows) stg[c] = *(const u32x4*)(xp + (int64_t)row * a.K + g * 128 + sslot * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nt) qv[j] = *(a.wq + ((int64_t)(tile0 + j) * a.tstride + (int64_t)g * a.gstride) * 64 + lane);
    if (nt > 0) {
#pragma unroll
      for (int hq = 0; hq < QS; ++hq) szv[hq] = *(const u32x4*)(szp + (int64_t)(g * QS + hq) * szrow);
    }
  };

  issue_loads(0);
  for (int g = 0; g < a.G; ++g) {
    __syncthreads();  // every wave is done reading the previous A tile
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int row = srow + 16 * c;
      *(u32x4*)(lds + row * 256 + ((sslot ^ (row & 15)) << 4)) = stg[c];
    }
    u32x4 qc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qc[j] = qv[j];
    u32x4 szc[QS];
#pragma unroll
    for (int hq = 0; hq < QS; ++hq) szc[hq] = szv[hq];
    __syncthreads();
    if (g + 1 < a.G) issue_loads(g + 1);  // next group's A rows + INT4 tiles fly under this group's MFMAs

    vec8 af[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = wr * 64 + rt * 16 + n;  // MFMA row m' = lane & 15
#pragma unroll
      for (int i = 0; i < 4; ++i) af[rt][i] = *(const vec8*)(lds + row * 256 + (((4 * i + mq) ^ (row & 15)) << 4));
    }
    f32x4 sx[4][QS];
    {
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const vec8 ob = __builtin_bit_cast(vec8, ones);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
        for (int hq = 0; hq < QS; ++hq) sx[rt][hq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) sx[rt][i / SPH] = A::mfma(af[rt][i], ob, sx[rt][i / SPH]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < nt) {
        f32x4 d[4][QS];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int hq = 0; hq < QS; ++hq) d[rt][hq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned w4[4];
          A::unpack(qc[j][i], w4);
          const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
          const vec8 bf = __builtin_bit_cast(vec8, wv);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) d[rt][i / SPH] = A::mfma(af[rt][i], bf, d[rt][i / SPH]);
        }
#pragma unroll
        for (int hq = 0; hq < QS; ++hq) {
          const float s = f16_bits_to_f32(szc[hq][j] & 0xffffu);
          const float zf = f16_bits_to_f32(szc[hq][j] >> 16) + 16.f;  // unpack() yields 16 + q
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[rt][j][r] = __builtin_fmaf(s, __builtin_fmaf(-zf, sx[rt][hq][r], d[rt][hq][r]), acc[rt][j][r]);
        }
      }
    }
  }

  // epilogue: D layout row = 4 * (lane >> 4) + r, col = lane & 15
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < nt) {
      const int col = (tile0 + j) * 16 + n;
      const float bv = a.bias ? A::to_f32(a.bias[col]) : 0.f;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + wr * 64 + rt * 16 + 4 * mq + r;
          if (row < a.rows) a.y[(int64_t)row * a.N + col] = A::from_f32(acc[rt][j][r] + bv);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp16 main kernel (v2): 256 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each
// 128 rows x 64 columns = 8 x 4 MFMA tiles, 128 accumulator VGPRs).
//   * A: the 256-row x 128-k tile of the current group goes global -> LDS directly
//     (global_load_lds_dwordx4, no staging VGPRs), double buffered (2 x 64 KiB), one barrier per
//     group.  The LDS image is lane-linear per wave-load, so the 16-slot XOR swizzle that makes the
//     ds_read_b128 fragments conflict-free is applied to the SOURCE address (cdna guide rule 21).
//   * B: four 1-KiB INT4 tiles per wave per group straight to VGPRs (next group prefetched), turned
//     into exact fp16 weights in registers: (off + q) - (off + z) is exact, * s rounds once -- the
//     reference's fp16 (q - z) * s dequant, bit for bit -- 13 packed-VALU ops per 8 weights, amortised
//     over 8 row tiles (32 MFMA per 52 VALU).
//   * accumulation runs through ALL groups in one MFMA chain per accumulator; no per-group epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int BM2 = 256;

template <int WCOLS>  // column waves: 2 -> 256 x 128 tile, 4 waves; 4 -> 256 x 256 tile, 8 waves (2 per SIMD)
__global__ __launch_bounds__(WCOLS * 128) void gemm2_f16_kernel(const GemmArgs a) {
  typedef Act<f16> A;
  typedef A::vec8 vec8;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM2 * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = WCOLS * 2;            // waves per workgroup
  constexpr int CBT = WCOLS * 4;           // column tiles per workgroup
  constexpr int CPW = 64 / NW;             // A wave-loads (4 rows each) per wave per group
  const int wr = wave / WCOLS, wc = wave % WCOLS;
  const int cb = blockIdx.x;
  const int row0 = blockIdx.y * BM2;

  const int p = a.pt.part_of_cb(cb);
  const int ltile0 = (cb - a.pt.cb_start[p]) * CBT + wc * 4;
  const int tile0 = a.pt.tile_start[p] + ltile0;
  const int nt = max(0, min(4, a.pt.tile_start[p + 1] - tile0));
  const int ts0 = a.pt.szt_start[p] + ltile0;  // multiple of 4
  const unsigned short* xp = a.xrot + (int64_t)p * a.rows * a.K;

  const int n = lane & 15, mq = lane >> 4;
  const int64_t szrow = (int64_t)(a.pt.tsz >> 2) * 64;
  // ragged partitions can push a wave's tiles past the padded scale/zero area: clamp (never stored)
  const unsigned* szp = a.sz + ((int64_t)(min(ts0, a.pt.tsz - 4) >> 2) * 16 + n) * 4;

  // --- A staging: wave w fills rows 64w .. 64w+63 (16 wave-loads of 4 rows).  Lane l lands on LDS slot
  // (row = 4c + l/16, phys = l%16) and must therefore FETCH logical slot phys ^ (row & 15).
  const int srow_in = lane >> 4, sphys = lane & 15;
  const unsigned short* asrc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int row = (wave * CPW + c) * 4 + srow_in;
    const int grow = min(row0 + row, a.rows - 1);  // tail rows re-read the last valid row (never stored)
    asrc[c] = xp + (int64_t)grow * a.K + ((sphys ^ (row & 15)) << 3);
  }
  // small M: rows beyond a.rows are not staged (a 256-row tile at M = 32 would spend 7/8 of its LDS-DMA
  // traffic -- x_rot re-read by every column block -- on copies of the last row); the MFMAs still run on
  // whatever those LDS rows hold, their results are never stored
  const int rows_here = min(BM2, a.rows - row0);
  auto issue_a = [&](int g, int buf) {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if ((wave * CPW + c) * 4 >= rows_here) continue;   // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[c] + g * 128),
                                       (__attribute__((address_space(3))) void*)(lds + buf * (BM2 * 256) + (wave * CPW + c) * 1024),
                                       16, 0, 0);
    }
  };

  u32x4 qn[4];
  u32x4 szn;
  auto load_b = [&](int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // every wave stages A and joins every barrier, so out-of-range tiles (ragged partitions) are
      // clamped to a valid tile and simply never stored
      const int tt = min(tile0 + (j < nt ? j : 0), a.pt.tiles - 1);
      qn[j] = *(a.wq + ((int64_t)tt * a.tstride + (int64_t)g * a.gstride) * 64 + lane);
    }
    szn = *(const u32x4*)(szp + (int64_t)g * szrow);
  };

  const A::Unpack upk = A::unpack_consts();
  f32x4 acc[8][4];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[rt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = wr*128 + rt*16 + m', logical slot 4i + kb -> phys = (4i + kb) ^ m'
  const int mrow = lane & 15;
  const int abase = (wr * 128 + mrow) * 256;
  int aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = abase + (((4 * i + mq) ^ mrow) << 4);

  // K-split (small M: too few output tiles to fill the chip): this workgroup covers groups [g0, g1)
  const int ks = blockIdx.z;
  const int g0 = ks * a.gps, g1 = min(a.G, g0 + a.gps);
  {
    issue_a(g0, 0);
    load_b(g0);
    for (int g = g0; g < g1; ++g) {
      u32x4 qc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) qc[j] = qn[j];
      const u32x4 szc = szn;
      __syncthreads();  // A(g) has landed (the barrier drains the LDS-DMA), everyone left the other buffer
      if (g + 1 < g1) {
        issue_a(g + 1, (g + 1 - g0) & 1);
        load_b(g + 1);
      }
      const unsigned char* abuf = lds + ((g - g0) & 1) * (BM2 * 256);
      // per-(tile, group) dequant constants as packed halves
      f16x2 s2[4], c_hi[4], c_lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16 sh = __builtin_bit_cast(f16, (unsigned short)(szc[j] & 0xffffu));
        const f16 zh = __builtin_bit_cast(f16, (unsigned short)(szc[j] >> 16));
        s2[j] = (f16x2){sh, sh};
        const f16 ch = (f16)(-1024.f) - zh, cl = (f16)(-64.f) - zh;  // exact: |.| <= 1039 integers
        c_hi[j] = (f16x2){ch, ch};
        c_lo[j] = (f16x2){cl, cl};
      }
      // The group's body in explicit issue order.  Left to the compiler it comes out as [13 dequant VALU]
      // [8 MFMA] clusters, and the PMC profile of that version showed VALU-active (37 %) and MFMA-busy
      // (40 %) time simply adding up: an in-order wave only overlaps the two pipes if independent VALU sit
      // BETWEEN its MFMAs.  So each of the 8 MFMAs of tile (i, j) is followed by a slice of the dequant of
      // the NEXT tile's word (13 VALU over 7 slots) and, in the last tile of a k-step, by the ds_read of
      // the fragment that MFMA just used, for the next k-step; sched_barrier(0) pins every slot.
      struct DQ {
        unsigned t, o0, o1, o2, o3;
        f16x2 a0, a1, a2, a3;
        u32x4 out;
      };
      auto dq_part = [&](int r, DQ& d, unsigned w, int j) {
        if (r == 0) {
          d.t = w >> 8;
          d.o0 = (w & upk.m0) | upk.k0;
        } else if (r == 1) {
          d.o1 = (w & upk.m1) | upk.k1;
          d.o2 = (d.t & upk.m0) | upk.k0;
        } else if (r == 2) {
          d.o3 = (d.t & upk.m1) | upk.k1;
          d.a0 = __builtin_bit_cast(f16x2, d.o0) + c_hi[j];
        } else if (r == 3) {
          d.out[0] = __builtin_bit_cast(unsigned, d.a0 * s2[j]);
          d.a1 = __builtin_bit_cast(f16x2, d.o1) + c_lo[j];
        } else if (r == 4) {
          d.out[1] = __builtin_bit_cast(unsigned, d.a1 * s2[j]);
          d.a2 = __builtin_bit_cast(f16x2, d.o2) + c_hi[j];
        } else if (r == 5) {
          d.out[2] = __builtin_bit_cast(unsigned, d.a2 * s2[j]);
          d.a3 = __builtin_bit_cast(f16x2, d.o3) + c_lo[j];
        } else if (r == 6) {
          d.out[3] = __builtin_bit_cast(unsigned, d.a3 * s2[j]);
        }
      };
      vec8 af[8];
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) af[rt] = *(const vec8*)(abuf + aoff[0] + rt * 4096);
      DQ d;
#pragma unroll
      for (int r = 0; r < 7; ++r) dq_part(r, d, qc[0][0], 0);
      u32x4 bcur = d.out;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool has_next = !(i == 3 && j == 3);
          const int jn = (j + 1) & 3, in = j == 3 ? i + 1 : i;
          const vec8 bf = __builtin_bit_cast(vec8, bcur);
#pragma unroll
          for (int rt = 0; rt < 8; ++rt) {
            acc[rt][j] = A::mfma(af[rt], bf, acc[rt][j]);
            if (has_next) dq_part(rt, d, qc[jn][in & 3], jn);
            if (j == 3 && i < 3) af[rt] = *(const vec8*)(abuf + aoff[i + 1] + rt * 4096);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (has_next) bcur = d.out;
        }
      }
    }
  }

  // epilogue: D layout row = 4 * (lane >> 4) + r, col = lane & 15
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < nt) {
      const int col = (tile0 + j) * 16 + n;
      const float bv = (a.bias && a.ksplit == 1) ? A::to_f32(a.bias[col]) : 0.f;
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + wr * 128 + rt * 16 + 4 * mq + r;
          if (row < a.rows) {
            if (a.ksplit == 1)
              a.y[(int64_t)row * a.N + col] = A::from_f32(acc[rt][j][r] + bv);
            else
              a.partial[((int64_t)ks * a.rows + row) * a.N + col] = acc[rt][j][r];   // summed by gemm_reduce_kernel
          }
        }
    }
  }
}

// y = fp16(sum over K-splits of the fp32 partial tiles + bias); 4 columns per thread
template <typename AT>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ partial,
                                                         const unsigned short* __restrict__ bias,
                                                         unsigned short* __restrict__ y, int64_t rows, int N, int ksplit) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index of a group of 4 columns
  const int64_t total4 = rows * N / 4;
  if (i4 >= total4) return;
  f32x4 v = *(const f32x4*)(partial + i4 * 4);
  for (int s = 1; s < ksplit; ++s) v += *(const f32x4*)(partial + (int64_t)s * rows * N + i4 * 4);
  const int col = (int)((i4 * 4) % N);
  u32x2 o;
  unsigned short h[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = Act<AT>::from_f32(v[e] + (bias ? Act<AT>::to_f32(bias[col + e]) : 0.f));
  o[0] = (unsigned)h[0] | ((unsigned)h[1] << 16);
  o[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
  *(u32x2*)(y + i4 * 4) = o;
}

}  // namespace paro

namespace paro {
// K-split of the v2 GEMM for small M: with one 256-row block the grid is only N / 128 workgroups (32 for
// N = 4096), each looping over all of K (o_proj, M = 32..128: 70 us); splitting K over grid.z and summing
// fp32 partial tiles in a second small kernel fills the chip.  Used for 17 <= rows < 4096 when the grid is at most
// half of the CUs (narrow outputs at M = 512 / 1024 ran 64 / 128 workgroups over all of K: down_proj 230 us); at most
// 8 splits of at least 2 groups.
int gemm_ksplit(const paro_linear_t* L, int64_t rows) {
  if (L->act_dtype != PARO_DTYPE_F16 || rows <= 16 || rows >= 4096) return 1;
  const int64_t wgs = ((L->N + 127) / 128) * ((rows + 255) / 256);
  if (wgs > 128) return 1;
  const int G = (int)(L->K / 128);
  int ks = (int)(256 / wgs);
  if (ks > 8) ks = 8;
  if (ks > G / 2) ks = G / 2;
  return ks < 1 ? 1 : ks;
}
// Variant 4's block shape between 33 and 4095 rows (batched decode, short and medium prefill): row tiles per block and K-split.
// Round 6 (tools/sweep_gemm4.py, profiles/r06_sweep_gemm4_*.jsonl -- 18 shapes per linear and row count): what matters is ONE round of
// ~192..256 workgroups over the 256 CUs; small row blocks reach it with few K-splits (the fp32 partial tiles of a split go through memory
// twice: Llama-3-8B qkv at 256 rows on one 256-row block x 8 splits wrote and re-read 50 MB beside 12.6 MB of weights), the weights a
// second row block re-reads come from L2.  So: of 2 / 4 / 8 row tiles, those whose grid of (row blocks x 256-column blocks)
// fit one round, whichever fills it best with its K-split (g4_shape below).  us rule before /
// after, Llama-3-8B: qkv 256 rows 51 -> 42, 512 rows 71 -> 55, 1024 rows 100 -> 77; o 1024 rows 68 -> 52; down 1024 rows 165 -> 145.
// (PARO_GEMM4_TUNE set: PARO_GEMM4_RT / PARO_GEMM4_KS are read at EVERY call and override both -- tools/sweep_gemm4.py, one process)
static const bool g4_tune = getenv("PARO_GEMM4_TUNE") != nullptr;
static int g4_env(const char* n) { const char* v = getenv(n); return v ? atoi(v) : 0; }
// K-splits of a grid of `wgs` blocks: fill ONE round, at most clamp(groups / 8, 4, 8) splits of at least two groups (deep K takes more:
// Qwen3-4B down 128 rows x 8 = 25 us against x 4 = 30), and no more than K / (4 rows) -- a split's fp32 partial tile is written and read
// back, 8 bytes per element and split against the weights' K / 2 per column: beyond ~4x the weight bytes the splits cost more than they
// fill (Qwen3-4B down 384 rows: x 8 = 49 us, x 6 = 42)
static int g4_splits(const paro_linear_t* L, int64_t rows, int64_t wgs) {
  if (rows <= 16 || rows >= 4096 || wgs > 128) return 1;
  const int G = (int)(L->K / 128);
  int ks = (int)(256 / wgs);
  const int cap = G / 8 > 4 ? (G / 8 > 8 ? 8 : G / 8) : 4;
  if (ks > cap) ks = cap;
  if (ks > G / 2) ks = G / 2;
  if (rows > 64 && ks > L->K / (4 * rows)) ks = (int)(L->K / (4 * rows));
  return ks < 1 ? 1 : ks;
}
static void g4_shape(const paro_linear_t* L, int64_t rows, int& rt_out, int& ks_out) {
  const int64_t cbs = (L->N + 255) / 256;
  const int G = (int)(L->K / 128);
  auto wgs = [&](int rt) { return ((rows + 32 * rt - 1) / (32 * rt)) * cbs; };
  if (rows <= 64 || rows >= 4096) {
    rt_out = rows <= 64 ? 2 : 8;
    ks_out = g4_splits(L, rows, wgs(rt_out));
    return;
  }
  // deep K above 256 rows: 128-row blocks with more K-splits beat 64-row blocks with fewer (Llama-3-8B down 384 rows 4 x 4 = 67 us
  // against 2 x 2 = 76; Qwen3-4B down 1024 rows 4 x 3 = 80 against 2 x 1 = 92)
  const int rt_min = (G >= 64 && rows > 256) ? 4 : 2;
  // one round: the smallest block whose grid covers more than half of the CUs UNSPLIT (a K-split costs its partial tiles and the reduce
  // launch: Qwen3-4B gate_up 128 rows 152 blocks unsplit 29 us, 76 x 3 splits 35), else the smallest whose blocks x splits fill three
  // quarters of the round, else the fullest
  int best = 0, best_ks = 1;
  int64_t best_fill = 0;
  for (int rt = rt_min; rt <= 8; rt *= 2) {
    if (wgs(rt) > 256) continue;
    const int ks = g4_splits(L, rows, wgs(rt));
    const int64_t fill = wgs(rt) * ks;
    if (wgs(rt) > 128 || fill >= 192) { best = rt; best_ks = ks; break; }
    if (fill > best_fill) { best_fill = fill; best = rt; best_ks = ks; }
  }
  if (best) { rt_out = best; ks_out = best_ks; return; }
  // more than one round even on 256-row blocks: the shape that wastes least of its last round, small blocks discounted for the
  // dequantisation they repeat (Qwen3-4B gate_up 1024 rows: 304 blocks of 256 rows = 1.19 rounds 147 us, 1216 of 64 rows 129 us)
  best = 8;
  double best_eff = 0.0;
  for (int rt = 8; rt >= rt_min; rt /= 2) {
    const double w = (double)wgs(rt), eff = w / (double)(((int64_t)(w + 255) / 256) * 256) * (rt == 8 ? 1.0 : (rt == 4 ? 0.95 : 0.85));
    if (eff > best_eff + 1e-9) { best_eff = eff; best = rt; }
  }
  rt_out = best;
  ks_out = 1;
}
int gemm4_row_tiles(const paro_linear_t* L, int64_t rows) {
  if (g4_tune) {
    const int rt = g4_env("PARO_GEMM4_RT");
    if (rt >= 2 && rt <= 8) return rt;
  }
  int rt, ks;
  g4_shape(L, rows, rt, ks);
  return rt;
}
int gemm4_ksplit(const paro_linear_t* L, int64_t rows) {
  if (g4_tune) {
    const int ks = g4_env("PARO_GEMM4_KS"), G = (int)(L->K / 128);
    if (ks >= 1) return ks > G ? G : ks;
  }
  int rt, ks;
  g4_shape(L, rows, rt, ks);
  return ks;
}
}  // namespace paro

namespace paro {
int launch_gemm3(const GemmArgs& a, int act_dtype, dim3 grid, hipStream_t st, int diag, int qs, int rt);   // gemm3.hip

}

// host-only: the block shape variant 4 runs `rows` rows of this layer with (tooling, logs, tests of the rule)
extern "C" int paro_gemm_launch_shape(const paro_linear_t* L, int64_t rows, int* block_rows, int* ksplit) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows < 1 || !block_rows || !ksplit) return fail(PARO_ERR_INVALID, "paro_gemm_launch_shape: bad arguments");
  *block_rows = 32 * gemm4_row_tiles(L, rows);
  *ksplit = gemm4_ksplit(L, rows);
  return PARO_OK;
}

extern "C" int paro_w4a16_gemm(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                               int64_t workspace_bytes, int variant, void* stream) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows == 0) return PARO_OK;
  if (rows < 0 || rows > 0x7fffffff) return fail(PARO_ERR_INVALID, "rows out of range");
  if (!x || !y) return fail(PARO_ERR_INVALID, "null pointer");
  int diag = 0;
  if (variant >= 41 && variant <= 44) {   // ablation builds of variant 4 (41..43: wrong results; 44: the fused-rotation experiment, correct; tools/bench_gemm.py)
    diag = variant - 40;
    variant = 4;
  }
  if (variant < 0 || variant > 4 || variant == 3)
    return fail(PARO_ERR_INVALID, "variant must be 0 (auto), 1, 2 or 4 (got %d; 3 -- the round-1 256 x 256 kernel -- was removed in ABI v11)", variant);
  if (diag == 4 && (!L->rmat || L->krot > 16))
    return fail(PARO_ERR_INVALID, "GEMM variant 44 (rotation fused into the GEMM, experiment) needs the dense rotation matrices (paro_linear_t.rmat)");
  const bool f16in = L->act_dtype == PARO_DTYPE_F16;
  if (variant == 2 && !f16in) return fail(PARO_ERR_UNSUPPORTED, "GEMM variant 2 is fp16-only; bf16 runs variant 1 or 4");
  // ---- kernel choice.  Auto: variant 4 (256-column blocks, 1 x 8 waves, 32x32x16 MFMA) from 33 rows on (64- / 128- / 256-row blocks by
  // gemm4_row_tiles); below that (the GEMV's range: only explicit calls get here) fp16 -> 256 x 128 tile with a K-split, bf16 -> the 128 x 128 kernel.
  const int qs = 128 / quant_group(L->group_size);
  if (qs == 2 && variant == 2)
    return fail(PARO_ERR_UNSUPPORTED, "group_size 64 runs GEMM variant 1 or 4 (variant 2 is built for group_size 128)");
  int v = variant;
  if (v == PARO_GEMM_AUTO) {
    // 33..192 rows (batched decode, short prefill): variant 4 with 64- .. 192-row blocks (32-row steps) and a K-split -- 256-column
    // blocks halve the re-reads of the activation tile, no padding rows are staged or multiplied (Llama-3-8B gate_up
    // 64 rows: 67 -> 34 us, down 42 -> 30, qkv 128 rows: 42 -> 31)
    if (rows >= 33) v = 4;   // (round 6: with 64- / 128-row blocks variant 4 is ahead of the 256 x 128 kernel on narrow grids above 192 rows too -- gemm4_row_tiles)
    else if (f16in && rows > 16 && qs == 1) v = 2;
    else v = 1;
  }
  const int rt4 = diag ? 8 : gemm4_row_tiles(L, rows);   // (the ablation / experiment builds exist for 256-row blocks)
  const int ksplit_req = v == 2 ? gemm_ksplit(L, rows) : (v == 4 && diag == 0 ? gemm4_ksplit(L, rows) : 1);
  const int64_t xrot_bytes = (int64_t)L->n_parts * rows * L->K * 2;
  const int64_t need = PARO_WS_COUNTER_BYTES + xrot_bytes + (ksplit_req > 1 ? 256 + (int64_t)ksplit_req * rows * L->N * 4 : 0);
  if (!workspace || workspace_bytes < need)
    return fail(PARO_ERR_INVALID, "workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  unsigned short* xrot = (unsigned short*)((char*)workspace + PARO_WS_COUNTER_BYTES);
  // (Merged projections: rotating partition p + 1 on a side stream while the GEMM of partition p runs was built in round 5 and does not
  // overlap -- a GEMM workgroup holds 128 KB of LDS, a pre-pass workgroup 68 KB, a CU has 160 KB: profiles/r05_prefill_overlap.txt, NOTES 5.4;
  // the host branch was removed in round 6.)
  static const int env_sched = getenv("PARO_PREROT_SCHED") ? atoi(getenv("PARO_PREROT_SCHED")) : 1;   // 0: the stage kernel, as up to round 5 (A/B)
  if (diag == 4)                // variant 44: the rotation runs inside the GEMM on the un-rotated x -- no pre-pass, no rotated copy
    rc = PARO_OK;
  else if (L->rmat && rows >= 256)   // many rows: one dense 128x128 product per group on the matrix cores
    rc = launch_rotate_mfma(x, xrot, L->rmat, rows, L->K, L->n_parts, L->act_dtype, st);
  else if (L->rot && L->krot <= 8 && env_sched != 0)   // short prefill / batched decode: the schedule pre-pass (rotate.hip), plain rows
    rc = launch_prerot_sched(x, xrot, L->rot, L->channel_scales, rows, L->K, L->krot, L->n_parts, L->act_dtype, 0, st);
  else
    rc = launch_rotate(x, xrot, L->pairs, L->theta, L->channel_scales, rows, L->K, L->krot, 128, L->act_dtype,
                       PARO_DTYPE_F16, st, L->n_parts);  // one launch, blockIdx.z = merged partition
  if (rc != PARO_OK) return rc;
  GemmArgs a;
  a.wq = (const u32x4*)L->wq;
  a.sz = (const unsigned*)L->sz;
  a.bias = (const unsigned short*)L->bias;
  a.xrot = diag == 4 ? (const unsigned short*)x : xrot;
  a.rmat = (const unsigned short*)L->rmat;
  a.y = (unsigned short*)y;
  a.K = (int)L->K;
  a.N = (int)L->N;
  a.G = (int)(L->K / 128);
  a.rows = (int)rows;
  a.tstride = L->wq_order ? 1 : a.G;
  a.gstride = L->wq_order ? (int)(L->N / 16) : 1;
  const bool wide = v == 4;
  if (!fill_part_table(a.pt, L->n_parts, L->part_cols, wide ? 16 : BN_TILES)) return fail(PARO_ERR_INVALID, "bad partition table");
  const int bm = v == 1 ? BM : (v == 4 ? 32 * rt4 : BM2);
  const int64_t rb = (rows + bm - 1) / bm;
  if (rb > 65535) return fail(PARO_ERR_INVALID, "rows too large for one launch (max %d)", 65535 * bm);
  a.ksplit = 1;
  a.gps = a.G;
  a.partial = nullptr;
  if ((v == 2 || v == 4) && ksplit_req > 1) {
    a.gps = (a.G + ksplit_req - 1) / ksplit_req;
    a.ksplit = (a.G + a.gps - 1) / a.gps;   // drop empty splits
    a.partial = (float*)((char*)workspace + PARO_WS_COUNTER_BYTES + ((xrot_bytes + 255) / 256) * 256);
  }
  dim3 grid((unsigned)a.pt.cbs, (unsigned)rb, (unsigned)a.ksplit);
  if (v == 4) {
    rc = launch_gemm3(a, L->act_dtype, grid, st, diag, qs, rt4);
    if (rc != PARO_OK) return rc;
    if (a.ksplit > 1) {
      const int64_t total4 = rows * L->N / 4;
      const dim3 rg((unsigned)((total4 + 255) / 256));
      if (f16in)
        hipLaunchKernelGGL(gemm_reduce_kernel<f16>, rg, dim3(256), 0, st, a.partial, (const unsigned short*)L->bias, (unsigned short*)y, rows, (int)L->N, a.ksplit);
      else
        hipLaunchKernelGGL(gemm_reduce_kernel<bf16>, rg, dim3(256), 0, st, a.partial, (const unsigned short*)L->bias, (unsigned short*)y, rows, (int)L->N, a.ksplit);
    }
  } else if (v == 2) {
    hipLaunchKernelGGL(gemm2_f16_kernel<2>, grid, dim3(256), 0, st, a);
    if (a.ksplit > 1) {
      const int64_t total4 = rows * L->N / 4;
      hipLaunchKernelGGL(gemm_reduce_kernel<f16>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, a.partial,
                         (const unsigned short*)L->bias, (unsigned short*)y, rows, (int)L->N, a.ksplit);
    }
  } else if (qs == 2) {
    if (f16in)
      hipLaunchKernelGGL((gemm_kernel<f16, 2>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((gemm_kernel<bf16, 2>), grid, dim3(256), 0, st, a);
  } else if (f16in) {
    hipLaunchKernelGGL(gemm_kernel<f16>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(gemm_kernel<bf16>, grid, dim3(256), 0, st, a);
  }
  return check_launch("paro_w4a16_gemm");
}

// Grouped W4A16 GEMM over expert segments (the reference exports ONE rotation per MoE projection shared by
// all experts, cli/convert.py:280-379, and rotates the tokens once before the routed experts, mlx/modules.py:159-212):
// x_rot [padded_rows][K] holds the ROTATED rows of the (token, expert) pairs sorted by expert, every expert's segment
// padded up to a multiple of block_rows; row block b multiplies by expert block_expert[b]'s packed weights (device
// memory, -1 = unused block): one launch for all experts, nothing on the host, HIP-graph capturable.
extern "C" int paro_w4a16_gemm_grouped(const paro_linear_t* L, const void* x_rot, void* y, int64_t padded_rows, int block_rows,
                                       const int32_t* block_expert, int64_t wq_stride_bytes, int64_t sz_stride_bytes, int32_t n_experts,
                                       void* stream) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (!x_rot || !y || !block_expert) return fail(PARO_ERR_INVALID, "null pointer");
  if (block_rows != 64 && block_rows != 128 && block_rows != 256) return fail(PARO_ERR_INVALID, "block_rows must be 64, 128 or 256 (got %d)", block_rows);
  if (padded_rows <= 0 || padded_rows % block_rows) return fail(PARO_ERR_INVALID, "padded_rows must be a positive multiple of block_rows");
  if (L->n_parts != 1) return fail(PARO_ERR_UNSUPPORTED, "grouped GEMM: one rotation partition per projection (the experts share it)");
  if (wq_stride_bytes % 16 || sz_stride_bytes % 4) return fail(PARO_ERR_INVALID, "expert strides must keep the packed buffers aligned");
  if (padded_rows / block_rows > 65535) return fail(PARO_ERR_INVALID, "too many row blocks");
  if (n_experts < 1) return fail(PARO_ERR_INVALID, "n_experts must be >= 1 (the device checks every block's expert id against it)");
  GemmArgs a;
  a.wq = (const u32x4*)L->wq;
  a.sz = (const unsigned*)L->sz;
  a.bias = nullptr;
  a.xrot = (const unsigned short*)x_rot;
  a.y = (unsigned short*)y;
  a.K = (int)L->K;
  a.N = (int)L->N;
  a.G = (int)(L->K / 128);
  a.rows = (int)padded_rows;
  a.tstride = L->wq_order ? 1 : a.G;
  a.gstride = L->wq_order ? (int)(L->N / 16) : 1;
  if (!fill_part_table(a.pt, L->n_parts, L->part_cols, 16)) return fail(PARO_ERR_INVALID, "bad partition table");
  a.ksplit = 1;
  a.gps = a.G;
  a.partial = nullptr;
  a.block_expert = block_expert;
  a.n_experts = n_experts;
  a.wq_estride = wq_stride_bytes / 16;
  a.sz_estride = sz_stride_bytes / 4;
  dim3 grid((unsigned)a.pt.cbs, (unsigned)(padded_rows / block_rows), 1u);
  rc = launch_gemm3(a, L->act_dtype, grid, (hipStream_t)stream, 0, 128 / quant_group(L->group_size), block_rows / 32);
  if (rc != PARO_OK) return rc;
  return check_launch("paro_w4a16_gemm_grouped");
}

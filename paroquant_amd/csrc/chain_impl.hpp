// Decode-chain GEMV for gfx950: INT4 dequant + GEMV on PRE-ROTATED activations, with the NEXT linear's pairwise
// rotation applied by the workgroup that owns the finished outputs (producer-side rotation).
//
// Why a second GEMV family next to gemv_impl.hpp: in the fused kernel every one of the 150..250 workgroups of a launch
// redoes the whole rotation of x (8 dependent cross-lane stages per 128-channel group, a 3 KiB schedule per group
// through the CU's vector-memory path -- more L2->CU bytes than INT4 weights at 1..2 tiles per wave) before it can
// consume its first tile; the per-wave chain [schedule wait -> 8 stages -> tiles] x units is what keeps the batch-1
// launches at 0.14..0.38 of the HBM roofline (profiles/r02_gemv_timeline.txt).  The rotation is
// block-diagonal over 128 channels, so the launch that PRODUCES an activation vector can rotate each 128-column
// block it owns ONCE, with the consumer's schedule, and hand over x already rotated:
//
//     reference per linear      y = rotate(x * cs; pairs, theta) @ dequant(W)          (modules.py:57-71, plugin.py:281-311)
//     here, linear i            y_i = x_rot_i @ dequant(W_i)         (+ rstd, bias, residual)
//                               x_rot_{i+1}[p'] = rotate_{i+1,p'}(act(y_i) * cs_{i+1,p'})   in y_i's epilogue
//
// Mapping:
//   * a workgroup owns one 128-column block (8 tiles of 16 columns) x one K-slice of `gps` groups (grid = blocks x
//     ksplit); with PAIR, the gate block and the up block of the same 128 channels (waves 0..W/2-1 / W/2..W-1), so that
//     silu(gate) * up exists inside one workgroup;
//   * unit = (group, 8 tiles): x fragments (from the rotated vector), 8 KiB of INT4 tiles and 2 x 16 B of scale / zero
//     words are requested together, double-buffered; nothing in front of the first request but the argument fetch;
//   * the waves of a block reduce through LDS; K-slices are combined with data-tagged 8-byte granules {tag, fp32}
//     (write-through store, sc1 poll) where tag = (block, per-block epoch): no re-arm store, a late or stale granule can
//     never be mistaken for this launch's; the epoch lives in the workspace's counter area and is
//     advanced by the block's reducer;
//   * the block owner (the only workgroup at ksplit 1, else the last K-slice) finishes the outputs -- RMSNorm scalar of
//     the INPUT (sum of squares delivered per block by whoever produced x), bias, residual, one rounding, store; the
//     block's own sum of squares for the next norm -- stages them in LDS, and its waves run the consumer's Givens
//     stages in registers (one ds_bpermute per stage; the schedule words of paro_pack_rotation, always eight
//     straight-line stages: schedules of krot < 8 are identity-padded) for every consumer partition and row chunk.
#pragma once
#include <cstddef>
#include <type_traits>

#include "common.hpp"

namespace paro {

template <typename T>
using CGP = const __attribute__((address_space(1))) T*;
template <typename T>
using WGP = __attribute__((address_space(1))) T*;

constexpr int kChainMaxRows = 16;
constexpr int kChainEpochWords = PARO_WS_STATUS_OFFSET / 4;   // per-block epochs live in the workspace's counter area
constexpr unsigned kChainBlockBits = 12;                       // tag = epoch << 12 | block

struct ChainArgs {
  // ---- needed before the first loads
  const u32x4* wq;
  const unsigned* sz;
  const unsigned short* x;        // rotated activations [n_parts][rows][K]
  int G, T;                       // K / 128, N / 16
  int order;                      // tile order of wq (paro_linear_t.wq_order)
  int rows, ksplit, gps, szrow;   // szrow: words per group row of the scale / zero array
  int pb[PARO_MAX_PARTS - 1];     // first 128-column block of partitions 1..7 (INT_MAX beyond the last)
  int blk0, up_off;               // PAIR: first gate block, blocks between a gate block and its up block
  // ---- epilogue
  unsigned short* y;              // [rows][N] or null
  unsigned long long* slabs;      // K-split granules [ksplit - 1][rows][N]
  unsigned* epochs;               // [kChainEpochWords] (+ the status word behind them)
  const unsigned short* bias;
  const unsigned short* residual; // [rows][N]
  const float* ssq_in;            // [rows][ssq_in_n] sums of squares of the un-normalised input, or null
  float* ssq_out;                 // [rows][N / 128] or null
  int ssq_in_n;
  float inv_norm_dim, eps;
  int N;
  // ---- the consumer's rotation (null nrot: none)
  const unsigned* nrot;           // paro_pack_rotation words of the consumer [P'][K' / 128][3][64][4]
  const unsigned short* ncs;      // consumer channel scales [P'][K']
  unsigned short* nx;             // rotated output [P'][rows][K']
  int np, Gn;                     // consumer partitions, K' / 128
  int nblk0;                      // first block of this layer's output that the consumer reads (channel 0 of K')
  int act;                        // 0 identity, 1 silu(gate) * up, 2 gelu_tanh(gate) * up (PAIR)
  unsigned long long* dbg;        // PARO_CHAIN_DIAG builds: 16 phase stamps per workgroup (tools/chain_harness.cpp), else null
};

// one wavefront rotates <= 4 rows of one 128-channel group held as (channel 2l, channel 2l + 1) per lane
template <typename AT, int R>
struct GivensRegs {
  typedef Act<AT> A;
  float sa[R], sb[R];
  u32x4 rc[3];
  float Pf[8], Qf[8];
  int srcl[8];
  // integer -> float converts and source-lane bytes of all stages, off the stage chain (call once the words have arrived)
  __device__ __forceinline__ void prepare() {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const unsigned w = rc[t >> 2][t & 3];
      const unsigned sw = rc[2][t >> 2];
      Pf[t] = (float)(int)(short)(w & 0xffffu);
      Qf[t] = (float)((int)w >> 16);
      srcl[t] = (int)((sw >> (8 * (t & 3))) & 0xffu);
    }
  }
  __device__ __forceinline__ void load(CGP<unsigned> rot, unsigned group_index, int lane) {
    CGP<u32x4> rp = (CGP<u32x4>)rot + (group_index * 192u + (unsigned)lane);
#pragma unroll
    for (int q = 0; q < 3; ++q) rc[q] = rp[q * 64];
  }
  __device__ __forceinline__ void seed(int r, float x0, float x1, unsigned csv) {
    sa[r] = x0 * (f16_bits_to_f32(csv & 0xffffu) * 0x1p-63f);
    sb[r] = x1 * (f16_bits_to_f32(csv >> 16) * 0x1p-63f);
  }
  __device__ __forceinline__ void stages() {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float P = Pf[t], Q = Qf[t];
      const int src = srcl[t];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float keep = __builtin_fmaf(P, sa[r], Q * sb[r]);
        const float give = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
        sb[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, give)));
        sa[r] = keep;
      }
    }
  }
  // one stage of row 0 (callers that interleave the stage chains of several groups: engine2.hip)
  __device__ __forceinline__ void stage(int t) {
    const float keep = __builtin_fmaf(Pf[t], sa[0], Qf[t] * sb[0]);
    const float give = __builtin_fmaf(Pf[t], sb[0], -(Qf[t] * sa[0]));
    sb[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcl[t], __builtin_bit_cast(int, give)));
    sa[0] = keep;
  }
  // last checkpoint stage of row 0 into registers: the two rounded activations and their byte offsets inside the group's 128 halves
  __device__ __forceinline__ void finish_vals(unsigned short& h1, unsigned short& h2, unsigned& oa, unsigned& ob) {
    const unsigned w0 = rc[2][2], w1 = rc[2][3];
    const float P = (float)(int)(short)(w0 & 0xffffu) * 0x1p-63f, Q = (float)((int)w0 >> 16) * 0x1p-63f;
    oa = w1 & 0xfeu;
    ob = (w1 >> 8) & 0xfeu;
    const float o1 = __builtin_fmaf(P, sa[0], Q * sb[0]);
    const float d = __builtin_fmaf(P, sb[0], -(Q * sa[0]));
    h1 = A::from_f32(o1);
    h2 = A::from_f32(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, d) ^ (w1 & 0x80000000u)));
  }
  // last checkpoint stage; ONE rounding; scattered into row buffers `xs` (128 halves per row) at the channels' places
  __device__ __forceinline__ void finish(unsigned short* xs) {
    const unsigned w0 = rc[2][2], w1 = rc[2][3];
    const float P = (float)(int)(short)(w0 & 0xffffu) * 0x1p-63f, Q = (float)((int)w0 >> 16) * 0x1p-63f;   // 2^(49 - 14 * 8)
    const unsigned oa = w1 & 0xfeu, ob = (w1 >> 8) & 0xfeu;
    const unsigned flip = w1 & 0x80000000u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float o1 = __builtin_fmaf(P, sa[r], Q * sb[r]);
      const float d = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
      const float o2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d) ^ flip);
      *(unsigned short*)((unsigned char*)(xs + r * 128) + oa) = A::from_f32(o1);
      *(unsigned short*)((unsigned char*)(xs + r * 128) + ob) = A::from_f32(o2);
    }
  }
};

__device__ __forceinline__ float wave_sum_dpp(float v) {   // total ends up in lane 63 (gemv_impl.hpp uses the same six adds)
  auto dpp_add = [](float a, auto ctrl_tag, auto mask_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
    return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, MASK, 0xf, false));
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});
  v = dpp_add(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});
  v = dpp_add(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});
  v = dpp_add(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});
  v = dpp_add(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  v = dpp_add(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return v;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for every global load
// and store in flight (the consumer's rotation schedule requested under the reduction, the y stores under the staging)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename AT, int MB, int WAVES, bool PAIR>
__global__ __launch_bounds__(WAVES * 64) void chain_kernel(const ChainArgs a_in) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int THREADS = WAVES * 64;
  constexpr int MR = MB <= 4 ? 1 : (MB <= 8 ? 2 : 4);   // accumulator registers per tile (batch rows ride in the MFMA M dimension)
  constexpr int NH = PAIR ? 2 : 1;                      // column blocks per workgroup
  constexpr int HW = WAVES / NH;                        // waves per column block
  constexpr int EIT = (MB * 128 + THREADS - 1) / THREADS;   // outputs per thread and block
  // Output `it` of a thread: the MR accumulator rows of one MFMA lane are MR consecutive `it` -- thread t owns (row quad q, column c) =
  // (qd >> 7, qd & 127), qd = t + (it / MR) THREADS, and the batch rows q MR + (it % MR) -- so that the waves' partial sums of all MR rows
  // are ONE LDS access per wave in the reduction (16 rows: 8 writes + 8 reads of 16 bytes per lane instead of 32 + 32 dwords; the
  // reduction was 1.4 .. 2.0 us of an 8 .. 13 us launch at 16 rows, profiles/r04_chain_timeline_rows16.txt)
  static_assert(EIT % MR == 0, "a thread's outputs come in groups of MR rows");
  typedef float vecMR __attribute__((ext_vector_type(MR == 1 ? 1 : MR)));
  constexpr int RR = MB < 2 ? MB : 2;                   // rows per rotation task: two chains interleave well, more rows per task only lengthen its stage chain (tasks are dealt to the waves)
  constexpr int RED_FLOATS = WAVES * 8 * MR * 64;
  constexpr int Z_FLOATS = MB * 128;
  constexpr int XS_HALVES = WAVES * RR * 128;
  constexpr int LDS_BYTES = RED_FLOATS * 4 + Z_FLOATS * 4 + XS_HALVES * 2 + MB * 4 /* rstd */ + MB * 2 * 4 /* ssq halves */ + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  float* red = (float*)lds;
  float* zs = red + RED_FLOATS;
  unsigned short* xsb = (unsigned short*)(zs + Z_FLOATS);
  float* rstd_s = (float*)(xsb + XS_HALVES);
  float* ssq_s = rstd_s + MB;

  // Every argument the kernel needs before its epilogue is pinned into scalar registers HERE: the compiler otherwise
  // fetches argument fields where they are first used -- five dependent scalar-load round trips (~0.1 us each) sat in
  // front of the first tile request in the first build of this kernel.
  ChainArgs a = a_in;
  {
    unsigned long long pw = (unsigned long long)a.wq, ps = (unsigned long long)a.sz, px = (unsigned long long)a.x,
                       pe = (unsigned long long)a.epochs, pq = (unsigned long long)a.ssq_in, pr = (unsigned long long)a.nrot,
                       pc = (unsigned long long)a.ncs;
    asm volatile("" : "+s"(pw), "+s"(ps), "+s"(px), "+s"(pe), "+s"(pq), "+s"(pr), "+s"(pc));
    a.wq = (const u32x4*)pw; a.sz = (const unsigned*)ps; a.x = (const unsigned short*)px; a.epochs = (unsigned*)pe;
    a.ssq_in = (const float*)pq; a.nrot = (const unsigned*)pr; a.ncs = (const unsigned short*)pc;
    asm volatile("" : "+s"(a.G), "+s"(a.T), "+s"(a.order), "+s"(a.rows), "+s"(a.ksplit), "+s"(a.gps), "+s"(a.szrow));
    asm volatile("" : "+s"(a.pb[0]), "+s"(a.pb[1]), "+s"(a.pb[2]), "+s"(a.pb[3]), "+s"(a.pb[4]), "+s"(a.pb[5]), "+s"(a.pb[6]));
    asm volatile("" : "+s"(a.blk0), "+s"(a.up_off), "+s"(a.ssq_in_n), "+s"(a.np), "+s"(a.Gn), "+s"(a.nblk0), "+s"(a.N));
    // ... and the epilogue's: fetched where they are used they were three more dependent round trips in the owner's tail
    unsigned long long py = (unsigned long long)a.y, pl = (unsigned long long)a.slabs, pb = (unsigned long long)a.bias,
                       pd = (unsigned long long)a.residual, po = (unsigned long long)a.ssq_out, pn = (unsigned long long)a.nx;
    asm volatile("" : "+s"(py), "+s"(pl), "+s"(pb), "+s"(pd), "+s"(po), "+s"(pn));
    a.y = (unsigned short*)py; a.slabs = (unsigned long long*)pl; a.bias = (const unsigned short*)pb;
    a.residual = (const unsigned short*)pd; a.ssq_out = (float*)po; a.nx = (unsigned short*)pn;
    unsigned f0 = __builtin_bit_cast(unsigned, a.inv_norm_dim), f1 = __builtin_bit_cast(unsigned, a.eps);
    asm volatile("" : "+s"(f0), "+s"(f1), "+s"(a.act));
    a.inv_norm_dim = __builtin_bit_cast(float, f0); a.eps = __builtin_bit_cast(float, f1);
  }
#ifdef PARO_CHAIN_DIAG
  unsigned long long ts[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) ts[k] = 0;
  ts[0] = __builtin_amdgcn_s_memtime();
  ts[10] = __builtin_amdgcn_s_memrealtime();
#define CHAIN_STAMP(k, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); ts[k] = t_; } while (0)
#else
#define CHAIN_STAMP(k, dep) do { } while (0)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = PAIR ? wave / HW : 0;
  const int hw = PAIR ? wave % HW : wave;
  const int bx = blockIdx.x, ks = blockIdx.y;
  const int blk = PAIR ? a.blk0 + bx + half * a.up_off : bx;   // 128-column block of this wave
  int p = 0;
#pragma unroll
  for (int q = 0; q < PARO_MAX_PARTS - 1; ++q) p += (blk >= a.pb[q]) ? 1 : 0;

  const int K = a.G * 128;
  const int rows = a.rows;
  const int g_begin = ks * a.gps;
  const int g_end = min(a.G, g_begin + a.gps);
  const int n_local = g_end - g_begin;
  const int my_count = hw < n_local ? (n_local - hw + HW - 1) / HW : 0;
  const int g_first = hw < n_local ? g_begin + hw : a.G - 1;   // a wave without work runs one clamped unit and discards it

  const int n = lane & 15, mq = lane >> 4;
  const int mrow = lane & 15;
  const int brow = (mrow >> 2) * MR + (mrow & 3);              // batch row carried by MFMA row `mrow`
  const int xrow = min(brow, rows - 1);
  const int tstride = a.order ? 1 : a.G, gstride = a.order ? a.T : 1;
  CGP<u32x4> wq_p = (CGP<u32x4>)a.wq;
  CGP<unsigned> sz_p = (CGP<unsigned>)a.sz;
  CGP<unsigned short> x_p = (CGP<unsigned short>)a.x + (unsigned)(p * rows * K);
  const int tile0 = blk * 8;

  struct UBuf {
    u32x4 q[8];
    u32x4 szv[2];
    u32x4 xa[4];
  };
  auto load_u = [&](UBuf& b, int g) {
    // every lane loads (row clamped into range): MFMA rows that carry no batch row compute a duplicate of a real one,
    // which the epilogue never reads -- no mask, no branch (a load inside a branch makes the compiler's vmcnt bookkeeping
    // give up)
#pragma unroll
    for (int i = 0; i < 4; ++i) b.xa[i] = *(CGP<u32x4>)(x_p + (unsigned)(xrow * K + g * 128 + 32 * i + 8 * mq));
#pragma unroll
    for (int j = 0; j < 8; ++j)
      b.q[j] = __builtin_nontemporal_load(wq_p + ((unsigned)((tile0 + j) * tstride + g * gstride) * 64u + (unsigned)lane));
    CGP<unsigned> sp = sz_p + ((unsigned)g * (unsigned)a.szrow + (unsigned)(((tile0 >> 2) * 16 + n) * 4));
    b.szv[0] = *(CGP<u32x4>)sp;
    b.szv[1] = *(CGP<u32x4>)(sp + 64);
  };

  UBuf uc, un;
  load_u(uc, g_first);
  __builtin_amdgcn_sched_barrier(0);
#ifdef PARO_CHAIN_DIAG
  ts[1] = __builtin_amdgcn_s_memtime();
  auto dump = [&]() {
    if (a.dbg && tid == 0) {
      ts[8] = __builtin_amdgcn_s_memtime();
      ts[11] = __builtin_amdgcn_s_memrealtime();
      ts[12] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
      unsigned long long* d = a.dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
#pragma unroll
      for (int k = 0; k < 16; ++k) d[k] = ts[k];
    }
  };
#endif

  const bool owner = ks == a.ksplit - 1;
  const bool rotate_next = a.nrot != nullptr;
  // the consumer group of this workgroup's block, and whether the consumer reads it at all
  const int cg = (PAIR ? a.blk0 + bx : bx) - a.nblk0;
  const bool consumed = rotate_next && cg >= 0 && cg < a.Gn;
  // rotation tasks (consumer partition, chunk of RR rows) are dealt to the waves round-robin; the first task's schedule
  // and channel scales are requested HERE, behind the first unit's tiles: they are first-touch misses like everything
  // else after a kernel boundary, and the whole main loop hides them (13 registers per lane)
  const int n_chunks = (rows + RR - 1) / RR;
  const int n_tasks = consumed && owner ? a.np * n_chunks : 0;
  GivensRegs<AT, RR> gr;
  unsigned csv = 0;
  {
    const int t0 = min(wave, max(n_tasks - 1, 0));        // clamped: the requests stay unconditional
    const int pp = n_tasks ? t0 / n_chunks : 0, cgc = n_tasks ? cg : 0;
    CGP<unsigned> rsrc = n_tasks ? (CGP<unsigned>)a.nrot : (CGP<unsigned>)a.sz;   // any readable address when there is no task
    gr.load(rsrc, n_tasks ? (unsigned)(pp * a.Gn + cgc) : 0u, lane);
    csv = *(CGP<unsigned>)((n_tasks ? (CGP<unsigned short>)a.ncs : (CGP<unsigned short>)a.sz) + (unsigned)(n_tasks ? (pp * a.Gn + cgc) * 128 + 2 * lane : 2 * lane));
  }
  __builtin_amdgcn_sched_barrier(0);
  // the input's RMSNorm scalar: this wave's rows of the per-block sums of squares, requested now, reduced in the epilogue
  constexpr int SQR = (MB + WAVES - 1) / WAVES;
  float sq_in[SQR][2];
#pragma unroll
  for (int i = 0; i < SQR; ++i) {
    sq_in[i][0] = sq_in[i][1] = 0.f;
    const int b = wave + i * WAVES;
    if (a.ssq_in && b < rows) {
      CGP<float> sp = (CGP<float>)a.ssq_in + (unsigned)(b * a.ssq_in_n);
      if (lane < a.ssq_in_n) sq_in[i][0] = sp[lane];
      if (lane + 64 < a.ssq_in_n) sq_in[i][1] = sp[lane + 64];
    }
  }
  // K-split tag of this block: (block, epoch + 1); the epoch word is advanced by the block's reducer at its very end
  unsigned tag[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) tag[h] = 0;
  if (a.ksplit > 1) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int bh = PAIR ? a.blk0 + bx + h * a.up_off : bx;
      // through the scalar cache (invalidated at kernel start; the word was written by the previous launch's reducer): a
      // vector load here would put a vmcnt(0) wait for ALL of the first unit's tiles in front of the next unit's requests
      unsigned e = *(const __attribute__((address_space(4))) unsigned*)(a.epochs + bh) + 1u;
      if ((e & 0xfffffu) == 0u) e += 1u;   // tag 0 is "never written"
      tag[h] = (e << kChainBlockBits) | ((unsigned)bh & ((1u << kChainBlockBits) - 1u));
    }
  }

  float acc[8][MR];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[j][r] = 0.f;

  const typename A::Unpack upk = A::unpack_consts();
  auto consume = [&](const UBuf& t) {
    f32x4 sx = {0.f, 0.f, 0.f, 0.f}, so = {0.f, 0.f, 0.f, 0.f};
    {
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const u32x4 offs = {A::kOffFrag0, A::kOffFrag1, A::kOffFrag0, A::kOffFrag1};
      const vec8 ob = __builtin_bit_cast(vec8, ones);
      const vec8 fb = __builtin_bit_cast(vec8, offs);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sx = A::mfma(__builtin_bit_cast(vec8, t.xa[i]), ob, sx);
        so = A::mfma(__builtin_bit_cast(vec8, t.xa[i]), fb, so);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned w4[4];
        A::unpack_fast(t.q[j][i], w4, upk);
        const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
        d = A::mfma(__builtin_bit_cast(vec8, t.xa[i]), __builtin_bit_cast(vec8, wv), d);
      }
      const unsigned szw = t.szv[j >> 2][j & 3];
      const float s = f16_bits_to_f32(szw & 0xffffu);
      const float zf = f16_bits_to_f32(szw >> 16);
#pragma unroll
      for (int r = 0; r < MR; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-zf, sx[r], d[r] - so[r]), acc[j][r]);
    }
  };

  for (int i = 0; i + 1 < my_count; ++i) {
    load_u(un, g_begin + hw + (i + 1) * HW);
    consume(uc);
#ifdef PARO_CHAIN_DIAG
    if (ts[2] == 0) CHAIN_STAMP(2, acc[0][0]);
#endif
    uc = un;
  }
  consume(uc);
  CHAIN_STAMP(3, acc[0][0]);
  if (my_count == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < MR; ++r) acc[j][r] = 0.f;
  }

  if (n_tasks) gr.prepare();   // the schedule words were requested at kernel entry
  // ---- reduce the waves of each block through LDS
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) red[((wave * 8 + j) * 64 + lane) * MR + r] = acc[j][r];   // (MR consecutive floats: merged into one b64 / b128 store)
  // the input's RMSNorm scalar per row
#pragma unroll
  for (int i = 0; i < SQR; ++i) {
    const int b = wave + i * WAVES;
    if (a.ssq_in && b < rows) {   // wave-uniform
      const float v = wave_sum_dpp(sq_in[i][0] + sq_in[i][1]);
      if (lane == 63) rstd_s[b] = __builtin_amdgcn_rsqf(v * a.inv_norm_dim + a.eps);
    }
  }
  lds_barrier();
#ifdef PARO_CHAIN_DIAG
  ts[4] = __builtin_amdgcn_s_memtime();
#endif

  // (b, c) of output `it`; false when the thread has no such output
  auto out_bc = [&](int it, int& b, int& c) -> bool {
    const int qd = tid + (it / MR) * THREADS;
    c = qd & 127;
    b = (qd >> 7) * MR + (it % MR);
    const bool live = b < rows && qd < (MB / MR) * 128;
    if (!live) b = 0;
    return live;
  };
  float v[EIT][NH];
#pragma unroll
  for (int qi = 0; qi < EIT / MR; ++qi) {
    const int qd = tid + qi * THREADS;
    const int c = qd & 127, bq = min(qd >> 7, MB / MR - 1);
    const int src = (((c >> 4) * 64) + bq * 16 + (c & 15)) * MR;       // tile c >> 4, MFMA lane (row quad, column in the tile)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float s[MR];
#pragma unroll
      for (int r = 0; r < MR; ++r) s[r] = 0.f;
#pragma unroll
      for (int w = 0; w < HW; ++w) {
        const vecMR q = *(const vecMR*)(red + (h * HW + w) * 8 * MR * 64 + src);
#pragma unroll
        for (int r = 0; r < MR; ++r) s[r] += q[r];
      }
#pragma unroll
      for (int r = 0; r < MR; ++r) v[qi * MR + r][h] = s[r];
    }
  }

  if (a.ksplit > 1) {
    if (!owner) {
      // producer: ONE write-through 8-byte {tag, fp32 partial} granule per output; no drain, no flag, no fence
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        int b, c;
        if (!out_bc(it, b, c)) continue;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int col = ((PAIR ? a.blk0 + bx + h * a.up_off : bx) << 7) + c;
          const unsigned long long gv = ((unsigned long long)tag[h] << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v[it][h]);
          __hip_atomic_store(a.slabs + ((int64_t)ks * rows + b) * a.N + col, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#ifdef PARO_CHAIN_DIAG
      dump();
#endif
      return;
    }
    // owner: the other slices' partials.  A poll is a ~0.75 us round trip to the memory side (write-through granules do
    // not stay in an L2), so every slice's poll is in flight before any is examined (up to BQ per output, i.e. ONE round
    // trip for up to 16 slices at batch 1), and a batch is re-polled as a whole until it is complete.
    // (measured, profiles/r03_chain_timeline.txt: a poll is a ~2000-cycle round trip; keeping three staggered rounds of every
    // granule in flight to catch the arrival earlier made it worse -- 30 write-through-line reads per lane queue behind each
    // other -- so: one round, whole batch re-polled until complete)
    constexpr int BQ = (16 / (EIT * NH)) < 4 ? 4 : (16 / (EIT * NH));
    const int nsp = a.ksplit - 1;
    for (int s0 = 0; s0 < nsp; s0 += BQ) {
      unsigned long long gq[EIT][NH][BQ];
      bool done = false;
      for (int spin = 0; !done; ++spin) {
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
          int b, c;
          out_bc(it, b, c);                      // (a thread without this output polls row 0 of its column: in range, never consumed)
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const int col = ((PAIR ? a.blk0 + bx + h * a.up_off : bx) << 7) + c;
#pragma unroll
            for (int q = 0; q < BQ; ++q) {
              const int s = min(s0 + q, nsp - 1);
              gq[it][h][q] = __hip_atomic_load(a.slabs + ((int64_t)s * rows + b) * a.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        done = true;
#pragma unroll
        for (int it = 0; it < EIT; ++it)
#pragma unroll
          for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int q = 0; q < BQ; ++q) done = done && (unsigned)(gq[it][h][q] >> 32) == tag[h];
        if (done) break;
        if (spin >= (1 << 16)) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int it = 0; it < EIT; ++it)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int q = 0; q < BQ; ++q) {
            if (s0 + q >= nsp) continue;
            if (done) {
              v[it][h] += __builtin_bit_cast(float, (unsigned)gq[it][h][q]);
            } else {
              // never a silently wrong sum: NaN + the sticky status word (unreachable while the grid is resident)
              a.epochs[kChainEpochWords] = PARO_WS_STATUS_GIVEUP;
              v[it][h] = __builtin_nanf("");
            }
          }
    }
  }

  CHAIN_STAMP(5, v[0][0]);
  // ---- the finished outputs of the block(s): norm scalar, bias, residual, one rounding, store; stage for the rotation
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    int b, c;
    const bool live = out_bc(it, b, c);
    float yf[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int col = ((PAIR ? a.blk0 + bx + h * a.up_off : bx) << 7) + c;
      float t = v[it][h];
      if (a.ssq_in) t *= rstd_s[b];
      if (a.bias) t += A::to_f32(((CGP<unsigned short>)a.bias)[col]);
      if (a.residual) t += A::to_f32(((CGP<unsigned short>)a.residual)[(unsigned)(b * a.N + col)]);
      const unsigned short yb = A::from_f32(t);
      if (live && a.y) ((WGP<unsigned short>)a.y)[(unsigned)(b * a.N + col)] = yb;
      yf[h] = A::to_f32(yb);
    }
    float z = yf[0];
    if constexpr (PAIR) {
      // silu(gate) * up -- or gelu_tanh(gate) * up: the same g u / (1 + exp(-a)) with a = 2 sqrt(2 / pi) (g + 0.044715 g^3) --
      // in fp32 on the rounded projections
      if (a.act != 0) {
        const float g = yf[0];
        const float arg = a.act == 2 ? 1.5957691216057308f * __builtin_fmaf(0.044715f * g * g, g, g) : g;
        z = g * yf[1] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * arg));
      }
    }
    if (live) zs[b * 128 + c] = z;
    if (a.ssq_out) {
      // the block's sum of squares per row: a wave covers 64 consecutive columns of one row (THREADS is a multiple of 128)
      const float s = wave_sum_dpp(live ? yf[0] * yf[0] : 0.f);
      if (lane == 63 && live) ssq_s[b * 2 + ((c >> 6) & 1)] = s;
    }
  }
  lds_barrier();
  if (a.ssq_out && tid < rows) ((WGP<float>)a.ssq_out)[(unsigned)(tid * (a.N >> 7) + bx)] = ssq_s[tid * 2] + ssq_s[tid * 2 + 1];

#ifdef PARO_CHAIN_DIAG
  ts[6] = __builtin_amdgcn_s_memtime();
#endif
  // ---- the consumer's rotation of this block, one (partition, row chunk) task per wave at a time
  unsigned short* xs = xsb + wave * RR * 128;
  for (int task = wave; task < n_tasks; task += WAVES) {
    const int pp = task / n_chunks, ch = task % n_chunks;
    if (task != wave) {
      gr.load((CGP<unsigned>)a.nrot, (unsigned)(pp * a.Gn + cg), lane);
      csv = *(CGP<unsigned>)((CGP<unsigned short>)a.ncs + (unsigned)((pp * a.Gn + cg) * 128 + 2 * lane));
      gr.prepare();
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int b = min(ch * RR + r, rows - 1);
      const f32x2 zz = *(const f32x2*)(zs + b * 128 + 2 * lane);
      gr.seed(r, zz.x, zz.y, csv);
    }
    gr.stages();
    gr.finish(xs);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int b = ch * RR + r;
      if (b < rows)
        *(WGP<unsigned>)((WGP<unsigned short>)a.nx + (unsigned)((pp * rows + b) * (a.Gn * 128) + cg * 128 + 2 * lane)) =
            *(const unsigned*)(xs + r * 128 + 2 * lane);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.ksplit > 1 && tid == 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int bh = PAIR ? a.blk0 + bx + h * a.up_off : bx;
      a.epochs[bh] = tag[h] >> kChainBlockBits;
    }
  }
#ifdef PARO_CHAIN_DIAG
  ts[7] = __builtin_amdgcn_s_memtime();
  ts[9] = (unsigned long long)n_tasks | ((unsigned long long)owner << 32);
  dump();
#endif
}

constexpr int PARO_ERR_NOT_RESIDENT_CHAIN = -100;
int device_cu_count();

template <auto Kern, int THREADS>
int chain_launch_checked(const ChainArgs& a, dim3 grid, hipStream_t st) {
  if (a.ksplit > 1) {
    static int per_cu = -1;
    if (per_cu < 0) {
      int v = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, Kern, THREADS, 0) != hipSuccess || v < 1) v = 1;
      per_cu = v;
    }
    const long long cap = (long long)per_cu * device_cu_count();
    if ((long long)grid.x * grid.y > cap)
      return fail(PARO_ERR_NOT_RESIDENT_CHAIN, "K-split grid of %u x %u workgroups exceeds the %lld that are resident at once", grid.x,
                  grid.y, cap);
  }
  hipLaunchKernelGGL(Kern, grid, dim3(THREADS), 0, st, a);
  return PARO_OK;
}

template <typename AT, int MB>
int chain_launch_mb(const ChainArgs& a, int waves, bool pair, dim3 grid, hipStream_t st) {
  if (pair) {
    if (waves == 8) return chain_launch_checked<chain_kernel<AT, MB, 8, true>, 512>(a, grid, st);
    if (waves == 4) return chain_launch_checked<chain_kernel<AT, MB, 4, true>, 256>(a, grid, st);
  } else {
    if (waves == 8) return chain_launch_checked<chain_kernel<AT, MB, 8, false>, 512>(a, grid, st);
    if (waves == 4) return chain_launch_checked<chain_kernel<AT, MB, 4, false>, 256>(a, grid, st);
  }
  return fail(PARO_ERR_UNSUPPORTED, "chain GEMV: %d waves per workgroup is not built", waves);
}

// one object per (activation type, row class): chain_inst.hip
#define PARO_DECL_CHAIN(T) \
  int launch_chain_##T##_m1(const ChainArgs&, int, bool, dim3, hipStream_t); \
  int launch_chain_##T##_m4(const ChainArgs&, int, bool, dim3, hipStream_t); \
  int launch_chain_##T##_m8(const ChainArgs&, int, bool, dim3, hipStream_t); \
  int launch_chain_##T##_m16(const ChainArgs&, int, bool, dim3, hipStream_t);
PARO_DECL_CHAIN(f16)
PARO_DECL_CHAIN(bf16)
#undef PARO_DECL_CHAIN

}  // namespace paro

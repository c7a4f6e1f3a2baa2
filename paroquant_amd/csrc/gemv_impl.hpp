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
//     group, so the wave rotates ITS OWN slice of x in its registers (one cross-lane fetch per
//     stage, see the stage loop) with no workgroup barrier.  All 8 stages' coefficients (cos / sin words that also carry the LDS offsets of the
//     pair) arrive as four coalesced 16-byte loads per lane (paro_pack_rotation), requested BEFORE the unit's INT4 tiles so that waiting for them
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
//     write-through (sc1) store per output and exit; the last split polls them (sc1 loads, the first poll of
//     every split issued before any is examined; bounded spin), re-arms them, and writes y exactly once --
//     no fences, no tickets, one round trip.
//   * kernel start: the arguments a wave needs before its first loads are a hand-packed 128-byte block at kernarg
//     offset 0 (GemvHot) fetched with two s_load_dwordx16 + one wait, the partition lookup is a dozen scalar
//     instructions (s_addc chain + s_movrels); see the note at GemvHot.
#pragma once
#include <cstddef>
#include <type_traits>

#include "common.hpp"
#include "attn_impl.hpp"

namespace paro {

// Kernel-argument bytes 0..127: EVERYTHING a wave needs before its first global loads are out, fetched by one pair of
// s_load_dwordx16 and one wait (the compiler's own argument fetch came out as three to four DEPENDENT scalar-load
// round trips at kernel start).  The second half is the partition table in a form the lookup can index with
// s_movrels: the scalar prologue is executed by EVERY wave and all waves of a CU share one scalar unit -- with 16
// waves per CU, each scalar instruction in front of the first loads costs the workgroup ~16 cycles (the select
// chains the compiler made of the lookup were ~90 of the ~220 scalar instructions there).
struct alignas(16) GemvHot {
  const u32x4* wq;               // dwords 0..1
  const unsigned* sz;            // 2..3
  const unsigned* rot;           // 4..5
  const unsigned short* cs;      // 6..7
  const unsigned short* x;       // 8..9   [rows][K], or pre-rotated [nparts][rows][K] when PREROT
  unsigned g_t;                  // 10     G = K / 128 | T = N / 16 << 16   (tile (t, g) = chunk t * G + g, or g * T + t when order)
  unsigned meta;                 // 11     rows (7 bits) | parts_out << 7 | krot << 8 | ksplit << 16 | skew << 24 | prio << 25 | prologue << 26 | experts << 28 | order << 29
  unsigned gps_tsz;              // 12     groups per K-split | scale/zero tiles per group row << 16
  unsigned residual_lo, residual_hi;  // 13, 14 pointer to [rows][N] added to the output, or null (FUSED); two dwords: offset 52 is not pointer-aligned
  unsigned xstride;              // 15     elements between rows of x (SiLU*mul: x = [rows][2 K], gate then up)
  int cbs[7];                    // 16..22 first column block of partition q = 1..7 at [q - 1]; INT_MAX beyond the last
  unsigned ent[9];               // 23..31 partition q = 0..8: first tile | first scale/zero tile << 16 (q = nparts: the totals)
};
static_assert(sizeof(GemvHot) == 128, "the hot argument block is two s_load_dwordx16");

struct GemvArgs {
  GemvHot hot;                   // must stay first: the kernel reads it at kernarg offset 0
  // ---- cold: first used once the first global loads are in flight
  float eps;                     // RMSNorm epsilon (FUSED)
  const unsigned short* bias;
  unsigned short* y;
  unsigned long long* slabs;     // K-split granules {tag << 32 | fp32 bits}: [ksplit - 1][rows][N]; parts_out launches: float [N][4] (the
                                 // caller's partial sums, no hand-off)
  unsigned* counters;            // K-split epoch words
  // deferred K-split reduction, consumer side (FUSED | 8): one 16-byte pair, fetched with one scalar load at kernel entry
  alignas(16) const float* parts_in;   // [K][4] fp32: the producer's partial sums of x's missing term, or null
  unsigned short* x_out;               // the completed x [K] in the activation type (unsplit launches), or null
  // expert slots (FUSED instantiations only; paro_w4a16_gemv_experts): blockIdx.z = slot, the slot's expert id
  // is read from DEVICE memory; all experts share the rotation (cli/convert.py:280-379, mlx/modules.py:159-212)
  const int* expert_idx;         // [slots] or null
  long long wq_estride, sz_estride;      // bytes between experts in wq / sz
  long long x_sstride, y_sstride;        // elements between slots of x (slot / x_div) and of y
  int x_div;
  int n_experts;                         // ids outside [0, n_experts) read expert 0 and give NaN outputs (checked on the device: graph replays too)
  // shared rotation (FUSED | 32, mode 3): every (partition, group) of x is rotated ONCE per launch -- by one wave of the grid -- and
  // handed to the workgroups that multiply by it as {launch tag, two rotated channels} granules: [n_parts][G][rows][64 lanes] x 8 bytes
  unsigned long long* xg;
  int shr_units;                         // producer tasks: n_parts * G * ceil(rows / 4) (rows <= 4: one per (partition, group))
  int shr_prod_wgs;                      // producer workgroups in front of every grid row: ceil(shr_units / waves)
  int shr_bytes;                         // size of the granule buffer (buffer descriptor bound)
  int shr_self;                          // 1: the hybrid form (FUSED | 64) -- every wave rotates its own first group, the producers the rest
  // all-reduce epilogue (FUSED instantiations, one row; allreduce.hip describes the buffers): the row-parallel partial
  // outputs of the world's ranks are exchanged as {fp32 partial, epoch} granules straight from the output threads
  unsigned char* ar_peer[kArMaxWorld];   // every rank's buffer as mapped in this process, BY VALUE: a pointer fetched from device
                                         // memory per peer is a dependent ~0.5 us round trip in front of every store
  unsigned char* ar_mine;           // == ar_peer[ar_rank]; null = no all-reduce epilogue
  unsigned* ar_state;               // ordinary device memory, zero at creation: [0] gave up once, [kArStateTiles + t] epoch of column tile t
  int ar_world, ar_rank;
  long long ar_slot;                // granules per (set, rank) slot of the fp32 region
  long long ar_off;                 // byte offset of that region inside a buffer
  // attention tail (FUSED | 128, paro_fusion_t.attn_tail): the grid has more rows (blockIdx.y >= ksplit) whose first attn_wgs
  // workgroups run the decode attention of this token (attn_impl.hpp) on the q / k / v this launch's K-slices leave as tagged granules
  AttnArgs attn;
  int attn_wgs;                     // KV heads x position chunks; 0 = no attention tail
  int attn_cbs;                     // column blocks (= gridDim.x): the attention workgroups fill ceil(attn_wgs / attn_cbs) more grid rows
  // ---- host side only (instantiation choice; the kernel never reads these)
  int rows, ksplit, prologue;
  int parts_out;                 // deferred K-split reduction (paro_fusion_t, v12): this launch leaves partial sums
  int attn_in;                   // x is a split attention launch's slots (paro_fusion_t.attn_in, v14): parts_in / x_out carry its two pointers
  int shared_rot;                // mode 3: the rotation is shared inside the launch (FUSED | 32)
  int qs;                        // quantisation groups per 128-channel span: 1 (group_size 128) or 2 (group_size 64)
  int pd;                        // 1, or a diagnostic build of the M = 1 kernel (11 / 21 / 31 / 41 / 51 / 61)
  int poll_delay;                // 64-cycle sleeps in front of the K-split reducer's first poll (gemv.hip: 4, or 8 for deep K-slices)
};
static_assert(offsetof(GemvArgs, hot) == 0, "hot block at kernarg offset 0");

// host: pack the hot block; false when a table entry does not fit 16 bits
inline bool pack_hot(GemvHot& h, const PartTable& pt, int G, int order, int rows, int krot, int ksplit, int gps, int skew, int prio,
                     int prologue, bool experts, long long xstride, bool parts_out = false) {
  if (rows > 127 || pt.tiles >= 0xffff || pt.tsz >= 0xffff || gps > 0xffff || G > 0xffff || xstride < 0 || xstride > 0xffffffffll) return false;
  h.g_t = (unsigned)G | ((unsigned)pt.tiles << 16);
  h.meta = (unsigned)rows | ((unsigned)krot << 8) | ((unsigned)ksplit << 16) | ((unsigned)(skew != 0) << 24) |
           ((unsigned)(prio != 0) << 25) | ((unsigned)prologue << 26) | ((unsigned)experts << 28) | ((unsigned)(order != 0) << 29) | ((unsigned)parts_out << 7);
  h.gps_tsz = (unsigned)gps | ((unsigned)pt.tsz << 16);
  h.xstride = (unsigned)xstride;
  for (int q = 1; q < PARO_MAX_PARTS; ++q) h.cbs[q - 1] = (q < pt.nparts) ? pt.cb_start[q] : 0x7fffffff;
  for (int q = 0; q <= PARO_MAX_PARTS; ++q) h.ent[q] = (unsigned)pt.tile_start[q] | ((unsigned)pt.szt_start[q] << 16);
  return true;
}

// a pointer into GLOBAL memory (address space 1)
template <typename T>
using GP = const __attribute__((address_space(1))) T*;

#ifndef PARO_SHR_TASK_ROWS
#define PARO_SHR_TASK_ROWS 2
#endif
constexpr int kShrTaskRows = PARO_SHR_TASK_ROWS;   // rows of one producer task of the shared rotation (2 or 4: row pairs are the granule unit)
constexpr int kXhStride = 136;  // halves per fragment row in LDS (128 + 8 pad: 16 rows' b128 reads spread over banks)

// PD: 1 = the shipping kernel; 11 / 21 / 31 / 41 / 51 / 61 / 71 / 81 = diagnostic builds of the M = 1 kernel (make DIAG=1;
// tools/ablate_gemv.py, tools/timeline_gemv.py): 11 skips schedule + stages, 21 also the unpack + MFMA (pure
// stream), 41 fetches the schedule but does not run the stages, 51 runs the stages without the cross-lane
// fetch, 61 exchanges through LDS memory instead of ds_bpermute, 31 records s_memtime phase stamps, 71 / 81 fetch only 2 KiB / 1 KiB of
// the group's 3 KiB schedule (what a smaller schedule format could save on the fetch side).
// FUSED (1: RMSNorm prologue and / or residual epilogue, 2: SiLU*mul prologue (+ residual); | 4: + the all-reduce
// epilogue of a row-parallel shard, one row): the decode-layer
// fusions either side of the linear, in extra instantiations so that the plain kernel's code is
// untouched:
//   prologue RMSNORM   y = GEMV(x) * rsqrt(mean(x^2) + eps): the norm WEIGHT is folded into channel_scales at load
//                      time, the scalar commutes with rotation and matmul.  sum(x^2) costs nothing extra: every
//                      group of K is seeded by exactly one wave of the workgroup (K-split is refused for it).
//   prologue SILU_MUL  x_k = silu(gate_k) * up_k computed in fp32 while seeding the rotation state, from the
//                      gate_up projection's output [rows][2 K] (the MLX MoE path rotates the activation output the
//                      same way before down_proj, mlx/modules.py:204-207).
//   epilogue residual  y += residual[row][col] (the decoder's residual stream), in the final write.
// FUSED | 8 (with FUSED & 3 == 0 or 1, one row): x arrives INCOMPLETE -- the K-split producer in front (o_proj, down_proj) left its
//   `n` <= 4 fp32 partial sums [K][4] instead of reducing them in its launch, and this kernel finishes the sum while it seeds the
//   rotation: x_k = round(base_k + ((p[n-1][k] + p[0][k]) + ... + p[n-2][k])) -- the order and the one rounding of the reducer
//   below, so both routes give the same bits (the producer stores its splits in that order, four slots per channel, unused ones zero).  Column block 0 also stores the completed x (the decoder's residual stream).
// FUSED | 16 (FUSED & 15 == 0, one row): x is the attention output, handed over UN-MERGED by a split attention launch
//   (attn.hip, paro_attn_decode_split): per element four slots' un-normalised outputs [K][4] and per head the slots' maxima and sums
//   [K / head_dim][8]; x_k = sum_c 2^(m_c - M) o_c[k] / sum_c 2^(m_c - M) l_c, one rounding, computed while seeding (attn_merge below).
// QS: quantisation groups per 128-channel rotation span (1: group_size 128, 2: group_size 64 -- two (scale, zero)
// words per tile and column, the tile's first two / last two MFMA k-steps accumulated separately).
template <typename AT, int TPW, int MB, int WAVES, bool PREROT, int PD, int FUSED = 0, int QS = 1>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(const GemvArgs a) {
  constexpr int FMODE = FUSED & 3;          // 0 plain, 1 RMSNorm prologue and / or residual, 2 SiLU*mul prologue (+ residual)
  constexpr bool AREP = (FUSED & 4) != 0;   // + all-reduce epilogue (its own instantiations: the code costs the others ~5 % otherwise)
  constexpr bool PARTS_IN = (FUSED & 8) != 0;   // x = base + the producer's partial sums, completed while seeding (see above)
  constexpr bool ATTN_IN = (FUSED & 16) != 0;    // x = the merge of a split attention launch's slots, completed while seeding
  static_assert(!PARTS_IN || ((FMODE == 0 || FMODE == 1) && MB == 1 && !AREP && !PREROT), "partial sums feed the one-row RMSNorm / plain prologue");
  static_assert(!ATTN_IN || (FMODE == 0 && MB == 1 && !AREP && !PREROT && !PARTS_IN), "attention slots feed the plain one-row kernel");
  // FUSED | 32 (FUSED & 31 == 0; 1..16 rows): SHARED ROTATION.  The in-kernel rotation above is replicated in every workgroup and its VALU cost grows
  // with the rows (8 rows: 1.9x the one-row launch for identical bytes, profiles/r05_rows_boundary.jsonl).  Here every (partition, group)
  // of x is rotated ONCE per launch: unit u = (p, g) belongs to wave u / n_wgs of workgroup u % n_wgs (one producer per CU first), which
  // rotates it in registers exactly as below and publishes it as 8-byte {launch tag, channels 2l | 2l+1} granules with ONE write-through
  // store per lane and row; every wave then GATHERS the groups it multiplies by (one 8-byte load per lane and row, checked against the
  // tag, bounded re-poll) instead of rotating them.  The tag is the hardware's dispatch id of this launch (the AQL packet index: the same
  // in every workgroup, new for every launch and every graph replay -- tools/probes/dispatch_id_probe.hip) mixed with the queue address:
  // no epoch word, no re-arm, a stale or foreign granule is never consumed.  Same values, same rounding as the replicated form (the
  // producer runs the same seed / stage / finish code): the outputs are bit-identical to mode 0.  Needs the whole grid resident (checked).
  constexpr bool SHR = (FUSED & 32) != 0;
  // FUSED | 64 (with 32): HYBRID -- every wave rotates its FIRST group itself, exactly as the replicated form does (that rotation runs
  // under the first tiles' HBM latency and costs nothing), and takes the rest from the producers, which by then have published: no
  // hand-over on the critical path at all.  The producers rotate only the groups that are nobody's first (local index >= WAVES in their
  // K-slice).
  constexpr bool SELF1 = (FUSED & 64) != 0;
  constexpr int SELF = SELF1 ? 1 : 0;
  // FUSED | 128 (with FUSED & 127 == 1 or 9: the qkv projection of the decode harness, one row, partial sums left to the consumer): ATTENTION
  // TAIL.  The decode attention that consumes q / k / v runs in THIS launch: the grid's last rows (blockIdx.y >= ksplit, dispatched behind
  // every projection workgroup) hold the attention's (KV head, position chunk) workgroups.  They request their K / V cache lines at once
  // -- nothing of that depends on this token's q / k / v -- while the projection streams; the K-slices leave their partial sums as
  // 8-byte {fp32, launch tag} granules (write-through, the data is the flag) and the attention polls them: one in-launch hand-over
  // (~1 us) instead of a launch boundary (2.4 us) plus the attention's own cold start.  Same arithmetic as the two launches.
  constexpr bool ATAIL = (FUSED & 128) != 0;
  static_assert(!ATAIL || (((FUSED & 127) == 1 || (FUSED & 127) == 9) && MB == 1 && !PREROT && QS == 1 && (WAVES == 4 || WAVES == 8)),
                "the attention tail rides the one-row RMSNorm-prologue projection");
  static_assert(!SELF1 || SHR, "the hybrid is a form of the shared rotation");
  static_assert(!SHR || ((FUSED & 31) == 0 && !PREROT && MB <= 16), "the shared rotation feeds the plain kernel");
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int DIAG = PD / 10;
  constexpr int MR = MB <= 4 ? 1 : (MB <= 8 ? 2 : 4);  // accumulator registers kept per tile and MFMA row tile
  constexpr int RT = MB <= 16 ? 1 : MB / 16;            // MFMA row tiles: 32 / 64 batch rows, pre-rotated mode only
  constexpr int MRT = MR * RT;
  static_assert(RT == 1 || PREROT, "more than 16 batch rows run on pre-rotated activations");
  // LDS: per wave only the fragment-row block (MB rows + one zero row) that transposes the rotated
  // slice into MFMA A-fragment order; the rotation state itself lives in registers.
  constexpr int XH_HALVES = PREROT ? 0 : (((MB + 1) * kXhStride + 7) / 8) * 8;
  constexpr int XH_BYTES = XH_HALVES * 2;
  constexpr int RED_FLOATS = WAVES * TPW * MRT * 64;
  constexpr int WORK_BYTES = WAVES * XH_BYTES;
  constexpr int EX_BYTES = (PD / 10 == 6) ? WAVES * 256 : 0;   // diagnostic exchange slots
  constexpr int SS_BYTES = FMODE ? WAVES * MB * 4 : 0;        // per-wave sum(x^2) partials (RMSNorm prologue)
  constexpr int LDS_BYTES = (WORK_BYTES > RED_FLOATS * 4 ? WORK_BYTES : RED_FLOATS * 4) + EX_BYTES + 16 + SS_BYTES;
  // scale/zero words of a unit: one aligned vector load per 4 tiles when TPW is a power of two, else one
  // dword load per tile (TPW = 3, 5, 6, 7 exist so that wide outputs can be cut into ~256 column blocks)
  constexpr bool SZ_VEC = TPW == 1 || TPW == 2 || TPW == 4 || TPW == 8;
  constexpr int NSZ = TPW <= 4 ? 1 : TPW / 4;
  constexpr int SZW = TPW < 4 ? TPW : 4;
  typedef unsigned SZV __attribute__((ext_vector_type(SZW)));
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (FUSED | 32: the first a.shr_prod_wgs workgroups of every grid row are the rotation's PRODUCERS -- dispatched first -- and the column
  // blocks start behind them; see SHR below)
  const int cb = ((FUSED & 32) != 0) ? (int)blockIdx.x - a.shr_prod_wgs : (int)blockIdx.x, ks = blockIdx.y;
  // The hot argument block (kernarg bytes 0..127) with two back-to-back scalar loads and ONE wait, and the partition
  // lookup of this column block, in one hand-written sequence:  p = #{q in 1..7 : cb >= cbs[q]} (seven compare +
  // add-with-carry pairs), then three indexed scalar reads (s_movrels) of cbs[p], ent[p], ent[p + 1].  The table half
  // of the block lands in fixed registers s[84:99] so that the indexed reads can name them.
  typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
  u32x16 k0;
  int p;
  unsigned p_cb0_u, ent0, ent1;
  unsigned long long cnt_ptr;   // GemvArgs::counters (K-split epoch words), fetched with the same batch
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 pin_ptrs = {0ull, 0ull};   // PARTS_IN: GemvArgs::parts_in, x_out; ATTN_IN: the slots' outputs, the slots' (max, sum) | log2(head_dim)
  {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned m0_save;
    // (PARTS_IN / ATTN_IN: one more request in front of the batch, whose wait covers it -- INSIDE the same statement: as a statement of
    // its own, the compiler took its output registers for valid at once, and under register pressure copied / spilled them between the
    // two statements, i.e. before the load had written them -- a garbage pointer in the bf16 attention-tail build, found in round 6;
    // the other instantiations' prologue is byte for byte what it was)
#define PARO_GEMV_PROLOGUE_ASM \
        "s_load_dwordx16 %[k0], %[kp], 0x0\n\t"                                                                                   \
        "s_load_dwordx16 s[84:99], %[kp], 0x40\n\t"                                                                               \
        "s_load_dwordx2 %[cnt], %[kp], %[cntoff]\n\t"                                                                             \
        "s_mov_b32 %[m0s], m0\n\t"                                                                                                \
        "s_mov_b32 %[p], 0\n\t"                                                                                                   \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                \
        "s_cmp_ge_i32 %[cb], s84\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s85\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s86\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s87\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s88\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s89\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_cmp_ge_i32 %[cb], s90\n\ts_addc_u32 %[p], %[p], 0\n\t"                                                                \
        "s_mov_b32 m0, %[p]\n\t"                                                                                                  \
        "s_nop 0\n\t" /* hazard: a scalar write of m0 needs one wait state before s_movrel (no compiler in here to insert it) */   \
        "s_movrels_b32 %[cb0], s83\n\t" /* cbs[p - 1] = first column block of partition p (p = 0: junk, fixed below) */           \
        "s_movrels_b32 %[e0], s91\n\t"  /* ent[p] */                                                                              \
        "s_movrels_b32 %[e1], s92\n\t"  /* ent[p + 1] */                                                                          \
        "s_cmp_eq_u32 %[p], 0\n\t"                                                                                                \
        "s_cselect_b32 %[cb0], 0, %[cb0]\n\t"                                                                                     \
        "s_mov_b32 m0, %[m0s]\n\t"                                                                                                \
        "s_nop 0"
#define PARO_GEMV_PROLOGUE_OUT [k0] "=&s"(k0), [p] "=&s"(p), [cb0] "=&s"(p_cb0_u), [e0] "=&s"(ent0), [e1] "=&s"(ent1), [m0s] "=&s"(m0_save), [cnt] "=&s"(cnt_ptr)
#define PARO_GEMV_PROLOGUE_CLOBBER "memory", "scc", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
    if constexpr (PARTS_IN || ATTN_IN) {
      asm volatile("s_load_dwordx4 %[pin], %[kp], %[pinoff]\n\t" PARO_GEMV_PROLOGUE_ASM
                   : PARO_GEMV_PROLOGUE_OUT, [pin] "=&s"(pin_ptrs)
                   : [kp] "s"(kp), [cb] "s"(cb), [cntoff] "i"(offsetof(GemvArgs, counters)), [pinoff] "i"(offsetof(GemvArgs, parts_in))
                   : PARO_GEMV_PROLOGUE_CLOBBER);
    } else {
      asm volatile(PARO_GEMV_PROLOGUE_ASM
                   : PARO_GEMV_PROLOGUE_OUT
                   : [kp] "s"(kp), [cb] "s"(cb), [cntoff] "i"(offsetof(GemvArgs, counters))
                   : PARO_GEMV_PROLOGUE_CLOBBER);
    }
#undef PARO_GEMV_PROLOGUE_ASM
#undef PARO_GEMV_PROLOGUE_OUT
#undef PARO_GEMV_PROLOGUE_CLOBBER
  }
  // pointers rebuilt from argument dwords carry no address space: name it (global), or every access through them
  // becomes a flat load that also counts against lgkmcnt
  struct Hot {
    GP<u32x4> wq;
    GP<unsigned> sz;
    GP<unsigned> rot;
    GP<unsigned short> cs;
    GP<unsigned short> x;
    GP<unsigned short> residual;
    int K, N, G, rows, krot, ksplit, gps, tsz, tstride, gstride, skew, prio, prologue, experts, parts_out, xfrag;
    long long xstride;
  } h;
  {
    auto ptr = [](unsigned lo, unsigned hi) { return ((unsigned long long)hi << 32) | (unsigned long long)lo; };
    h.wq = (GP<u32x4>)ptr(k0[0], k0[1]);
    h.sz = (GP<unsigned>)ptr(k0[2], k0[3]);
    h.rot = (GP<unsigned>)ptr(k0[4], k0[5]);
    h.cs = (GP<unsigned short>)ptr(k0[6], k0[7]);
    h.x = (GP<unsigned short>)ptr(k0[8], k0[9]);
    h.G = (int)(k0[10] & 0xffffu);
    h.K = h.G * 128;
    const int T = (int)(k0[10] >> 16);
    h.N = T * 16;
    const unsigned meta = k0[11];
    h.rows = (int)(meta & 0x7fu);
    h.parts_out = (int)((meta >> 7) & 1u);
    h.krot = (int)((meta >> 8) & 0xffu);
    h.ksplit = (int)((meta >> 16) & 0xffu);
    h.skew = (int)((meta >> 24) & 1u);
    h.prio = (int)((meta >> 25) & 1u);
    h.prologue = (int)((meta >> 26) & 3u);
    h.experts = (int)((meta >> 28) & 1u);
    h.xfrag = (int)((meta >> 30) & 1u);   // pre-rotated x in MFMA-fragment order (the schedule pre-pass of mode 1, rotate.hip)
    h.gps = (int)(k0[12] & 0xffffu);
    h.tsz = (int)(k0[12] >> 16);
    const bool order = (meta >> 29) & 1u;
    h.tstride = order ? 1 : h.G;
    h.gstride = order ? T : 1;
    h.residual = (GP<unsigned short>)ptr(k0[13], k0[14]);
    h.xstride = (long long)k0[15];
  }
  // ---- ATAIL: launch tag (as for SHR below), and the attention row
  unsigned atag = 0;
  if constexpr (ATAIL) {
    const unsigned long long did = paro_dispatch_id();
    const unsigned long long qp = (unsigned long long)__builtin_amdgcn_queue_ptr();
    atag = ((unsigned)did + (unsigned)(qp >> 6) * 0x9E3779B1u) | 0x80000000u;
    if (ks >= h.ksplit) {                             // the attention rows: workgroup (ks - ksplit) * column blocks + cb
      const int aw = (ks - h.ksplit) * a.attn_cbs + cb;
      if (wave >= 4 || aw >= a.attn_wgs) return;      // (the attention body is a 256-thread workgroup)
      const int as = aw / a.attn.Hkv, ah = aw - as * a.attn.Hkv;
      if (a.attn.Hq > 2 * a.attn.Hkv) attn_decode_body<AT, 128, 4, 0, true, true, true>(a.attn, ah, as, atag);
      else attn_decode_body<AT, 128, 2, 0, true, true, true>(a.attn, ah, as, atag);
      return;
    }
  }
  // ---- SHR: launch tag, granule buffer, and the PRODUCER workgroups (they never reach the GEMV below)
  constexpr int kAuxSc1 = 16;                    // gfx940+ cache-policy immediate of the buffer intrinsics: bit 4 = sc1 (bit 0 sc0, bit 1 nt)
  constexpr int PRR = MB < kShrTaskRows ? MB : kShrTaskRows;           // rows per producer task
  const int shr_nrp = SHR ? (h.rows + 1) / 2 : 1;    // row pairs of the batch: granules are stored per (partition, group, row pair, lane)
  unsigned xtag = 0;
  __amdgpu_buffer_rsrc_t xg_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(SHR ? a.xg : nullptr), 0, SHR ? a.shr_bytes : 0, 0x00020000);
  if constexpr (SHR) {
    // launch tag: the dispatch id (AQL packet index of this launch on its queue) + the queue's address, never zero
    const unsigned long long did = paro_dispatch_id();
    const unsigned long long qp = (unsigned long long)__builtin_amdgcn_queue_ptr();
    xtag = ((unsigned)did + (unsigned)(qp >> 6) * 0x9E3779B1u) | 0x80000000u;
    if (cb < 0) {
      // producer workgroup: wave w rotates task u = blockIdx.x * WAVES + w = (partition, group, quad of rows) in registers -- the same
      // seed / stage / finish arithmetic as the replicated form below, one rounding -- and publishes it: lane l stores channels 2l, 2l + 1
      // of two rows as one 16-byte write-through store {x, tag, x, tag}
      if (ks != 0) return;
      const int nq = (h.rows + PRR - 1) / PRR;
      unsigned short* xq = (unsigned short*)(lds + wave * (((PRR + 1) * kXhStride * 2 + 15) / 16 * 16));
      const float fscale = __builtin_ldexpf(1.0f, 49 - 14 * h.krot);
      // tasks: (partition, group that is not a wave's own first group, chunk of PRR rows).  A K-slice of n groups has n - SELF * WAVES of them
      const int l_full = max(0, h.gps - SELF * WAVES), l_last = max(0, (h.G - (h.ksplit - 1) * h.gps) - SELF * WAVES);
      const int l_sum = (h.ksplit - 1) * l_full + l_last;
      for (int u = (int)blockIdx.x * WAVES + wave; u < a.shr_units; u += a.shr_prod_wgs * WAVES) {
        const int v = u / nq, r0 = (u - v * nq) * PRR;
        const int pu = v / l_sum, w_ = v - pu * l_sum;
        const int sl = l_full > 0 ? min(w_ / l_full, h.ksplit - 1) : h.ksplit - 1;
        const int g = sl * h.gps + SELF * WAVES + (w_ - sl * l_full);
        const int pg = pu * h.G + g;
        GP<u32x4> rp = (GP<u32x4>)h.rot + (unsigned)(pg * 192 + lane);
        u32x4 rc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) rc[q] = rp[q * 64];
        const unsigned csv = *(GP<unsigned>)(h.cs + (unsigned)(pu * h.K + g * 128 + 2 * lane));
        float sa[PRR], sb[PRR];
        {
          unsigned xv[PRR];
#pragma unroll
          for (int r = 0; r < PRR; ++r) {
            const int rr = min(r0 + r, h.rows - 1);
            xv[r] = *(GP<unsigned>)(h.x + (unsigned)(rr * h.K + g * 128 + 2 * lane));
          }
          const float c0 = f16_bits_to_f32(csv & 0xffffu) * 0x1p-63f, c1 = f16_bits_to_f32(csv >> 16) * 0x1p-63f;
#pragma unroll
          for (int r = 0; r < PRR; ++r) {
            const unsigned v = (r0 + r < h.rows) ? xv[r] : 0u;
            sa[r] = A::to_f32(v & 0xffffu) * c0;
            sb[r] = A::to_f32(v >> 16) * c1;
          }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < h.krot) {
            const unsigned w = rc[t >> 2][t & 3];
            const unsigned sw = rc[2][t >> 2];
            const float P = (float)(int)(short)(w & 0xffffu), Q = (float)((int)w >> 16);
            const int src = (int)((sw >> (8 * (t & 3))) & 0xffu);
#pragma unroll
            for (int r = 0; r < PRR; ++r) {
              const float keep = __builtin_fmaf(P, sa[r], Q * sb[r]);
              const float give = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
              sb[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, give)));
              sa[r] = keep;
            }
          }
        }
        {
          const unsigned w0 = rc[2][2], w1 = rc[2][3];
          const float P = (float)(int)(short)(w0 & 0xffffu) * fscale, Q = (float)((int)w0 >> 16) * fscale;
          const unsigned oa = w1 & 0xfeu, ob = (w1 >> 8) & 0xfeu;
          const unsigned flip = w1 & 0x80000000u;
#pragma unroll
          for (int r = 0; r < PRR; ++r) {
            const float o1 = __builtin_fmaf(P, sa[r], Q * sb[r]);
            const float d = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
            const float o2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d) ^ flip);
            *(unsigned short*)((unsigned char*)(xq + r * kXhStride) + oa) = A::from_f32(o1);
            *(unsigned short*)((unsigned char*)(xq + r * kXhStride) + ob) = A::from_f32(o2);
          }
        }
        __builtin_amdgcn_wave_barrier();
        // rows r0 + 2 q, r0 + 2 q + 1 -> row pair (r0 >> 1) + q (r0 is a multiple of 4, or 0)
#pragma unroll
        for (int q = 0; q < (PRR + 1) / 2; ++q) {
          if (r0 + 2 * q < h.rows) {
            const unsigned v0 = *(const unsigned*)(xq + (2 * q) * kXhStride + 2 * lane);
            const unsigned v1 = (2 * q + 1 < PRR) ? *(const unsigned*)(xq + (2 * q + 1 < PRR ? 2 * q + 1 : 0) * kXhStride + 2 * lane) : 0u;
            const u32x4 gv = {v0, xtag, v1, xtag};
            __builtin_amdgcn_raw_buffer_store_b128(gv, xg_rsrc, (unsigned)(((pg * shr_nrp + (r0 >> 1) + q) * 64 + lane) * 16), 0, kAuxSc1);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
  // PARTS_IN: where the completed x goes -- a buffer descriptor whose size is K halves in column block 0 (and when the caller
  // wants it) and ZERO elsewhere: out-of-range buffer stores are dropped by the hardware, no branch around the store
  __amdgpu_buffer_rsrc_t xout_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pin_ptrs[1], 0, (PARTS_IN && cb == 0 && pin_ptrs[1] != 0) ? h.K * 2 : 0, 0x00020000);
  GP<u32x4> wq_p = h.wq;
  GP<unsigned> sz_p = h.sz;
  GP<unsigned short> x_p = h.x;
  int slot = 0;
  bool bad_expert = false;
  if constexpr (FMODE != 0) {
    if (h.experts) {
      slot = blockIdx.z;
      // the id is the same for the whole workgroup: pulled back into a scalar register so that the pointers derived
      // from it stay scalar (otherwise EVERY fused instantiation addresses its loads through 64-bit vector registers)
      const int ex_raw = __builtin_amdgcn_readfirstlane(a.expert_idx[slot]);
      bad_expert = (unsigned)ex_raw >= (unsigned)a.n_experts;
      const long long ex = bad_expert ? 0 : ex_raw;
      wq_p = (GP<u32x4>)((GP<unsigned char>)h.wq + ex * a.wq_estride);
      sz_p = (GP<unsigned>)((GP<unsigned char>)h.sz + ex * a.sz_estride);
      x_p = h.x + (long long)(slot / a.x_div) * a.x_sstride;
    }
  }
  // DIAG 3: phase stamps (s_memtime, shader clock) into a.slabs
  unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rt0 = 0;   // DIAG 3: chip-wide real-time counter (100 MHz) at workgroup entry
  if constexpr (DIAG == 3) { ts[0] = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  // a stamp that cannot be scheduled before `dep` exists (s_memtime alone has no data dependencies and
  // floats above the s_waitcnt it is meant to follow)
  auto stamp_after = [](float dep) -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
  };

  // The FIRST thing a wave requests is its first unit's coefficients (exchange schedule, channel scales, x): they need
  // only the partition index (from the lookup above) and the wave's first group, so the requests are issued before the
  // tile bookkeeping below (all waves of a CU share ONE scalar unit: every scalar instruction in front of the first
  // request delays the later waves' requests several times over).
  const int g_begin = ks * h.gps;
  const int g_end = min(h.G, g_begin + h.gps);
  const int n_local = g_end - g_begin;
  const int gf_first = wave < n_local ? g_begin + wave : h.G - 1;   // == unit_group(0), or the clamped dummy unit

  unsigned short* xh = (unsigned short*)(lds + wave * XH_BYTES);
  if constexpr (!PREROT) {
    // zero row (136 halves): one predicated 8-byte store instead of three 2-byte ones
    static_assert(kXhStride % 4 == 0 && (XH_BYTES % 16) == 0, "zero-row store alignment");
    if (lane < kXhStride / 4) *(u32x2*)(xh + MB * kXhStride + 4 * lane) = (u32x2){0u, 0u};
  }

  const int n = lane & 15, mq = lane >> 4;
  // A-fragment source row of this lane: MFMA row m' = lane & 15 carries batch row (m'>>2)*MR + (m'&3)
  const int mrow = lane & 15;
  const int brow = (mrow >> 2) * MR + (mrow & 3);
  const bool avalid = ((mrow & 3) < MR) && (brow < h.rows);

  float acc[TPW][MRT];   // [tile][row tile * MR + r]
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MRT; ++r) acc[j][r] = 0.f;

  struct PBuf {
    unsigned xv[PREROT ? 1 : MB];
    unsigned xu[FMODE == 2 ? MB : 1];   // SiLU*mul prologue: the `up` pair of the same two channels
    f32x4 xp[(PARTS_IN || ATTN_IN) ? 2 : 1];   // PARTS_IN: the producer's partial sums of the same two channels: [channel][slot], slots in summation order; ATTN_IN: the slots' outputs
    f32x4 am[ATTN_IN ? 2 : 1];          // ATTN_IN: the head's slot maxima, slot sums
    int g;                              // PARTS_IN: the group (where the completed x is stored)
    unsigned csv;
    u32x4 rc[3];                // exchange schedule of the group (paro_pack_rotation); unused when PREROT
    u32x4 xa[PREROT ? 4 * RT : 1];   // [row tile][k-step]
    u32x4 gq[SHR ? (MB + 1) / 2 : 1];      // SHR: this lane's granules of every PAIR of rows: {channels 2l | 2l + 1, tag} x 2
    unsigned goff;                         // SHR: where they came from (byte offset in the granule buffer: the re-poll)
  };
  struct TBuf {
    u32x4 q[TPW];
    unsigned szw[QS][TPW];
  };
  static_assert(QS == 1 || QS == 2, "one or two quantisation groups per 128-channel span");
  constexpr int SPH = 4 / QS;   // MFMA k-steps (32 channels each) per quantisation group

  // element / chunk offsets are 32-bit (host-checked ranges): 64-bit scalar index arithmetic is several instructions
  // per term, in a prologue every wave of the CU executes on the one shared scalar unit
  // pre-rotated x: plain rows [p][row][K] (mode 2: the caller's layout), or -- behind the schedule pre-pass of mode 1 (rotate.hip) --
  // MFMA-fragment order [p][g][row tile][k-step][16 mq + row % 16] x 16 bytes: one contiguous 1-KiB wave load per (row tile, k-step)
  // instead of sixteen 64-byte row pieces.  Element offset = lane part + uniform part (group, row tile, k-step strides).
  const unsigned xl_base = !PREROT ? 0u : (h.xfrag ? (unsigned)(p * h.G * RT * 2048 + (mq * 16 + brow) * 8) : (unsigned)(p * h.rows * h.K + brow * h.K + 8 * mq));
  const unsigned xg_stride = h.xfrag ? (unsigned)(RT * 2048) : 128u, xrt_stride = h.xfrag ? 2048u : (unsigned)(16 * h.K), xi_stride = h.xfrag ? 512u : 32u;

  // the rotation inputs of group g of partition pu: exchange schedule, channel scales, x (SHR: the producer's unit -- any partition)
  auto load_rot = [&](PBuf& b, int pu, int g) {
    if constexpr (PREROT) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = rt * 16 + brow;
          b.xa[rt * 4 + i] = (u32x4){0u, 0u, 0u, 0u};
          if (((mrow & 3) < MR) && row < h.rows)
            b.xa[rt * 4 + i] = *(GP<u32x4>)(x_p + ((unsigned)g * xg_stride + (unsigned)rt * xrt_stride + (unsigned)i * xi_stride) + xl_base);
        }
    } else {
      if constexpr (PARTS_IN) {
        // [K][4] fp32: the four slots of a channel are one 16-byte load, already in the reducer's summation order (slot 0 = the LAST
        // split, slot q = split q - 1), unused slots zero -- the producer's epilogue below arranges both
        b.g = g;
        GP<f32x4> pp = (GP<f32x4>)pin_ptrs[0] + (unsigned)(g * 128 + 2 * lane);
        b.xp[0] = pp[0];
        b.xp[1] = pp[1];
      }
      if constexpr (ATTN_IN) {
        const unsigned e = (unsigned)(g * 128 + 2 * lane);
        GP<f32x4> pp = (GP<f32x4>)pin_ptrs[0] + e;
        b.xp[0] = pp[0];
        b.xp[1] = pp[1];
        GP<f32x4> ml = (GP<f32x4>)(pin_ptrs[1] & ~15ull) + 2u * (e >> (unsigned)(pin_ptrs[1] & 15ull));
        b.am[0] = ml[0];
        b.am[1] = ml[1];
      }
      // 3 KiB per group, three coalesced 1-KiB wave loads: [3][lane] x 16 bytes
      GP<u32x4> rp = (GP<u32x4>)h.rot + (unsigned)((pu * h.G + g) * 192 + lane);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        // DIAG 7 / 8 (timing only, wrong results): what a 2 KiB / 1 KiB schedule would cost to FETCH -- the third / second and third
        // 1-KiB wave load of the group is not issued (profiles/r06_gemv_ablation.jsonl)
        if constexpr (DIAG == 7) { b.rc[q] = q < 2 ? rp[q * 64] : b.rc[1]; }
        else if constexpr (DIAG == 8) { b.rc[q] = q < 1 ? rp[q * 64] : b.rc[0]; }
        else b.rc[q] = rp[q * 64];
      }
      b.csv = *(GP<unsigned>)(h.cs + (unsigned)(pu * h.K + g * 128 + 2 * lane));
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int rr = r < h.rows ? r : 0;  // clamp instead of branching: keeps the load count static
        if constexpr (FMODE) {
          GP<unsigned short> xr = x_p + ((unsigned)rr * (unsigned)h.xstride + (unsigned)(g * 128 + 2 * lane));
          b.xv[r] = *(GP<unsigned>)xr;
          if constexpr (FMODE == 2) b.xu[r] = *(GP<unsigned>)(xr + h.K);
        } else if constexpr (ATTN_IN) {
          b.xv[r] = 0u;
        } else {
          b.xv[r] = *(GP<unsigned>)(x_p + (unsigned)(rr * h.K + g * 128 + 2 * lane));
        }
      }
    }
  };
  // what a wave needs of group g before it can multiply: the rotation inputs -- or, with the shared rotation, the group's granules
  // (write-through loads: they bypass this CU's L1, which another CU's stores never refresh)
  auto load_p = [&](PBuf& b, int g) {
    if constexpr (SHR) {
      b.goff = (unsigned)((((p * h.G + g) * shr_nrp) * 64 + lane) * 16);
#pragma unroll
      for (int q = 0; q < (MB + 1) / 2; ++q) {
        const int qq = q < shr_nrp ? q : 0;     // clamp instead of branching (static load count)
        b.gq[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xg_rsrc, b.goff + (unsigned)(qq * 1024), 0, kAuxSc1));
      }
    } else {
      load_rot(b, p, g);
    }
  };
  // ---- first unit's coefficient requests, at priority 3 (see the note at the driver loop), then the bookkeeping
  // K-split epoch word of this column block (see ks_tag below): the wave's OLDEST vector request -- unconditional (any
  // readable word when the launch is not split), so it is waited for with the first coefficients, costs no round trip of its
  // own and keeps the scalar-memory counter out of the rotation's cross-lane waits (a scalar load here sat in front of
  // every ds_bpermute wait of the first unit: o_proj 4.82 -> 5.02 us)
  const bool ks_handoff = h.ksplit > 1 && !h.parts_out;   // this launch reduces its K-splits itself (granules + epochs)
  const unsigned ep_raw = *(GP<unsigned>)((ks_handoff ? (GP<unsigned>)cnt_ptr : (GP<unsigned>)h.cs) + (ks_handoff ? (unsigned)cb : 0u));
  PBuf pc_first;
  if (h.prio) __builtin_amdgcn_s_setprio(3);
  if constexpr (!SHR) load_p(pc_first, gf_first);
  else if constexpr (SELF1) load_rot(pc_first, p, gf_first);
  __builtin_amdgcn_sched_barrier(0);
  if (h.prio) __builtin_amdgcn_s_setprio(0);

  // ---- tile bookkeeping of this column block (after the coefficient requests are out)
  const int p_cb0 = (int)p_cb0_u, p_t0 = (int)(ent0 & 0xffffu), p_t1 = (int)(ent1 & 0xffffu), p_sz0 = (int)(ent0 >> 16);
  const int ltile0 = (cb - p_cb0) * TPW;
  const int tile0 = p_t0 + ltile0;
  const int nt = min(TPW, p_t1 - tile0);
  const int ts0 = p_sz0 + ltile0;
  const unsigned szrow = (unsigned)(h.tsz >> 2) * 64u;  // words per group row of the scale/zero array
  // all-reduce epilogue (its own instantiations): the epoch of this thread's first output tile, whether the buffer has
  // given up before, and -- lane p -- rank p's buffer address, requested FIRST: after a kernel boundary they are misses
  // like x and the coefficients, and the whole kernel hides them.  Everything the epilogue needs is read through a
  // LAUNDERED kernarg pointer, here into vector registers and again at the epilogue: the kernel sits at the scalar-
  // register limit, and argument fields the compiler fetches at entry and keeps live spilled ~400 v_readlane into
  // the main loop (+2.5 us per launch).  Epochs are kept PER 16-COLUMN TILE in ordinary L2-cached memory: whatever
  // launch shape owns a tile, its tag sequence grows by one per launch, on every rank alike; no arrival count, no atomic
  // (256 workgroups taking a ticket on one word cost more than the launch this epilogue saves).
  typedef const __attribute__((address_space(4))) unsigned char* KArgP;
  typedef __attribute__((address_space(1))) unsigned* GWP;
  unsigned ar_first = 0, ar_gaveup = 0;
  unsigned long long ar_ptr = 0;
  if constexpr (AREP) {
    KArgP ka = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    GP<unsigned> st = (GP<unsigned>)*(const __attribute__((address_space(4))) unsigned long long*)(ka + offsetof(GemvArgs, ar_state));
    ar_first = st[kArStateTiles + tile0 + min(tid / (MRT * 64), max(nt - 1, 0))];
    ar_gaveup = st[0];
    ar_ptr = *(GP<unsigned long long>)((unsigned long long)ka + offsetof(GemvArgs, ar_peer) + 8u * (unsigned)(lane & (kArMaxWorld - 1)));
  }

  // Every load is unconditional (ragged column blocks re-read their last tile and mask it later; waves
  // with fewer units re-read their last unit) so that the compiler's vmcnt bookkeeping is exact: a wait
  // for coefficients never waits for the younger tile loads (vmcnt retires in order).
  auto load_t = [&](TBuf& b, int g) {
    auto load_q = [&]() {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int jj = j < nt ? j : nt - 1;
        b.q[j] = __builtin_nontemporal_load(wq_p + ((unsigned)((tile0 + jj) * h.tstride + g * h.gstride) * 64u + (unsigned)lane));
      }
    };
    // (PARTS_IN builds request the scale / zero words BEFORE the tiles: the register allocator copies the first unit's last-loaded
    // words in front of the unit loop there, and a copy of the youngest request is a wait for everything -- the HBM tiles included)
    if constexpr (!PARTS_IN) load_q();
#pragma unroll
    for (int hq = 0; hq < QS; ++hq) {
      const unsigned grow = (unsigned)(g * QS + hq) * szrow;   // row of the scale/zero array = quantisation group
      if constexpr (SZ_VEC) {
        GP<unsigned> sp = sz_p + (grow + (unsigned)(((ts0 >> 2) * 16 + n) * 4 + (ts0 & 3)));
#pragma unroll
        for (int v = 0; v < NSZ; ++v) {
          const SZV q = *(GP<SZV>)(sp + v * 64);
#pragma unroll
          for (int e = 0; e < SZW; ++e) b.szw[hq][v * 4 + e] = q[e];
        }
      } else {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int ts = ts0 + (j < nt ? j : nt - 1);
          b.szw[hq][j] = sz_p[grow + (unsigned)(((ts >> 2) * 16 + n) * 4 + (ts & 3))];
        }
      }
    }
    if constexpr (PARTS_IN) load_q();
  };

  // residual epilogue: the value this thread adds to its FIRST output is requested here, at kernel entry -- not as a
  // dependent load after the reduction (~1 us of pure latency at the tail) -- and UNCONDITIONALLY (clamped address,
  // validity applied at the use): a load inside a branch makes the compiler's vmcnt bookkeeping give up, and the wait
  // for the first coefficients then becomes a wait for the HBM tiles as well (measured: +0.6 us per fused launch).
  unsigned short res_raw = 0;
  bool res_valid = false;
  if constexpr (FMODE) {
    const int el = tid & 63, q = (tid >> 6) % MRT, j = tid / (MRT * 64);
    const int b = (q / MR) * 16 + (el >> 4) * MR + (q % MR);
    const bool has_res = h.residual != nullptr;
    res_valid = has_res && tid < TPW * MRT * 64 && ks == h.ksplit - 1 && j < nt && b < h.rows;   // only the workgroup that writes y
    GP<unsigned short> rbase = has_res ? h.residual : h.cs;                                     // any readable address when there is none
    res_raw = rbase[res_valid ? (unsigned)(b * h.N + (tile0 + j) * 16 + (el & 15)) : 0u];
  }
  // the first unit's tiles are requested HERE, straight after the bookkeeping they need (the unit/skew logic of the
  // driver loop below is not needed for them: unit 0 of a wave is always group gf_first)
  TBuf tc_first;
  load_t(tc_first, gf_first);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DIAG == 3) ts[1] = __builtin_amdgcn_s_memtime();
  // K-split tag of this column block: (block, per-block epoch + 1).  The epoch word lives in the workspace's counter area
  // (zero-filled once), is requested at kernel entry (ep_raw) and advanced by the block's
  // reducer at its very end: every launch that splits a block uses a tag no granule in memory can carry yet, so nothing is
  // re-armed and a late or stale granule -- or whatever another kernel left in the granule area -- is never consumed.
  // (Shared with the chain family, chain_impl.hpp: same words, same tag format.)
  unsigned ks_tag = 0;
  if (ks_handoff) {
    unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)ep_raw) + 1u;
    if ((e & 0xfffffu) == 0u) e += 1u;   // tag 0 is "never written"
    ks_tag = (e << 12) | ((unsigned)cb & 0xfffu);
  }

  // ---- rotation pieces (state in REGISTERS: lane l holds both members (A, B) of one pair of the stage)
  // The packed coefficients are 16-bit integers in units of 2^-14.  They are used AS integers (two
  // converts, no scaling op: the rotation is replicated in every workgroup and VALU-issue bound), so the
  // state grows by 2^14 per stage; exact power-of-two bookkeeping keeps it in range: the state starts at
  // 2^-63 x, and the final stage's coefficients carry 2^(49 - 14 krot).
  // start: channels 2l, 2l+1 of the group (one coalesced load), times their channel scales
  // residual epilogue: the value this thread will add to its FIRST output is requested here, at kernel entry, not
  // as a dependent load after the reduction (a global access at the tail is ~1 us of pure latency)
  // (the residual value of this thread's first output was requested above, unconditionally: res_raw / res_valid)
  float ssq[FMODE ? MB : 1];   // RMSNorm prologue: this lane's share of sum(x^2), per row
#pragma unroll
  for (int r = 0; r < (FMODE ? MB : 1); ++r) ssq[r] = 0.f;
  auto seed = [&](const PBuf& b, float (&sa)[MB], float (&sb)[MB]) {
    const float c0 = f16_bits_to_f32(b.csv & 0xffffu) * 0x1p-63f, c1 = f16_bits_to_f32(b.csv >> 16) * 0x1p-63f;
#pragma unroll
    for (int r = 0; r < MB; ++r) {
      const unsigned xv = r < h.rows ? b.xv[r] : 0u;
      float x0 = A::to_f32(xv & 0xffffu), x1 = A::to_f32(xv >> 16);
      if constexpr (ATTN_IN) {
        x0 = A::to_f32(A::from_f32(attn_merge(b.xp[0], b.am[0], b.am[1])));
        x1 = A::to_f32(A::from_f32(attn_merge(b.xp[1], b.am[0], b.am[1])));
      }
      if constexpr (PARTS_IN) {
        // no branch in here (a branch makes the compiler's vmcnt bookkeeping give up: the wait for these coefficients became a wait
        // for the unit's HBM tiles, +1.0 .. 1.6 us per launch): unused slots hold zeros, and only column block 0's buffer
        // descriptor has a non-zero size, so every other workgroup's store is dropped
        const float v0 = ((b.xp[0][0] + b.xp[0][1]) + b.xp[0][2]) + b.xp[0][3];
        const float v1 = ((b.xp[1][0] + b.xp[1][1]) + b.xp[1][2]) + b.xp[1][3];
        const unsigned h0 = A::from_f32(v0 + x0), h1 = A::from_f32(v1 + x1);   // the reducer's `v += residual`, one rounding
        x0 = A::to_f32(h0);
        x1 = A::to_f32(h1);
        __builtin_amdgcn_raw_buffer_store_b32(h0 | (h1 << 16), xout_rsrc, (unsigned)(b.g * 256 + 4 * lane), 0, 0);
      }
      if constexpr (FMODE == 2) {
        {
          const unsigned uv = r < h.rows ? b.xu[r] : 0u;
          // silu(g) * u = g * u / (1 + exp(-g)); v_exp_f32 is 2^x.  The tanh form of GELU (Gemma) is the same expression on
          // another argument: 0.5 g (1 + tanh(t)) = g / (1 + exp(-2 t)),  t = sqrt(2 / pi) (g + 0.044715 g^3)
          float a0 = x0, a1 = x1;
          if (h.prologue == PARO_PROLOGUE_GELU_TANH_MUL) {
            a0 = 1.5957691216057308f * __builtin_fmaf(0.044715f * x0 * x0, x0, x0);
            a1 = 1.5957691216057308f * __builtin_fmaf(0.044715f * x1 * x1, x1, x1);
          }
          x0 = x0 * A::to_f32(uv & 0xffffu) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a0));
          x1 = x1 * A::to_f32(uv >> 16) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a1));
        }
      } else if constexpr (FMODE == 1) {
        if (h.prologue == PARO_PROLOGUE_RMSNORM) ssq[r] = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, ssq[r]));
      }
      sa[r] = x0 * c0;
      sb[r] = x1 * c1;
    }
  };
  const float final_scale = __builtin_ldexpf(1.0f, 49 - 14 * h.krot);
  // Stage t: keep' = P A + Q B, give' = P B - Q A, then ONE cross-lane fetch: the lane keeps keep' and
  // pulls the give' of lane `src` (ds_bpermute_b32: no LDS memory, no bank conflicts).  Which member is
  // kept, the pair orientation and the running signs are folded into (P, Q) at load time (repack.hip).
  // Stage 0 is the identity that moves the natural layout into the first checkpoint stage's pairs.
  auto stage = [&](const PBuf& b, int t, float (&sa)[MB], float (&sb)[MB]) {
    const unsigned w = b.rc[t >> 2][t & 3];
    const unsigned sw = b.rc[2][t >> 2];
    const float P = (float)(int)(short)(w & 0xffffu), Q = (float)((int)w >> 16);
    const int src = (int)((sw >> (8 * (t & 3))) & 0xffu);   // 4 * source lane
#pragma unroll
    for (int r = 0; r < MB; ++r) {
      const float keep = __builtin_fmaf(P, sa[r], Q * sb[r]);
      const float give = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
      if constexpr (DIAG == 5) {          // diagnostic: no cross-lane fetch at all (wrong results)
        sb[r] = give;
      } else if constexpr (DIAG == 6) {   // diagnostic: the exchange through LDS memory (write own slot, read the source's)
        float* ex = (float*)(lds + LDS_BYTES - 16) - (wave + 1) * 64;
        ex[lane] = give;
        __builtin_amdgcn_wave_barrier();
        sb[r] = *(const float*)((const char*)ex + src);
        __builtin_amdgcn_wave_barrier();
      } else {
        sb[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, give)));
      }
      sa[r] = keep;
    }
  };
  // last checkpoint stage in place: out[a] = P A + Q B, out[b] = sigma (P B - Q A); ONE rounding to the
  // activation dtype, scattered to the channels' places in the fragment rows of unit slot `xs`
  auto finish = [&](const PBuf& b, unsigned short* xs, const float (&sa)[MB], const float (&sb)[MB]) {
    const unsigned w0 = b.rc[2][2], w1 = b.rc[2][3];
    const float P = (float)(int)(short)(w0 & 0xffffu) * final_scale, Q = (float)((int)w0 >> 16) * final_scale;
    unsigned oa = w1 & 0xfeu, ob = (w1 >> 8) & 0xfeu;   // byte offsets of the two channels in a fragment row
    const unsigned flip = w1 & 0x80000000u;
    if constexpr (DIAG == 1 || DIAG == 2 || DIAG == 4) { oa = 4 * lane; ob = 4 * lane + 2; }
#pragma unroll
    for (int r = 0; r < MB; ++r) {
      float o1 = sa[r], o2 = sb[r];
      if constexpr (DIAG == 0 || DIAG == 3 || DIAG == 5 || DIAG == 6 || DIAG == 7 || DIAG == 8) {
        o1 = __builtin_fmaf(P, sa[r], Q * sb[r]);
        const float d = __builtin_fmaf(P, sb[r], -(Q * sa[r]));
        o2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d) ^ flip);
      }
      *(unsigned short*)((unsigned char*)(xs + r * kXhStride) + oa) = A::from_f32(o1);
      *(unsigned short*)((unsigned char*)(xs + r * kXhStride) + ob) = A::from_f32(o2);
    }
  };
  auto touch = [&](const PBuf& b, float (&sa)[MB]) {   // DIAG 4: the schedule words are waited for, the stages not run
    unsigned acc_w = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) acc_w ^= b.rc[q][0] ^ b.rc[q][1] ^ b.rc[q][2] ^ b.rc[q][3];
    sa[0] += (acc_w == 0x12345u) ? 1.f : 0.f;
  };

  // ---- consume one unit's tiles: fragments from LDS (or registers), sums, unpack -> MFMA -> scale / zero
  const typename A::Unpack upk = A::unpack_consts();
  auto consume = [&](const vec8 (&af)[4 * RT], const TBuf& t) {
    // sx = sum_k x_k and so = sum_k x_k off_k (off_k = the per-element offset unpack_fast leaves in), per row tile
    f32x4 sx[RT][QS], so[RT][QS];
    {
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const u32x4 offs = {A::kOffFrag0, A::kOffFrag1, A::kOffFrag0, A::kOffFrag1};
      const vec8 ob = __builtin_bit_cast(vec8, ones);
      const vec8 fb = __builtin_bit_cast(vec8, offs);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int hq = 0; hq < QS; ++hq) {
          sx[rt][hq] = (f32x4){0.f, 0.f, 0.f, 0.f};
          so[rt][hq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sx[rt][i / SPH] = A::mfma(af[rt * 4 + i], ob, sx[rt][i / SPH]);
          so[rt][i / SPH] = A::mfma(af[rt * 4 + i], fb, so[rt][i / SPH]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      f32x4 d[RT][QS];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int hq = 0; hq < QS; ++hq) d[rt][hq] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (DIAG == 2) {
        d[0][0][0] = __builtin_bit_cast(float, (t.q[j][0] ^ t.q[j][1] ^ t.q[j][2] ^ t.q[j][3]) & 0x3fffffffu);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned w4[4];
          A::unpack_fast(t.q[j][i], w4, upk);      // one unpack feeds every row tile
          const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
          const vec8 bfrag = __builtin_bit_cast(vec8, wv);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) d[rt][i / SPH] = A::mfma(af[rt * 4 + i], bfrag, d[rt][i / SPH]);
        }
      }
#pragma unroll
      for (int hq = 0; hq < QS; ++hq) {
        const unsigned szw = t.szw[hq][j];
        const float s = f16_bits_to_f32(szw & 0xffffu);
        const float zf = f16_bits_to_f32(szw >> 16);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < MR; ++r)
            acc[j][rt * MR + r] =
                __builtin_fmaf(s, __builtin_fmaf(-zf, sx[rt][hq][r], d[rt][hq][r] - so[rt][hq][r]), acc[j][rt * MR + r]);
      }
    }
  };
  auto frags_from_lds = [&](const unsigned short* xs, vec8 (&af)[4 * RT]) {
    const unsigned short* afrag = xs + (avalid ? brow : MB) * kXhStride + 8 * mq;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const vec8*)(afrag + 32 * i);
  };

  // ---- SHR: the gather of this wave's first group.  The producers need about 2 us from the launch; until then the wave PROBES: one
  // 16-byte request for the whole wave (lane 0's granule of the first row pair) per round, then the real gather -- which is checked
  // in full below, so the probe is a heuristic, never a proof.  (A gather at kernel entry finds nothing, and 1500 waves re-polling
  // whole groups is L2 traffic in front of the producers' own loads: the first cut of this mode, 8 rows 14.9 us against 8.4.)
  if constexpr (SHR && !SELF1) {
    const unsigned poff = (unsigned)((((p * h.G + gf_first) * shr_nrp) * 64) * 16);
    for (int spin = 0; spin < (1 << 15); ++spin) {
      const u32x4 pv = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xg_rsrc, poff, 0, kAuxSc1));
      if (__builtin_amdgcn_readfirstlane((int)pv[1]) == (int)xtag || a.pd == 99) break;
      __builtin_amdgcn_s_sleep(3);
    }
    load_p(pc_first, gf_first);
  }
  // SHR: every lane's granules of the group must carry THIS launch's tag before they are used; a granule that does not yet is polled
  // again (bounded; the re-poll is inline assembly with its own wait, so that the compiler's vmcnt bookkeeping of the pipelined loads
  // around it stays exact -- the hardware counter is at zero when it leaves).  Give-up: NaN channels (they reach every output of the
  // group through the matrix cores) and the workspace's sticky status word -- never a silently stale activation.
  bool shr_gaveup = false;
  auto shr_validate = [&](PBuf& b) {
    auto any_bad = [&]() {
      bool bad = false;
#pragma unroll
      for (int q = 0; q < (MB + 1) / 2; ++q)
        bad = bad || (2 * q < h.rows && b.gq[q][1] != xtag) || (2 * q + 1 < h.rows && b.gq[q][3] != xtag);
      return __builtin_amdgcn_ballot_w64(bad) != 0ull;
    };
    if (__builtin_expect(any_bad() && a.pd != 99, 0)) {
      int spin = 0;
      do {      // every row pair of the group again, ONE batch per round
        __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int q = 0; q < (MB + 1) / 2; ++q) {
          const int qq = q < shr_nrp ? q : 0;
          b.gq[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xg_rsrc, b.goff + (unsigned)(qq * 1024), 0, kAuxSc1));
        }
        ++spin;
      } while (any_bad() && spin < (1 << 15));
      if (any_bad()) {
#pragma unroll
        for (int q = 0; q < (MB + 1) / 2; ++q) b.gq[q] = (u32x4){0x7fc07fc0u, xtag, 0x7fc07fc0u, xtag};   // NaN in both activation types
        shr_gaveup = true;
      }
    }
  };

  bool has_work_any = true;
  {
    // One unit at a time, distance-1 software pipeline.
    // (Tried and measured equal or slower, MI355X, every Llama-3-8B / Qwen3-4B shape: deeper prefetch
    // (tiles two units ahead: gate_up 14.5 -> 16.1 us; coefficients two ahead; both);
    // all of a wave's 2..4 units rotated together with interleaved stage chains and every load issued up
    // front -- the rotation then is VALU-issue bound (7 issue slots per unit and stage, replicated in every
    // workgroup), not latency bound, so overlapping the chains buys nothing; a workgroup barrier between
    // the coefficient and the tile requests; the first tiles requested only once the first coefficients
    // have arrived.  See DESIGN.md, "where the time goes".)
    PBuf pc, pn;
    TBuf tc, tn;
    // PFP: request the NEXT unit's coefficients before this unit's rotation; PFT: request the next
    // unit's tiles before this unit's tiles are consumed.  Coefficients are always requested before
    // the tiles of the same unit.
    auto step = [&](auto pfp_tag, auto pft_tag, int gp, int gt, auto self_tag) {
      constexpr bool PFP = decltype(pfp_tag)::value;
      constexpr bool PFT = decltype(pft_tag)::value;
      constexpr bool ROT_SELF = decltype(self_tag)::value;     // SHR hybrid: this unit is the wave's own first group
      if constexpr (PFP && !ROT_SELF) load_p(pn, gp);
      vec8 af[4 * RT];
      if constexpr (PREROT) {
#pragma unroll
        for (int i = 0; i < 4 * RT; ++i) af[i] = __builtin_bit_cast(vec8, pc.xa[i]);
      } else if constexpr (SHR && ROT_SELF) {
        float sa[MB], sb[MB];
        seed(pc, sa, sb);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < h.krot) stage(pc, t, sa, sb);
        }
        finish(pc, xh, sa, sb);
        __builtin_amdgcn_wave_barrier();
        frags_from_lds(xh, af);
        // the next group's granules are requested only now (~1.5 .. 2 us after the launch: the producers are publishing), not at entry
        if constexpr (PFP) load_p(pn, gp);
      } else if constexpr (SHR) {
        shr_validate(pc);
#pragma unroll
        for (int r = 0; r < MB; ++r) *(unsigned*)(xh + r * kXhStride + 2 * lane) = pc.gq[r >> 1][2 * (r & 1)];
        __builtin_amdgcn_wave_barrier();
        frags_from_lds(xh, af);
      } else {
        float sa[MB], sb[MB];
        seed(pc, sa, sb);
        if constexpr (DIAG == 3) { if (ts[2] == 0) ts[2] = stamp_after(sa[0] + sb[0]); }
        if constexpr (DIAG == 0 || DIAG == 3 || DIAG == 5 || DIAG == 6 || DIAG == 7 || DIAG == 8) {
          // (the universal krot = 8 without the per-stage compare + branch was tried twice: round 2 -- the 4-wave o_proj build 0.8 us
          // SLOWER -- and round 3 in the 8- / 16-wave builds only: down_proj -0.1 .. -0.3 us at one row, but the 16-wave build for
          // 2..4 rows came out of the compiler with nondeterministic results (tools/_stress-style loop: 136 mismatching runs of 150,
          // none without it; profiles/NOTES.md).  The compare stays.)
          {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              if (t < h.krot) stage(pc, t, sa, sb);
            }
          }
        }
        if constexpr (DIAG == 4) touch(pc, sa);
        if constexpr (DIAG == 3) { if (ts[9] == 0) ts[9] = stamp_after(sa[0] + sb[0]); }
        finish(pc, xh, sa, sb);
        __builtin_amdgcn_wave_barrier();
        frags_from_lds(xh, af);
      }
      if constexpr (PFT) load_t(tn, gt);
      consume(af, tc);
      if constexpr (DIAG == 3) { if (ts[4] == 0) ts[4] = stamp_after(acc[0][0]); }
      if constexpr (PFP) pc = pn;
      if constexpr (PFT) tc = tn;
    };
    const std::true_type yes{};
    const std::false_type no{};
    // The first unit's loads are issued unconditionally (group index clamped) BEFORE any branch, so the
    // compiler fetches the whole argument block in one scalar batch at kernel entry.  A wave without
    // work (fewer groups than waves) runs one clamped unit and discards it.
    // Which units (groups) this wave takes.  Even split: unit i of wave w is group w + i * WAVES.  The
    // waves that share a SIMD do not start together, though: the per-wave timeline shows the second / third /
    // fourth wave of a SIMD receiving its first coefficients one step later each (3800 vs 9150 cycles with
    // 8 tiles per wave; 2250 / 3080 / 4080 / 5530 with 16 waves) and finishing that much later.  With
    // h.skew the first wave of every SIMD therefore takes one unit more and the last one unit less (static,
    // so results stay bit-reproducible): rounds 0 .. c-2 as before, round c-1 without the last rank, round c
    // for the first rank only.
    constexpr int RANKS = WAVES / 4;
    const int c_even = n_local / WAVES;
    const bool skew = h.skew && RANKS >= 2 && n_local == c_even * WAVES && c_even >= (RANKS == 2 ? 3 : 2);
    const int rank = wave >> 2;
    int my_count;
    if (skew)
      my_count = c_even + (rank == 0 ? 1 : 0) - (rank == RANKS - 1 ? 1 : 0);
    else
      my_count = wave < n_local ? (n_local - wave + WAVES - 1) / WAVES : 0;
    auto unit_group = [&](int i) {
      int li = wave + i * WAVES;
      if (skew && i >= c_even - 1) {
        const int base = (c_even - 1) * WAVES;
        li = (i == c_even - 1) ? base + wave : base + (WAVES - 4) + wave;   // round c-1: ranks 0 .. RANKS-2; round c: rank 0
      }
      return g_begin + li;
    };
    const bool has_work = my_count > 0;
    // (the first unit's coefficient and tile requests went out at kernel entry: `pc_first` / `tc_first`, group gf_first ==
    // unit_group(0), or the clamped dummy unit of a wave without work)
    // Issue priority: the CU returns vector-memory data in request order.  Left alone, each SIMD issues
    // [wave 0: coefficients, tiles][wave 4: coefficients, tiles] ..., so the small L2-resident coefficient loads of
    // the later waves come back behind the earlier waves' HBM tile loads.  Every wave therefore ran at priority 3
    // until its first coefficient requests were out and dropped to 0 before its tile requests (h.prio).
    pc = pc_first;
    tc = tc_first;
    if constexpr (SELF1) {
      if (my_count > 1) {
        const int g1 = unit_group(1);
        step(yes, yes, g1, g1, yes);
        for (int i = 1; i + 1 < my_count; ++i) {
          const int gn = unit_group(i + 1);
          step(yes, yes, gn, gn, no);
        }
        step(no, no, 0, 0, no);
      } else {
        step(no, no, 0, 0, yes);
      }
    } else {
      for (int i = 0; i + 1 < my_count; ++i) {
        const int gn = unit_group(i + 1);
        step(yes, yes, gn, gn, no);
      }
      step(no, no, 0, 0, no);   // the last unit: nothing left to request
    }
    if (!has_work) {
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < MRT; ++r) acc[j][r] = 0.f;
    }
    has_work_any = has_work;
  }

  if constexpr (DIAG == 3) ts[5] = stamp_after(acc[0][0]);  // all units done
  // the epilogue's scalars, fetched here (a second laundered pointer: nothing of it is live in the main loop)
  int ar_world = 0, ar_rank = 0;
  long long ar_slot = 0, ar_off = 0;
  GWP ar_st = nullptr;
  if constexpr (AREP) {
    KArgP kb = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kb));
    ar_st = (GWP)*(const __attribute__((address_space(4))) unsigned long long*)(kb + offsetof(GemvArgs, ar_state));
    ar_world = *(const __attribute__((address_space(4))) int*)(kb + offsetof(GemvArgs, ar_world));
    ar_rank = *(const __attribute__((address_space(4))) int*)(kb + offsetof(GemvArgs, ar_rank));
    ar_slot = *(const __attribute__((address_space(4))) long long*)(kb + offsetof(GemvArgs, ar_slot));
    ar_off = *(const __attribute__((address_space(4))) long long*)(kb + offsetof(GemvArgs, ar_off));
  }
  auto ar_peer_base = [&](int r) -> unsigned char* {   // rank r's buffer: lane r of ar_ptr
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ar_ptr, r);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ar_ptr >> 32), r);
    return (unsigned char*)(((unsigned long long)hi << 32) | lo);
  };
  // every rank's partial of output column `col` -> the world's sum (rank order, fp32: bit-identical on all ranks)
  // (kept small: this code runs once per launch from a cold instruction cache -- the first version, unrolled 16 ways at
  // two call sites, was 2000 instructions of mostly skipped blocks, every skip a taken branch into an unfetched line)
  auto ar_exchange = [&](float v, int col, unsigned ar_epoch) -> float {
    const long long set0 = ar_off + (long long)((int)(ar_epoch & 1u) * ar_world) * ar_slot * 8;
    const unsigned long long gran = ((unsigned long long)ar_epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    for (int p = 0; p < ar_world; ++p) {
      if (p == ar_rank) continue;
      unsigned long long* dst = (unsigned long long*)(ar_peer_base(p) + set0) + (long long)ar_rank * ar_slot + col;
      __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned char* mine = ar_peer_base(ar_rank);
    const unsigned long long* src = (const unsigned long long*)(mine + set0) + col;
    constexpr int CH = 8;   // peers polled at once: one round for a world of up to 8, two for 16
    float s = 0.f;
    for (int r0 = 0; r0 < ar_world; r0 += CH) {
      unsigned long long got[CH];
      bool all = false;
      for (int spin = 0; !all; ++spin) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (r0 + q < ar_world && r0 + q != ar_rank) got[q] = __hip_atomic_load(src + (long long)(r0 + q) * ar_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        all = true;
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (r0 + q < ar_world && r0 + q != ar_rank) all = all && (unsigned)(got[q] >> 32) == ar_epoch;
        if (all) break;
        if (spin >= (ar_gaveup != 0u ? 0 : (1 << 22))) {
          ((unsigned*)mine)[1] = PARO_WS_STATUS_GIVEUP;   // a peer never arrived: sticky, read by the host
          ar_st[0] = 1u;                                  // ... and by later launches, which poll once instead of waiting again
          ar_gaveup = 1u;
          break;
        }
        if (spin < 4096) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q)   // rank order, the own partial in its place: the same sum on every rank
        if (r0 + q < ar_world) s += r0 + q == ar_rank ? v : __builtin_bit_cast(float, (unsigned)got[q]);
    }
    return s;
  };
  // ---- reduce the workgroup's waves (different groups, same columns) through LDS
  __syncthreads();
  if constexpr (DIAG == 3) ts[7] = __builtin_amdgcn_s_memtime();   // every wave of the workgroup has finished its units
  float* red = (float*)lds;
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MRT; ++r) red[((wave * TPW + j) * MRT + r) * 64 + lane] = acc[j][r];
  float* ssl = (float*)(lds + LDS_BYTES - SS_BYTES);   // [wave][row]
  if constexpr (FMODE == 1) {
    if (h.prologue == PARO_PROLOGUE_RMSNORM) {
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        // wave-wide sum on the VALU's DPP paths (six dependent adds, ~50 cycles; the total ends up in lane 63) -- six
        // __shfl_xor steps are six dependent ds_bpermute round trips (~600 cycles at the tail of every fused launch)
        float v = has_work_any ? ssq[r] : 0.f;
        auto dpp_add = [](float a, auto ctrl_tag, auto mask_tag) {
          constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
          return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, MASK, 0xf, false));
        };
        v = dpp_add(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
        v = dpp_add(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
        v = dpp_add(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
        v = dpp_add(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror: every lane = its row's sum
        v = dpp_add(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
        v = dpp_add(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3
        if (lane == 63) ssl[wave * MB + r] = v;
      }
    }
  }
  __syncthreads();
  if constexpr (DIAG == 3) ts[8] = __builtin_amdgcn_s_memtime();   // partials staged

  unsigned short* y_p = a.y;
  if constexpr (FMODE != 0) {
    if (h.experts) y_p = a.y + (long long)slot * a.y_sstride;
  }
  const bool direct = (h.ksplit == 1);
  for (int e = tid; e < TPW * MRT * 64; e += WAVES * 64) {
    const int el = e & 63, q = (e >> 6) % MRT, j = e / (MRT * 64);
    const int b = (q / MR) * 16 + (el >> 4) * MR + (q % MR);   // row tile * 16 + row inside the tile
    if (j >= nt || b >= h.rows) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[e + w * TPW * MRT * 64];
    const int col = (tile0 + j) * 16 + (el & 15);
    if (direct) {
      if constexpr (FMODE == 1) {
        if (h.prologue == PARO_PROLOGUE_RMSNORM) {
          float ss = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) ss += ssl[w * MB + b];
          v *= __builtin_amdgcn_rsqf(ss / (float)h.K + a.eps);
        }
      }
      if constexpr (!AREP) {
        if (a.bias) v += A::to_f32(a.bias[col]);
        if constexpr (FMODE) {
          if (h.residual) v += (e == tid && res_valid) ? A::to_f32(res_raw) : A::to_f32(h.residual[(int64_t)b * h.N + col]);
          if (bad_expert) v = __builtin_nanf("");   // a slot whose expert id is out of range: loud, and its loads stayed in bounds
        }
        y_p[(int64_t)b * h.N + col] = A::from_f32(v);
      }
    } else if (h.parts_out) {
      // deferred reduction (one row, K-split launches): every split leaves its fp32 partial sum and exits -- no granule, no poll;
      // the CONSUMER launch adds the partials while it seeds its rotation (FUSED | 8 above)
      // Layout [N][4]: slot 0 = the last split, slot q = split q - 1 (the order in which the in-launch reducer adds them), the
      // last split also zeroes the slots beyond the split count -- the consumer adds four slots, no count, no mask.
      if constexpr (ATAIL) {
        // (attention tail: the consumer runs in THIS launch -- {fp32, launch tag} granules, written through; the unused slots too, so
        // that it can check four tags without knowing the split count)
        unsigned long long* pc = a.slabs + (int64_t)col * 4;
        const bool last = ks == h.ksplit - 1;
        const unsigned long long tg = (unsigned long long)atag << 32;
        __hip_atomic_store(pc + (last ? 0 : ks + 1), tg | (unsigned long long)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last) for (int q = h.ksplit; q < 4; ++q) __hip_atomic_store(pc + q, tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if constexpr (!AREP) {
        float* pc = (float*)a.slabs + (int64_t)col * 4;
        const bool last = ks == h.ksplit - 1;
        pc[last ? 0 : ks + 1] = v;
        if (last) for (int q = h.ksplit; q < 4; ++q) pc[q] = 0.f;
      }
    } else if (ks != h.ksplit - 1) {
      // producer: ONE 8-byte {tag, fp32 partial} granule per output, written through (sc1); no
      // drain, no flag, no fence -- the data IS the flag (cdna guide G16 recipe R2); then exit.
      const unsigned long long gv = ((unsigned long long)ks_tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
      __hip_atomic_store(a.slabs + ((int64_t)ks * h.rows + b) * h.N + col, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (AREP) continue;
    } else {
      // reducer (the last K-split of this column block; dispatched after the others): keep the own
      // partial in registers, poll the other splits' granules until this launch's tag appears (bounded),
      // write y once; the block's epoch word is advanced below.
      // The first poll of EVERY split is issued before any of them is looked at (up to 4 per batch): a poll is a
      // ~0.6 us round trip to the coherence point, and one after the other they were most of the hand-off's cost.
      const int nsp = h.ksplit - 1;
      for (int dly = 0; dly < a.poll_delay; ++dly) __builtin_amdgcn_s_sleep(1);
      for (int s0 = 0; s0 < nsp; s0 += 4) {
        unsigned long long gq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int s = min(s0 + q, nsp - 1);   // clamped duplicate beyond the last split (never consumed)
          gq[q] = __hip_atomic_load(a.slabs + ((int64_t)s * h.rows + b) * h.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int s = s0 + q;
          if (s >= nsp) break;
          unsigned long long* gp = a.slabs + ((int64_t)s * h.rows + b) * h.N + col;
          unsigned long long gv = gq[q];
          for (int spin = 0; (unsigned)(gv >> 32) != ks_tag && spin < (1 << 17); ++spin) {
            __builtin_amdgcn_s_sleep(2);
            gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((unsigned)(gv >> 32) != ks_tag) {
            // give-up: never a silent wrong sum -- the output becomes NaN and the sticky status word of the
            // workspace is set (paro_workspace_status / ops.check_workspace); a producer that arrives late
            // carries this launch's tag, which no later launch waits for.  Unreachable while every workgroup of the launch is resident
            // (checked on the host before a K-split launch).
            a.counters[PARO_WS_STATUS_OFFSET / 4] = PARO_WS_STATUS_GIVEUP;
            v = __builtin_nanf("");
          } else {
            v += __builtin_bit_cast(float, (unsigned)gv);
          }
        }
      }
      if constexpr (!AREP) {
        if (a.bias) v += A::to_f32(a.bias[col]);
        if constexpr (FMODE) {
          if (h.residual) v += (e == tid && res_valid) ? A::to_f32(res_raw) : A::to_f32(h.residual[(int64_t)b * h.N + col]);
        }
        y_p[(int64_t)b * h.N + col] = A::from_f32(v);
      }
    }
    if constexpr (AREP) {
      // the world's sum of this column (ONE call site), then bias and residual once, the store, and the tile's next epoch
      unsigned ep = (e == tid ? ar_first : ar_st[kArStateTiles + (col >> 4)]) + 1u;
      if (ep == 0u) ep = 2u;   // tag 0 is "never written"; 2 keeps the set parity alternating across the wrap
      v = ar_exchange(v, col, ep);
      if (a.bias) v += A::to_f32(a.bias[col]);
      if (h.residual) v += (e == tid && res_valid) ? A::to_f32(res_raw) : A::to_f32(h.residual[(int64_t)b * h.N + col]);
      y_p[(int64_t)b * h.N + col] = A::from_f32(v);
      // (after the output store: a store in front of the residual read would be waited for -- vmcnt counts stores;
      // the tile's 16 lanes read the word in one instruction, before this store)
      if ((col & 15) == 0) ar_st[kArStateTiles + (col >> 4)] = ep;
    }
  }
  if (ks_handoff && ks == h.ksplit - 1 && tid == 0) a.counters[cb] = ks_tag >> 12;
  if constexpr (SHR) {
    if (shr_gaveup) a.counters[PARO_WS_STATUS_OFFSET / 4] = PARO_WS_STATUS_GIVEUP;
  }
  if constexpr (FMODE == 1) {
    // RMSNorm prologue on a launch that leaves partial sums: no workgroup sees all of K, so the norm's scalar travels with the partial
    // sums -- row N of the buffer gets this K-slice's sum of squares (same slot order, unused slots zero) and whoever completes the
    // sums scales them by rsqrt(sum / K + eps) (the attention kernel for qkv, attn.hip)
    if (h.parts_out && h.prologue == PARO_PROLOGUE_RMSNORM && cb == 0 && tid == 0) {
      float ss = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) ss += ssl[w * MB];
      const bool last = ks == h.ksplit - 1;
      if constexpr (ATAIL) {
        unsigned long long* pr = a.slabs + (int64_t)h.N * 4;
        const unsigned long long tg = (unsigned long long)atag << 32;
        __hip_atomic_store(pr + (last ? 0 : ks + 1), tg | (unsigned long long)__builtin_bit_cast(unsigned, ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last) for (int q = h.ksplit; q < 4; ++q) __hip_atomic_store(pr + q, tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        float* pr = (float*)a.slabs + (int64_t)h.N * 4;
        pr[last ? 0 : ks + 1] = ss;
        if (last) for (int q = h.ksplit; q < 4; ++q) pr[q] = 0.f;
      }
    }
  }
  if constexpr (DIAG == 3) {
    ts[6] = __builtin_amdgcn_s_memtime();
    // 80 words per workgroup: wave 0's phase stamps [0..8], HW_ID | XCC_ID << 32 [9], stages done [10], then
    // (start, first coefficients arrived, first unit done, all units done) of every wave from [16]
    if (a.slabs) {
      unsigned long long* dbg = a.slabs + ((int64_t)(ks * gridDim.x + cb)) * 80;
      if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dbg[k] = ts[k];
        dbg[10] = ts[9];
        dbg[11] = rt0;                                  // chip-wide clock at entry / exit: dispatch ramp and tail
        dbg[12] = __builtin_amdgcn_s_memrealtime();
        dbg[9] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
                 ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
      }
      if (lane == 0 && wave < 16) {
        dbg[16 + 4 * wave] = ts[0];
        dbg[17 + 4 * wave] = ts[2];
        dbg[18 + 4 * wave] = ts[4];
        dbg[19 + 4 * wave] = ts[5];
      }
    }
  }
}

// ---- per-translation-unit launch tables (one TU per activation type x PREROT, built in parallel)
constexpr bool tpw_is_pow2(int t) { return t == 1 || t == 2 || t == 4 || t == 8; }

// The in-launch K-split has reducers spin on granules that other workgroups of the SAME launch publish:
// forward progress is guaranteed -- whatever order the hardware dispatches workgroups in -- exactly when
// every workgroup of the grid is resident at once.  The occupancy query runs once per instantiation.
constexpr int PARO_ERR_NOT_RESIDENT = -100;   // internal: mapped to PARO_ERR_UNSUPPORTED at the ABI boundary
int device_cu_count();
template <auto Kern, int THREADS>
int launch_checked(const GemvArgs& a, dim3 grid, hipStream_t st) {
  if ((a.ksplit > 1 && !a.parts_out) || a.shared_rot) {   // (partial sums left to the consumer: nobody waits inside the launch; shared rotation: every wave waits for the producers)
    static int per_cu = -1;
    if (per_cu < 0) {
      int v = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, Kern, THREADS, 0) != hipSuccess || v < 1) v = 1;
      per_cu = v;
    }
    const long long cap = (long long)per_cu * device_cu_count();
    // (shared rotation: the producer workgroups never wait and the workgroups in front of the other grid rows exit at once -- what must
    // fit the chip together is the producers plus every column block x K-slice)
    const long long waiting = a.shared_rot ? (long long)(grid.x - a.shr_prod_wgs) * grid.y + a.shr_prod_wgs : (long long)grid.x * grid.y;
    if (waiting > cap)
      return fail(PARO_ERR_NOT_RESIDENT, "%s grid of %u x %u workgroups exceeds the %lld that are resident at once; "
                  "use a smaller ksplit or more tiles per wave", a.shared_rot ? "shared-rotation" : "K-split", grid.x, grid.y, cap);
  }
  hipLaunchKernelGGL(Kern, grid, dim3(THREADS), 0, st, a);
  return PARO_OK;
}

template <typename AT, int TPW, int MB, bool PREROT, int FUSED>
int launch_waves_fused_mode(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if constexpr (!PREROT && MB <= 4 && tpw_is_pow2(TPW)) {
    if (a.qs == 2) {   // group_size 64: 8- and 4-wave workgroups
      if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, false, 1, FUSED, 2>, 512>(a, grid, st);
      if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, false, 1, FUSED, 2>, 256>(a, grid, st);
      return fail(PARO_ERR_UNSUPPORTED, "group_size 64: fused prologue / epilogue is built for 4 or 8 waves per workgroup (got %d)", waves);
    }
    if constexpr (TPW < 8) {
      if (waves == 16) return launch_checked<gemv_kernel<AT, TPW, MB, 16, false, 1, FUSED>, 1024>(a, grid, st);
    }
    if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, false, 1, FUSED>, 512>(a, grid, st);
    if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, false, 1, FUSED>, 256>(a, grid, st);
  }
  return fail(PARO_ERR_UNSUPPORTED, "fused prologue / epilogue: not built for %d tiles per wave x %d waves x %d rows%s", TPW, waves,
              MB, PREROT ? " (pre-rotated mode)" : "");
}
template <typename AT, int TPW, int MB, bool PREROT>
int launch_waves_fused(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if (a.ar_mine) {   // + all-reduce epilogue (FUSED | 4): one row only
    if constexpr (MB == 1) {
      if (a.prologue == PARO_PROLOGUE_SILU_MUL || a.prologue == PARO_PROLOGUE_GELU_TANH_MUL) return launch_waves_fused_mode<AT, TPW, MB, PREROT, 6>(a, waves, grid, st);
      return launch_waves_fused_mode<AT, TPW, MB, PREROT, 5>(a, waves, grid, st);
    }
    return fail(PARO_ERR_UNSUPPORTED, "the all-reduce epilogue is built for one row");
  }
  if (a.attn_in) {   // x = the merge of a split attention launch's slots (FUSED | 16): one row, no prologue
    if constexpr (MB == 1 && !PREROT) return launch_waves_fused_mode<AT, TPW, MB, PREROT, 16>(a, waves, grid, st);
    return fail(PARO_ERR_UNSUPPORTED, "attention slots as input are built for one row, in-kernel rotation");
  }
  if (a.attn_wgs) {   // + attention tail (FUSED | 128): the one-row RMSNorm-prologue projection that leaves partial sums
    if constexpr (MB == 1 && !PREROT && tpw_is_pow2(TPW)) {
      if (a.qs == 1 && (waves == 4 || waves == 8) && a.parts_out && a.prologue == PARO_PROLOGUE_RMSNORM) {
        if (a.parts_in) {
          if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, 1, 8, false, 1, 137>, 512>(a, grid, st);
          return launch_checked<gemv_kernel<AT, TPW, 1, 4, false, 1, 137>, 256>(a, grid, st);
        }
        if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, 1, 8, false, 1, 129>, 512>(a, grid, st);
        return launch_checked<gemv_kernel<AT, TPW, 1, 4, false, 1, 129>, 256>(a, grid, st);
      }
    }
    return fail(PARO_ERR_UNSUPPORTED, "the attention tail is built for one row, RMSNorm prologue, partial sums out, group_size 128, 4 or 8 waves, 1 / 2 / 4 / 8 tiles per wave");
  }
  if (a.parts_in) {   // x = base + the producer's partial sums (FUSED | 8): one row, RMSNorm or no prologue
    if constexpr (MB == 1 && !PREROT) {
      if (a.prologue == PARO_PROLOGUE_NONE && !(a.hot.residual_lo | a.hot.residual_hi)) return launch_waves_fused_mode<AT, TPW, MB, PREROT, 8>(a, waves, grid, st);
      return launch_waves_fused_mode<AT, TPW, MB, PREROT, 9>(a, waves, grid, st);
    }
    return fail(PARO_ERR_UNSUPPORTED, "partial sums as input are built for one row, in-kernel rotation");
  }
  if (a.prologue == PARO_PROLOGUE_SILU_MUL || a.prologue == PARO_PROLOGUE_GELU_TANH_MUL) return launch_waves_fused_mode<AT, TPW, MB, PREROT, 2>(a, waves, grid, st);
  return launch_waves_fused_mode<AT, TPW, MB, PREROT, 1>(a, waves, grid, st);
}

template <typename AT, int TPW, int MB, bool PREROT, int PD>
int launch_waves_pd(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if (a.qs == 2) {   // group_size 64: power-of-two tiles per wave, 8- and 4-wave workgroups, shipping build only
    if constexpr (tpw_is_pow2(TPW) && PD == 1) {
      if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, PREROT, 1, 0, 2>, 512>(a, grid, st);
      if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, PREROT, 1, 0, 2>, 256>(a, grid, st);
    }
    return fail(PARO_ERR_UNSUPPORTED, "group_size 64 is built for 1 / 2 / 4 / 8 tiles per wave and 4 or 8 waves per workgroup (got %d x %d)", TPW, waves);
  }
  if constexpr (tpw_is_pow2(TPW) && TPW < 8 && MB <= 4) {   // 8 tiles x 16 waves does not fit 128 VGPRs
    if (waves == 16) return launch_checked<gemv_kernel<AT, TPW, MB, 16, PREROT, PD>, 1024>(a, grid, st);
  }
  if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, PREROT, PD>, 512>(a, grid, st);
  if constexpr (tpw_is_pow2(TPW)) {
    if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, PREROT, PD>, 256>(a, grid, st);
  }
  return fail(PARO_ERR_UNSUPPORTED, "waves per workgroup = %d not built for %d tiles per wave x %d batch rows", waves, TPW, MB);
}

// shared rotation (FUSED = 32): plain kernel, 1..16 rows, power-of-two tiles per wave, group_size 128 and 64
template <typename AT, int TPW, int MB, int F = 32>
int launch_waves_shared(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if constexpr (F == 32 && MB >= 2 && MB <= 8) {   // the hybrid (every wave rotates its own first group): 2..8 rows (one row never shares its rotation)
    if (a.shr_self == 1) return launch_waves_shared<AT, TPW, MB, 96>(a, waves, grid, st);
  }
  if (a.shr_self != (F == 96 ? 1 : 0)) return fail(PARO_ERR_UNSUPPORTED, "shared rotation: the hybrid form is built for 2..8 rows");
  if constexpr (tpw_is_pow2(TPW)) {     // (8 tiles x 16 rows exists here only: nothing of the rotation is live in a consumer's registers)
    if (a.qs == 2) {
      if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, false, 1, F, 2>, 512>(a, grid, st);
      if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, false, 1, F, 2>, 256>(a, grid, st);
      return fail(PARO_ERR_UNSUPPORTED, "group_size 64: the shared rotation is built for 4 or 8 waves per workgroup (got %d)", waves);
    }
    if constexpr (TPW < 8 && MB <= 4) {
      if (waves == 16) return launch_checked<gemv_kernel<AT, TPW, MB, 16, false, 1, F>, 1024>(a, grid, st);
    }
    if (waves == 8) return launch_checked<gemv_kernel<AT, TPW, MB, 8, false, 1, F>, 512>(a, grid, st);
    if (waves == 4) return launch_checked<gemv_kernel<AT, TPW, MB, 4, false, 1, F>, 256>(a, grid, st);
  }
  return fail(PARO_ERR_UNSUPPORTED, "shared rotation: not built for %d tiles per wave x %d waves x %d rows", TPW, waves, MB);
}

template <typename AT, int TPW, int MB, bool PREROT>
int launch_waves(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
#ifdef PARO_GEMV_DIAG   // make DIAG=1: diagnostic builds of the M = 1 kernel, selected with PARO_GEMV_PD = 11 / 21 / 31 / 41
  if constexpr (MB == 1 && !PREROT) {
    if (a.pd == 11) return launch_waves_pd<AT, TPW, MB, PREROT, 11>(a, waves, grid, st);
    if (a.pd == 21) return launch_waves_pd<AT, TPW, MB, PREROT, 21>(a, waves, grid, st);
    if (a.pd == 31) return launch_waves_pd<AT, TPW, MB, PREROT, 31>(a, waves, grid, st);
    if (a.pd == 41) return launch_waves_pd<AT, TPW, MB, PREROT, 41>(a, waves, grid, st);
    if (a.pd == 51) return launch_waves_pd<AT, TPW, MB, PREROT, 51>(a, waves, grid, st);
    if (a.pd == 61) return launch_waves_pd<AT, TPW, MB, PREROT, 61>(a, waves, grid, st);
    if (a.pd == 71) return launch_waves_pd<AT, TPW, MB, PREROT, 71>(a, waves, grid, st);
    if (a.pd == 81) return launch_waves_pd<AT, TPW, MB, PREROT, 81>(a, waves, grid, st);
  }
#endif
  if (a.shared_rot) {
    if constexpr (!PREROT && MB <= 16) return launch_waves_shared<AT, TPW, MB>(a, waves, grid, st);
    return fail(PARO_ERR_UNSUPPORTED, "the shared rotation runs on un-rotated activations, 1..16 rows");
  }
  if (a.pd != 1) return fail(PARO_ERR_UNSUPPORTED, "PARO_GEMV_PD=%d needs a diagnostic build (make DIAG=1) and batch-1 fused mode", a.pd);
  if (a.prologue != PARO_PROLOGUE_NONE || (a.hot.residual_lo | a.hot.residual_hi) || a.expert_idx || a.ar_mine || a.parts_in || a.attn_in) return launch_waves_fused<AT, TPW, MB, PREROT>(a, waves, grid, st);
  return launch_waves_pd<AT, TPW, MB, PREROT, 1>(a, waves, grid, st);
}

template <typename AT, int TPW, bool PREROT>
int launch_rows(const GemvArgs& a, int waves, dim3 grid, hipStream_t st) {
  if constexpr (TPW == 8 && !PREROT) {
    if (a.shared_rot && a.rows > 8 && a.rows <= 16) return launch_waves_shared<AT, TPW, 16>(a, waves, grid, st);
  }
  if (a.rows <= 1) return launch_waves<AT, TPW, 1, PREROT>(a, waves, grid, st);
  // (two rows: their own instantiation since round 6 -- the rotation's VALU work is per row, a 4-row build run on 2 rows rotates two
  // rows of zeros: Qwen3-4B step at 2 rows 1.329 -> see profiles/r06_rows_boundary.jsonl)
  if (a.rows <= 2) return launch_waves<AT, TPW, 2, PREROT>(a, waves, grid, st);
  if (a.rows <= 4) return launch_waves<AT, TPW, 4, PREROT>(a, waves, grid, st);
  if constexpr (tpw_is_pow2(TPW)) {
    if (a.rows <= 8) return launch_waves<AT, TPW, 8, PREROT>(a, waves, grid, st);
    if constexpr (TPW <= 4) {
      if (a.rows <= 16) return launch_waves<AT, TPW, 16, PREROT>(a, waves, grid, st);
      if constexpr (PREROT) {   // 17..64 rows: 2 / 4 MFMA row tiles per weight fragment, pre-rotated activations only
        if (a.rows <= 32) return launch_waves<AT, TPW, 32, PREROT>(a, waves, grid, st);
        if constexpr (TPW <= 2) {
          if (a.rows <= 64) return launch_waves<AT, TPW, 64, PREROT>(a, waves, grid, st);
        }
      }
    }
  }
  return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave = %d is not built for %d batch rows", TPW, a.rows);
}

// defined in gemv_inst.hip, one object per (type, pre-rotated, tiles per wave)
#define PARO_DECL_GEMV(T, P) \
  int launch_gemv_##T##_##P##_t1(const GemvArgs&, int, dim3, hipStream_t); \
  int launch_gemv_##T##_##P##_t2(const GemvArgs&, int, dim3, hipStream_t); \
  int launch_gemv_##T##_##P##_t4(const GemvArgs&, int, dim3, hipStream_t); \
  int launch_gemv_##T##_##P##_t8(const GemvArgs&, int, dim3, hipStream_t);
PARO_DECL_GEMV(f16, 0)
PARO_DECL_GEMV(f16, 1)
PARO_DECL_GEMV(bf16, 0)
PARO_DECL_GEMV(bf16, 1)
#undef PARO_DECL_GEMV

}  // namespace paro

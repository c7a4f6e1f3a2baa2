// Persistent decode engine, second build (round 5): a whole CHAIN of ParoQuant linears (batch 1) in ONE launch on the
// LDS-DMA loader / consumer geometry of /opt/skills/guides/MI355X_MICROARCH.md (rows engine-vs-launches, prefetch-credit,
// ldsdma-fill, gather-pass).
//
// What the reference does per linear: rotate -> INT4 GEMM (transformers/modules.py:57-71, vllm/plugin.py:281-311); at batch 1 a
// chain of dependent launches.  Round 4's engine (engine.hip) kept one resident grid of sixteen-wave workgroups in which every wave
// requests, unpacks, polls and publishes, and an edge was two dependent trips through memory (profiles/NOTES.md 4.2).  This build:
//
//   * one workgroup per CU = ONE LOADER wave + kE2Cons CONSUMER waves.  The loader does nothing but stream: INT4 tiles (1 KiB,
//     MFMA B-fragment order) and their scale / zero words go HBM -> LDS by LDS-DMA (`global_load_lds ... nt`) into a ring of
//     kE2Ring slots of 16 tiles; it runs ahead ACROSS phases (the weights never depend on the activations), bounded only by the
//     ring: up to 112 KiB of a CU's next tiles are in LDS when a phase's x arrives.  It owns the tile addresses; the consumers never
//     touch HBM for weights.  Phase and work records reach both roles one phase ahead (scalar registers / four vector registers).
//   * per phase (= linear) a CU owns (K-chunk s of S) x (a run of <= 16 tiles inside ONE rotation partition), tile order
//     group-major.  Consumer wave w OWNS the groups w, w + C, ... of the CU's chunk: ONE hop per edge -- it issues the 16-byte
//     write-through-coherent loads of those groups' {tag, fp32} partial sums FIRST, stages its first tiles (ring -> registers ->
//     B fragments) and requests the rotation words under their flight, checks the tags (re-polls while they are not there), adds
//     the K-chunks in slot order (+ bias), rounds ONCE to the activation type (the value the reference's linear would have
//     stored), runs the eight Givens stages in registers (up to three groups' chains interleaved), transposes through LDS into A
//     fragments -- and behind that there are only MFMAs: A = the group broadcast to all 16 rows, 4 x v_mfma_f32_16x16x32 per tile,
//     the two dequantisation sums from the matrix cores too (B = ones / the unpack's offsets), one `ds_add_f32` per tile into the
//     wave's own accumulator row.  Wide linears' remaining tiles follow two at a time.
//   * the CU's waves meet at an LDS arrival counter; each adds the rows in wave order for ITS share of the columns and publishes
//     {tag, fp32} granules (write-through stores; the data IS the flag).
//   * tags = epoch + phase; the epoch word is advanced on the device by CU 0 after an arrival count: a captured launch replays
//     without host work, nothing is ever re-armed; per-phase hop buffers; every wait is bounded and a give-up is a NaN, never a
//     silently wrong number.
//
// MEASURED (profiles/r05_engine2_timeline*.jsonl, profiles/NOTES.md round 5): parity-green at 3, 5 and 7 consumers, and SLOWER
// than both the per-call launches and round 4's engine on every model (Qwen3-4B: 7.5 .. 8.7 us per linear against 6.5 per launch
// and 6.5 in engine.hip).  Per phase the median CU needs ~1.2 us until its partial sums are there, ~1.0 us for the rotation chain,
// 1 .. 4 us for its tiles and ~0.5 us to publish, and the SLOWEST CU is another ~2 us behind the median in every phase (a poll
// that just misses the last producer pays a second trip under load).  It stays in the tree as `DecodeEngine(version=2)` /
// `bench.py --route engine2`, not as a default: the kill criterion of VERDICT r4 item 2 applies.
//
// Numerics: per (K-chunk, column) the tiles are accumulated per wave in tile order, the waves in wave order, the K-chunks in slot
// order, one rounding per linear: deterministic run to run (tests/test_gpu_engine.py), within the oracle tolerance; not
// bit-identical to the per-call kernels (other K partition).
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "chain_impl.hpp"
#include "paro_abi_experimental.h"

#ifndef PARO_E2_CONS
#define PARO_E2_CONS 7
#endif
#ifndef PARO_E2_DEPTH
#define PARO_E2_DEPTH 2
#endif

namespace paro {

constexpr int kE2Cons = PARO_E2_CONS;            // consumer waves per CU
constexpr int kE2Waves = kE2Cons + 1;            // wave 0 = the loader
constexpr int kE2Ring = 7;                       // slots in the LDS ring
constexpr int kE2SlotTiles = 16;
constexpr int kE2SlotBytes = 17 * 1024;          // 16 tiles + 16 x 64 B of scale / zero words
constexpr int kE2Depth = PARO_E2_DEPTH;          // slots the loader keeps in flight behind the one it has just issued
#ifndef PARO_E2_STAGE1
#define PARO_E2_STAGE1 6
#endif
#ifndef PARO_E2_STAGE3
#define PARO_E2_STAGE3 2
#endif
constexpr int kE2StageOne = PARO_E2_STAGE1;      // tiles multiplied from registers when a wave owns one group of the K-chunk (a multiple of 3)
constexpr int kE2StageThree = PARO_E2_STAGE3;    // ... per group when it owns more (1 or 2)
constexpr int kE2MaxNg = 64;                     // groups of one K-chunk
constexpr int kE2MaxNt = 16;                     // tiles of one CU's run
constexpr int kE2MaxSplit = 4;                   // K-chunks per linear
constexpr int kE2MaxShapes = 16;
constexpr int kE2XsStride = 136;                 // halves per rotated group in LDS (128 + 8: rows on different banks)
constexpr unsigned kE2Spin = 1u << 20;             // bound of every wait (a give-up is sticky: later waits of the launch poll once)
int validate_linear(const paro_linear_t* L);     // gemv.hip

struct alignas(16) E2PhaseHead {                 // the loader's part of a phase record: 32 bytes, read with scalar loads one phase ahead
  const u32x4* wq;
  const unsigned* sz;
  int tstride, gstride, szrow;                   // tile (t, g) = t * tstride + g * gstride (1 KiB units); words per group row of sz
  int work_off_next;                             // first E2Work of the NEXT phase (records are fetched one phase ahead)
};
struct alignas(16) E2Phase {                     // 128 bytes per phase
  E2PhaseHead h;
  // the consumers' part: fetched one phase ahead by ONE vector load (16 bytes per lane, lanes 0..7 this record, lanes 8..9 the CU's
  // E2Work) and spread to scalar registers with v_readlane at the head of the phase
  const unsigned* rot;
  const unsigned short* cs;                      // [P][K]
  const unsigned short* bias_prev;               // bias of the linear that produced this phase's input (added where its sums are completed)
  long long yoff, yoff_prev;                     // granule index of this phase's partial sums (slot s at yoff + s * N) / of the previous phase's
  int K, N, G, P;
  int S, S_prev, in_col0, N_prev;                // channel c of this phase = column in_col0 + c of the previous phase's output
  int work_off;                                  // first E2Work of this phase
  int pad[5];
};
static_assert(sizeof(E2Phase) == 128, "phase record");

struct alignas(16) E2Work {                      // 32 bytes per (distinct linear shape, CU)
  short s, p, g0, ng;                            // K-chunk, partition, first group, groups (ng = 0: nothing in this phase)
  int t0, tz0;                                   // first tile (global tile index) and its padded scale / zero tile
  short nt, pad0;
  int inv_nt;                                    // ceil(65536 / nt): i / nt == (i * inv_nt) >> 16 for i < 1024
  int ntile;                                     // nt * ng
  int pad1;
};
static_assert(sizeof(E2Work) == 32, "work record");

struct E2Args {
  const E2Phase* phases;
  const E2Work* work;
  const unsigned short* x;
  unsigned short* y;
  unsigned* ctl;                                 // [0] epoch, [1] status, [2] CUs that have read the epoch
  unsigned long long* gran;
  const unsigned short* bias_last;
  long long yoff_last;
  int n_phases, ncu, N_last, S_last;
  int work_off0;                                 // first E2Work of phase 0
  int flags;                                     // bits 0..1: how the loader yields to a hand-off of its own CU (0 not at all, 1 one burst of 4 KiB in flight, 2 stands still)
  unsigned long long* trace;                     // TRACE builds: [n_phases][ncu][8 waves][8 events] stamps of the 100 MHz counter
};

template <typename T>
__device__ __forceinline__ T e2_sload(const T* p) {          // uniform address -> scalar loads (constant address space)
  static_assert(sizeof(T) % 16 == 0, "16-byte records");
  typedef unsigned w4 __attribute__((ext_vector_type(4)));
  w4 w[sizeof(T) / 16];
  const __attribute__((address_space(4))) w4* cp = (const __attribute__((address_space(4))) w4*)p;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; ++i) w[i] = cp[i];
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
__device__ __forceinline__ unsigned long long e2_ld_gran(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void e2_st_gran(unsigned long long* p, unsigned tag, unsigned v) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Flags between the waves of a workgroup live in LDS and are moved with bare ds instructions: a wave's LDS operations execute in order,
// so a flag written behind data is seen behind it -- and no compiler-made release can drain vmcnt, which in the loader would wait for
// every LDS-DMA in flight (the "memory" clobber keeps the compiler's own accesses on their side of the statement).
__device__ __forceinline__ unsigned e2_lds_ld(const unsigned* p) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"((unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) unsigned*)p) : "memory");
  return v;
}
__device__ __forceinline__ void e2_lds_st(unsigned* p, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned*)p), "v"(v) : "memory");
}

// LDS-DMA, hidden from the compiler's waitcnt bookkeeping (cdna guide 5.7): 64 lanes x 16 B (or x 4 B) from `base + voff` to LDS `dst`
// (+ lane x 16 / x 4).  M0 is saved and restored inside the statement.  The caller counts vmcnt.
__device__ __forceinline__ unsigned long long e2_uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void e2_dma16(unsigned voff, unsigned long long base, unsigned dst) {
  unsigned keep;
  base = e2_uniform64(base);
  dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ void e2_dma4(unsigned voff, unsigned long long base, unsigned dst) {
  unsigned keep;
  base = e2_uniform64(base);
  dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
// wait until at most `pend` (rounded DOWN to a multiple of 4: conservative) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void e2_wait_vm(int pend) {
  switch (pend >> 2) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
  }
}

// one Givens stage of row 0 straight from the packed words (no GivensRegs::prepare(): 24 registers fewer per group; the converts ride
// in the shadow of the previous stage's cross-lane exchange when several groups' chains are interleaved)
template <typename GR>
__device__ __forceinline__ void e2_stage_direct(GR& g, int t) {
  const unsigned w = g.rc[t >> 2][t & 3];
  const unsigned sw = g.rc[2][t >> 2];
  const float P = (float)(int)(short)(w & 0xffffu), Q = (float)((int)w >> 16);
  const int src = (int)((sw >> (8 * (t & 3))) & 0xffu);
  const float keep = __builtin_fmaf(P, g.sa[0], Q * g.sb[0]);
  const float give = __builtin_fmaf(P, g.sb[0], -(Q * g.sa[0]));
  g.sb[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, give)));
  g.sa[0] = keep;
}

template <typename AT, bool TRACE = false>
__global__ __launch_bounds__(kE2Waves * 64) void engine2_kernel(const E2Args a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int RING_BYTES = kE2Ring * kE2SlotBytes;
  constexpr int XS_BYTES = kE2MaxNg * kE2XsStride * 2;
  constexpr int RED_FLOATS = kE2MaxNt * 16;                               // one wave's accumulator row
  constexpr int RED_BYTES = 2 * kE2Cons * RED_FLOATS * 4;                 // double-buffered by the phase's parity
  constexpr int CTL_BYTES = 192;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RING_BYTES + XS_BYTES + RED_BYTES + CTL_BYTES];
  unsigned char* ring = lds;
  unsigned short* xs = (unsigned short*)(lds + RING_BYTES);
  float* red = (float*)(lds + RING_BYTES + XS_BYTES);
  unsigned* lctl = (unsigned*)(lds + RING_BYTES + XS_BYTES + RED_BYTES);
  unsigned* l_landed = lctl;                     // slots whose DMA has landed (a count: slot q is readable once landed > q)
  unsigned* l_gath = lctl + 1;                   // consumer waves of this CU that are polling a hand-off right now (the loader thins its stream)
  unsigned* l_cdone = lctl + 4;                  // [kE2Cons] slots each consumer has finished reading
  unsigned* l_arrive = lctl + 20;                // [2] consumer waves that have finished accumulating, per phase parity (never reset)
  unsigned* l_rdone = lctl + 22;                 // [2] consumer waves that have added up and published their share, per phase parity (never reset)
  static_assert(kE2Cons <= 16, "control words");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cu = blockIdx.x;
  if (tid < CTL_BYTES / 4) lctl[tid] = 0u;
  __syncthreads();

  auto stamp = [&](int pi, int slot) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      if (lane == 0) a.trace[(((long long)pi * a.ncu + cu) * 8 + wave) * 8 + slot] = t;     // [phase][cu][wave][event]
    }
  };
  unsigned long long waited = 0;                 // TRACE: ticks of the 100 MHz counter this wave spent waiting inside the phase
  auto tick = [&]() -> unsigned long long {
    if constexpr (TRACE) return __builtin_amdgcn_s_memrealtime();
    return 0ull;
  };
  unsigned limit = kE2Spin;                      // (0 once a wait of this launch was abandoned, here or on another CU: nothing waits twice)
  auto bail = [&](unsigned spin) -> bool {
    if (spin >= limit) {
      if (lane == 0) __hip_atomic_store(a.ctl + 1, (unsigned)PARO_WS_STATUS_GIVEUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      limit = 0;
      return true;
    }
    if ((spin & 2047u) == 2047u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      limit = 0;
      return true;
    }
    return false;
  };

  // The phase and work records are fetched ONE PHASE AHEAD: read where they are needed they were two dependent scalar-cache misses
  // (~0.5 us each beside a live weight stream) at the head of every phase.  The loader keeps its 64 bytes in scalar registers; a consumer
  // wave holds the next phase's 160 bytes in four vector registers (lane j = 16-byte piece j) and spreads them with v_readlane.

  if (wave == 0) {
    // =================================================================== the LOADER: HBM -> LDS ring, ahead across the phases
    __builtin_amdgcn_s_setprio(1);
    const unsigned lds_ring = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) unsigned char*)ring);
    const unsigned voff16 = (unsigned)lane * 16u;
    const int lq = lane >> 4, ln = lane & 15;
    const int thin_mode = a.flags & 3;            // 0: never throttled, 1 / 2: see the slot loop
    const bool thin_on = thin_mode != 0;
    unsigned long long gated = 0;
    unsigned seq = 0;                            // slots issued so far
    unsigned pub = 0;                            // the landed count the consumers have been told
    unsigned rslot = 0;                          // seq % kE2Ring
    int out1 = 0, out2 = 0;                      // VMEM instructions of the previous slot and of the one before (still possibly in flight)
    E2PhaseHead ph_n = e2_sload(&a.phases[0].h);
    E2Work wk_n = e2_sload(a.work + a.work_off0 + cu);
    for (int pi = 0; pi < a.n_phases; ++pi) {
#ifdef PARO_E2_DBG_SLOADREC
      const E2Phase phf = e2_sload(a.phases + pi);
      const E2PhaseHead ph = phf.h;
      const E2Work wk = e2_sload(a.work + phf.work_off + cu);
#else
      const E2PhaseHead ph = ph_n;
      const E2Work wk = wk_n;
      if (pi + 1 < a.n_phases) {
        ph_n = e2_sload(&a.phases[pi + 1].h);
        wk_n = e2_sload(a.work + ph.work_off_next + cu);
      }
#endif
      const int ntile = wk.ntile, nt = wk.nt;
      const unsigned long long wqb = (unsigned long long)ph.wq, szb = (unsigned long long)ph.sz;
      int gi = 0, jt = 0;
      waited = 0;
      for (int i0 = 0; i0 < ntile; i0 += kE2SlotTiles) {
        const int n = min(kE2SlotTiles, ntile - i0);
        if (seq >= (unsigned)kE2Ring) {
          // the slot's previous tenant must have been read by every consumer
          const unsigned need = seq - (unsigned)kE2Ring + 1u;
          auto freed = [&]() {
            unsigned m = e2_lds_ld(l_cdone);
#pragma unroll
            for (int w = 1; w < kE2Cons; ++w) m = min(m, e2_lds_ld(l_cdone + w));
            return m >= need;
          };
          if (!freed()) {
            // ring full: nothing to gain from run-ahead right now -- let everything in flight land and say so, then wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pub = seq;
            if (lane == 0) e2_lds_st(l_landed, pub);
            out1 = out2 = 0;
            const unsigned long long t0 = tick();
            for (unsigned spin = 0; !freed(); ++spin) {
              if (bail(spin)) break;
              __builtin_amdgcn_s_sleep(2);
            }
            waited += tick() - t0;
          }
        }
        if (i0 == 0) stamp(pi, 0);
        const unsigned slot = lds_ring + rslot * (unsigned)kE2SlotBytes;
        // ---- the tiles: tile j of the slot is (group gi, tile jt) of the CU's run, group-major
#pragma unroll
        for (int j = 0; j < kE2SlotTiles; ++j) {
          if (thin_mode && (j & 3) == 0 && j < n) {
            // every 4 KiB: is a consumer wave of this CU in a hand-off (polling granules, publishing partial sums)?  Its loads and
            // stores queue behind this wave's refill burst in the CU's memory pipeline (measured: a publish of two stores per lane took
            // 2.7 us behind an unthrottled stream).  Mode 1: drain what is in flight, then go on in 4 KiB steps; mode 2: stand still.
            if (e2_lds_ld(l_gath) != 0u) {
              const unsigned long long t0 = tick();
              if (thin_mode >= 2) {
                for (unsigned spin = 0; e2_lds_ld(l_gath) != 0u; ++spin) {
                  if (bail(spin)) break;
                  __builtin_amdgcn_s_sleep(1);
                }
              } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              }
              gated += tick() - t0;
            }
          }
          if (j < n) {
            const unsigned toff = (unsigned)((wk.t0 + jt) * ph.tstride + (wk.g0 + gi) * ph.gstride);
            e2_dma16(voff16, wqb + (unsigned long long)toff * 1024ull, slot + (unsigned)j * 1024u);
            if (++jt == nt) { jt = 0; ++gi; }
          }
        }
        // ---- their scale / zero words: four tiles per instruction, lane = (tile jj = 4 q + lane / 16, column lane % 16)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q < n) {
            const int jj = 4 * q + lq;
            const int i = i0 + jj;
            const int g_ = (i * wk.inv_nt) >> 16, t_ = i - g_ * nt;
            const int tz = wk.tz0 + t_;
            const unsigned off = ((unsigned)((wk.g0 + g_) * ph.szrow) + (unsigned)((tz >> 2) * 64 + ln * 4 + (tz & 3))) * 4u;
            if (jj < n) e2_dma4(off, szb, slot + 16384u + (unsigned)q * 256u);
          }
        }
        const int cur = n + ((n + 3) >> 2);
        // How many slots stay in flight behind this wait: kE2Depth normally; ONE while a consumer wave of this CU polls a hand-off
        // (its loads queue behind the CU's own refill burst otherwise: guide row gather-pass).
        {
          const bool thin = thin_on && e2_lds_ld(l_gath) != 0u;
          int keep = cur;
          unsigned behind = 1;                   // slots still in flight after the wait (this one included)
          if (!thin && kE2Depth >= 2) { keep += out1; behind = 2; }
          if (!thin && kE2Depth >= 3) { keep += out2; behind = 3; }
          e2_wait_vm(keep);
          const unsigned now = seq + 1u >= behind ? seq + 1u - behind : 0u;      // slots 0 .. now - 1 have landed
          if (now > pub) {
            pub = now;
            if (lane == 0) e2_lds_st(l_landed, pub);
          }
        }
        out2 = out1;
        out1 = cur;
        ++seq;
        if (++rslot == (unsigned)kE2Ring) rslot = 0;
      }
      stamp(pi, 1);
      if constexpr (TRACE) {
        if (lane == 0) a.trace[(((long long)pi * a.ncu + cu) * 8 + 0) * 8 + 2] = waited;
        if (lane == 0) a.trace[(((long long)pi * a.ncu + cu) * 8 + 0) * 8 + 3] = gated;
        gated = 0;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) e2_lds_st(l_landed, seq);
    return;
  }

  // ======================================================================= the CONSUMERS
  const int cw = wave - 1;
  const unsigned epoch = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (cw == 0 && lane == 0) {
    unsigned one = 1u;
    asm volatile("" : "+v"(one) : "v"(epoch));              // the add cannot overtake the read of the epoch
    __hip_atomic_fetch_add(a.ctl + 2, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int mq = lane >> 4, n16 = lane & 15;
  const typename A::Unpack upk = A::unpack_consts();
  unsigned seq = 0;                                           // ring slots of the phases before this one
  auto fetch_rec = [&](int pi, int work_off) -> u32x4 {       // lanes 0..7: the phase record, 8..9: this CU's work record, others: lane 9's
    const int j = min(lane, 9);
    const u32x4* src = j < 8 ? (const u32x4*)(a.phases + pi) + j : (const u32x4*)(a.work + work_off + cu) + (j - 8);
    return *(CGP<u32x4>)src;
  };
  u32x4 rec_n = fetch_rec(0, a.work_off0);

  for (int pi = 0; pi < a.n_phases; ++pi) {
    E2Phase ph;
    E2Work wk;
#ifdef PARO_E2_DBG_SLOADREC
    ph = e2_sload(a.phases + pi);
    wk = e2_sload(a.work + ph.work_off + cu);
#else
    {
      unsigned w[40];
#pragma unroll
      for (int i = 0; i < 40; ++i) w[i] = (unsigned)__builtin_amdgcn_readlane((int)rec_n[i & 3], i >> 2);
      __builtin_memcpy(&ph, w, 128);
      __builtin_memcpy(&wk, w + 32, 32);
    }
#endif
    if (pi + 1 < a.n_phases) rec_n = fetch_rec(pi + 1, ph.h.work_off_next);
    const unsigned tag = epoch + (unsigned)pi + 1u;          // of this phase's output
    const unsigned tag_in = epoch + (unsigned)pi;            // of the previous phase's partial sums
    float* redp = red + ((pi & 1) * kE2Cons + cw) * RED_FLOATS;
    stamp(pi, 0);
    waited = 0;
    // the previous phase's granules through a buffer descriptor: 16-byte write-through-coherent loads (two granules = the lane's two
    // channels per instruction), offsets in bytes from the phase's first granule
    const __amdgpu_buffer_rsrc_t gsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.gran + ph.yoff_prev), 0, pi ? ph.S_prev * ph.N_prev * 8 : 0, 0x00020000);

    // ---- Wave cw OWNS the groups gi = cw, cw + C, ... of the CU's K-chunk: it completes and rotates them (the edge) and multiplies
    //      THEIR tiles.  The order of a phase is the order of its latencies:
    //        1. the hand-off loads of the wave's first (up to three) groups are ISSUED -- a round trip of ~1 us;
    //        2. under it, everything that does not depend on x: the rotation words are requested, the first tiles of those groups
    //           (eight of one group, or two each of up to three) are read from the ring and unpacked to B fragments in registers;
    //        3. the tags are checked (and polled again while they are not there), the partial sums added, rounded, rotated
    //           (eight dependent cross-lane stages), transposed through LDS into A fragments;
    //        4. behind that there are only MFMAs for the staged tiles; the rest of the tiles (wide linears) follow two at a time.
    //      Hot paths are straight-line code: a taken branch costs ~20 ns here (profiles/NOTES.md, cold code), so a missing tile or
    //      group is a clamped duplicate whose result adds +0 (or rewrites the same values), not a branch.
    const int my_groups = wk.ng > cw ? (wk.ng - cw + kE2Cons - 1) / kE2Cons : 0;
    const int nslots = (wk.ntile + kE2SlotTiles - 1) / kE2SlotTiles;
    // groups of the first rotation batch: up to three when their staged tiles sit close enough in the ring -- a wave's groups are
    // kE2Cons groups = kE2Cons * nt tiles apart, the loader is at most kE2Ring slots ahead of the slowest wave, and a wave that waited
    // for a tile beyond that window while it holds back earlier slots would wait for ever -- else one
    const int nb0 = (my_groups >= 2 && 2 * kE2Cons * wk.nt + kE2StageThree <= 4 * kE2SlotTiles) ? min(3, my_groups) : min(1, my_groups);
    unsigned landed_seen = 0;                                // (a register copy of l_landed: most tiles' slots are known to have landed)
    unsigned done = seq;                                     // slots of the ring this wave no longer needs (told to the loader)

    // (everything a guard may skip starts defined: a value that is undefined on one path is carried around the phase loop --
    // 36 registers of the previous phase's granules were spilled to scratch before these initialisers)
    GivensRegs<AT, 1> gr[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
      for (int w = 0; w < 3; ++w) gr[b].rc[w] = (u32x4){0u, 0u, 0u, 0u};
      gr[b].sa[0] = gr[b].sb[0] = 0.f;
    }
    unsigned csv[3] = {0u, 0u, 0u}, goff[3] = {0u, 0u, 0u};
    int gis[3] = {0, 0, 0};
    u32x4 q[3][kE2MaxSplit];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int s = 0; s < kE2MaxSplit; ++s) q[b][s] = (u32x4){0u, 0u, 0u, 0u};
    float x0[3] = {0.f, 0.f, 0.f}, x1[3] = {0.f, 0.f, 0.f};
    auto batch_groups = [&](int m0, int nb) {                  // the batch's groups (beyond nb: the last one again)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        gis[b] = cw + (m0 + min(b, nb - 1)) * kE2Cons;
        goff[b] = (unsigned)(ph.in_col0 + (wk.g0 + gis[b]) * 128 + 2 * lane) * 8u;
      }
    };
    auto poll_issue = [&](int nb) {                           // (uniform guards, ordered so that the common path falls through)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        if (b < nb) {
#pragma unroll
          for (int s = 0; s < kE2MaxSplit; ++s)
            if (s < ph.S_prev)
              q[b][s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(gsrc, goff[b] + (unsigned)(s * ph.N_prev) * 8u, 0, 16));   // aux 16 = sc1
        }
    };
    auto poll_ok = [&](int nb) -> bool {
      bool ok = true;
#pragma unroll
      for (int b = 0; b < 3; ++b)
        if (b < nb) {
#pragma unroll
          for (int s = 0; s < kE2MaxSplit; ++s)
            if (s < ph.S_prev) {
              const unsigned t0_ = q[b][s][1], t1_ = q[b][s][3];
              ok = ok & (t0_ == tag_in) & (t1_ == tag_in);
            }
        }
      return __all(ok);
    };
    auto rot_issue = [&](auto nbtag) {
      constexpr int NB = decltype(nbtag)::value;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int g = wk.g0 + gis[b];
        gr[b].load((CGP<unsigned>)ph.rot, (unsigned)(wk.p * ph.G + g), lane);
        csv[b] = *(CGP<unsigned>)(ph.cs + (unsigned)(wk.p * ph.K + g * 128 + 2 * lane));
      }
    };
    // the edge proper: x of the batch's groups is complete -> rounded, rotated, transposed into LDS rows
    auto finish_edge = [&](auto nbtag, int nb, bool ok) {
      constexpr int NB = decltype(nbtag)::value;
      if (pi == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const unsigned xv = *(CGP<unsigned>)(a.x + (unsigned)((wk.g0 + gis[b]) * 128 + 2 * lane));
          x0[b] = A::to_f32(xv & 0xffffu);
          x1[b] = A::to_f32(xv >> 16);
        }
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float v0 = 0.f, v1 = 0.f;
          if (b < nb) {
#pragma unroll
            for (int s = 0; s < kE2MaxSplit; ++s)
              if (s < ph.S_prev) {                            // (slot order, as the per-call reducer adds them)
                const unsigned u0 = q[b][s][0], u1 = q[b][s][2];   // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with hipcc 7.2)
                v0 += __builtin_bit_cast(float, u0);
                v1 += __builtin_bit_cast(float, u1);
              }
          }
          if (!ok) v0 = v1 = __builtin_nanf("");              // a hand-off that gave up is never a silently wrong number
          if (ph.bias_prev) {
            const unsigned bv = *(CGP<unsigned>)(ph.bias_prev + (unsigned)(ph.in_col0 + (wk.g0 + gis[b]) * 128 + 2 * lane));
            v0 += A::to_f32(bv & 0xffffu);
            v1 += A::to_f32(bv >> 16);
          }
          x0[b] = A::to_f32(A::from_f32(v0));                 // the one rounding of the producing linear
          x1[b] = A::to_f32(A::from_f32(v1));
        }
        // (a batch of fewer than NB groups rotates its last group again: the twins take that group's input and rewrite its row with the same values)
#pragma unroll
        for (int b = 1; b < NB; ++b)
          if (b >= nb) { x0[b] = x0[b - 1]; x1[b] = x1[b - 1]; }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) gr[b].seed(0, x0[b], x1[b], csv[b]);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) e2_stage_direct(gr[b], t);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        unsigned short h1, h2;
        unsigned oa, ob;
        gr[b].finish_vals(h1, h2, oa, ob);
        unsigned char* row = (unsigned char*)(xs + gis[b] * kE2XsStride);
        *(unsigned short*)(row + oa) = h1;
        *(unsigned short*)(row + ob) = h2;
      }
      asm volatile("" ::: "memory");                          // (the rows are read back by this wave only: its LDS operations stay in order)
    };
    auto poll_until = [&](int nb) -> bool {                   // the loads are in flight; true once every tag is this phase's
      if (pi == 0) return true;
      __hip_atomic_fetch_add(l_gath, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bool ok = poll_ok(nb);
      for (unsigned spin = 0; !ok; ++spin) {
        if (bail(spin)) break;
        __builtin_amdgcn_s_sleep(1);
        poll_issue(nb);
        ok = poll_ok(nb);
      }
      __hip_atomic_fetch_sub(l_gath, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return ok;
    };

    // ---- ring access: tile i of the CU's run (group-major, i = gi * nt + jt) sits in slot seq + (i >> 4) at place i & 15
    auto wait_landed = [&](int i_last) {
      const unsigned need = seq + (unsigned)(i_last >> 4);    // slots land in order
      if (landed_seen <= need) {
        for (unsigned spin = 0; (landed_seen = e2_lds_ld(l_landed)) <= need; ++spin) {
          if (bail(spin)) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
    };
    auto tile_at = [&](int i) -> const unsigned char* {
      return ring + ((seq + (unsigned)(i >> 4)) % (unsigned)kE2Ring) * kE2SlotBytes + (i & 15) * 1024;
    };
    auto sz_at = [&](int i) -> const unsigned char* {
      return ring + ((seq + (unsigned)(i >> 4)) % (unsigned)kE2Ring) * kE2SlotBytes + 16384 + (i & 15) * 64 + n16 * 4;
    };
    auto mark_free = [&](int i_next) {                        // this wave reads no tile below i_next any more (its reads so far are older LDS operations)
      const unsigned nd = seq + (unsigned)(i_next < wk.ntile ? (i_next >> 4) : nslots);
      if (nd > done) {
        done = nd;
        if (lane == 0) e2_lds_st(l_cdone + cw, done);
      }
    };
    auto unpack16 = [&](const u32x4& pk, unsigned (&o)[16]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned w4[4];
        A::unpack_fast(pk[k], w4, upk);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[k * 4 + r] = w4[r];
      }
    };
    auto mfma4 = [&](const vec8 (&af)[4], const unsigned (&b)[16]) -> f32x4 {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 w = {b[k * 4], b[k * 4 + 1], b[k * 4 + 2], b[k * 4 + 3]};
        d = A::mfma(af[k], __builtin_bit_cast(vec8, w), d);
      }
      return d;
    };
    // A fragments of a rotated group and the two sums the dequantisation needs -- sum(x) for the zero points, sum(x * off) for the
    // offsets the cheap unpack leaves in (common.hpp) -- from the matrix cores too (B = ones / the offsets' pattern: the k order of the tiles)
    auto group_frags = [&](int gi, vec8 (&af)[4], float& sx, float& so) {
      const unsigned short* xr = xs + gi * kE2XsStride + 8 * mq;     // every MFMA row carries x (16 lanes read one address: a broadcast)
#pragma unroll
      for (int k = 0; k < 4; ++k) af[k] = *(const vec8*)(xr + 32 * k);
      f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, doff = {0.f, 0.f, 0.f, 0.f};
      const u32x4 ones4 = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const u32x4 offs4 = {A::kOffFrag0, A::kOffFrag1, A::kOffFrag0, A::kOffFrag1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dsum = A::mfma(af[k], __builtin_bit_cast(vec8, ones4), dsum);
        doff = A::mfma(af[k], __builtin_bit_cast(vec8, offs4), doff);
      }
      sx = dsum[0];
      so = doff[0];
    };
    auto tile_value = [&](unsigned z, float d0, float sx, float so) -> float {
      return f16_bits_to_f32(z & 0xffffu) * __builtin_fmaf(-f16_bits_to_f32(z >> 16), sx, d0 - so);
    };
    // the tiles jt >= jb of a rotated group, two at a time; the next pair's LDS reads fly under this pair's MFMAs
    auto tiles_rest = [&](int gi, int jb, const vec8 (&af)[4], float sx, float so, int i_after) {
      if (jb >= wk.nt) return;
      const int ibase = gi * wk.nt, ilast = ibase + wk.nt - 1;
      u32x4 pa, pb;
      unsigned za, zb;
      {
        const int ia = ibase + jb, ib = min(ia + 1, ilast);
        wait_landed(ib);
        pa = *(const u32x4*)(tile_at(ia) + lane * 16);
        pb = *(const u32x4*)(tile_at(ib) + lane * 16);
        za = *(const unsigned*)sz_at(ia);
        zb = *(const unsigned*)sz_at(ib);
      }
      for (int jt = jb; jt < wk.nt; jt += 2) {
        unsigned ba[16], bb[16];
        unpack16(pa, ba);
        unpack16(pb, bb);
        const unsigned z0 = za, z1 = zb;
        const int two = jt + 1 < wk.nt ? 1 : 0;
        {
          // (the pair after this one, clamped to the group's last tile: a finished group re-reads it, nothing branches)
          const int ia = min(ibase + jt + 2, ilast), ib = min(ia + 1, ilast);
          wait_landed(ib);
          pa = *(const u32x4*)(tile_at(ia) + lane * 16);
          pb = *(const u32x4*)(tile_at(ib) + lane * 16);
          za = *(const unsigned*)sz_at(ia);
          zb = *(const unsigned*)sz_at(ib);
          mark_free(jt + 2 < wk.nt ? ia : i_after);
        }
        const f32x4 d0 = mfma4(af, ba), d1 = mfma4(af, bb);
        const float v0 = tile_value(z0, d0[0], sx, so);
        const float v1 = two ? tile_value(z1, d1[0], sx, so) : 0.f;   // (a lone tile's twin is the tile itself: it adds +0 to the same column)
        if (lane < 16) {
          __hip_atomic_fetch_add(redp + jt * 16 + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(redp + (jt + two) * 16 + lane, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    };

    if (my_groups > 0) {
      // ================= 1. the first batch's hand-off loads
      batch_groups(0, nb0);
      if (pi != 0) poll_issue(nb0);
      stamp(pi, 1);
      // ================= 2. under their flight.  This wave's accumulator row of parity pi & 1 was last read by the reductions of
      //                      phase pi - 2: every wave's share must be done before it is cleared
      if (pi >= 2) {
        const unsigned need = (unsigned)(pi >> 1) * (unsigned)kE2Cons;
        for (unsigned spin = 0; e2_lds_ld(l_rdone + (pi & 1)) < need; ++spin) {
          if (bail(spin)) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      for (int o = lane; o < wk.nt * 16; o += 64) redp[o] = 0.f;
      const int i_batch1 = my_groups > nb0 ? (cw + nb0 * kE2Cons) * wk.nt : wk.ntile; // the first tile this wave reads after its first batch
      int staged = 0;                                       // tiles per group of the first batch that were multiplied from registers
      if (nb0 == 1) {
        // ---------- one group: its first six tiles staged
        constexpr int PJ = kE2StageOne;
        staged = PJ;
        rot_issue(std::integral_constant<int, 1>{});
        const int ibase = gis[0] * wk.nt, ilast = ibase + wk.nt - 1;
        unsigned bf[PJ][16], zq[PJ];
        wait_landed(min(ibase + PJ - 1, ilast));
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
          const int i = min(ibase + j, ilast);
          const u32x4 pk = *(const u32x4*)(tile_at(i) + lane * 16);
          zq[j] = *(const unsigned*)sz_at(i);
          unpack16(pk, bf[j]);
        }
        mark_free(wk.nt > PJ ? ibase + PJ : i_batch1);
        // ================= 3.
        const bool ok = poll_until(1);
        stamp(pi, 2);
        finish_edge(std::integral_constant<int, 1>{}, 1, ok);
        stamp(pi, 3);
        // ================= 4.
        vec8 af[4];
        float sx, so;
        group_frags(gis[0], af, sx, so);
#pragma unroll
        for (int h = 0; h < PJ; h += 3) {
          f32x4 d[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) d[u] = mfma4(af, bf[h + u]);
          float v[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) v[u] = h + u < wk.nt ? tile_value(zq[h + u], d[u][0], sx, so) : 0.f;
          if (lane < 16) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
              __hip_atomic_fetch_add(redp + min(h + u, wk.nt - 1) * 16 + lane, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      } else {
        // ---------- two or three groups (or the first three of more): two tiles of each staged
        constexpr int PJ = kE2StageThree;
        staged = PJ;
        rot_issue(std::integral_constant<int, 3>{});
        unsigned bf[3][PJ][16], zq[3][PJ];
        wait_landed(gis[2] * wk.nt + min(PJ, wk.nt) - 1);
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int j = 0; j < PJ; ++j) {
            const int i = gis[m] * wk.nt + min(j, wk.nt - 1);
            const u32x4 pk = *(const u32x4*)(tile_at(i) + lane * 16);
            zq[m][j] = *(const unsigned*)sz_at(i);
            unpack16(pk, bf[m][j]);
          }
        mark_free(wk.nt > PJ ? gis[0] * wk.nt + PJ : i_batch1);
        const bool ok = poll_until(nb0);
        stamp(pi, 2);
        finish_edge(std::integral_constant<int, 3>{}, nb0, ok);
        stamp(pi, 3);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          vec8 af[4];
          float sx, so;
          group_frags(gis[m], af, sx, so);
          f32x4 d[PJ];
#pragma unroll
          for (int j = 0; j < PJ; ++j) d[j] = mfma4(af, bf[m][j]);
          const bool live = m < nb0;                        // (a twin group's tiles add +0)
          float v[PJ];
#pragma unroll
          for (int j = 0; j < PJ; ++j) v[j] = live && j < wk.nt ? tile_value(zq[m][j], d[j][0], sx, so) : 0.f;
          if (lane < 16) {
#pragma unroll
            for (int j = 0; j < PJ; ++j)
              __hip_atomic_fetch_add(redp + min(j, wk.nt - 1) * 16 + lane, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      // ================= the rest: what was not staged of the first batch's groups (wide linears), then further rotation batches
      //                   (deep K-chunks: edge, then the groups' tiles), everything two tiles at a time
      for (int m0 = 0; m0 < my_groups; m0 += (m0 == 0 ? nb0 : 3)) {
        const int nb = m0 == 0 ? nb0 : min(3, my_groups - m0);
        int jb = staged;
        if (m0 != 0) {
          jb = 0;
          batch_groups(m0, nb);
          if (pi != 0) poll_issue(nb);
          rot_issue(std::integral_constant<int, 3>{});
          const bool ok = poll_until(nb);
          finish_edge(std::integral_constant<int, 3>{}, nb, ok);
        }
        if (jb < wk.nt) {
          for (int b = 0; b < nb; ++b) {
            vec8 af[4];
            float sx, so;
            group_frags(gis[b], af, sx, so);
            const int i_after = b + 1 < nb ? gis[b + 1] * wk.nt + jb : (m0 + nb < my_groups ? (cw + (m0 + nb) * kE2Cons) * wk.nt : wk.ntile);
            tiles_rest(gis[b], jb, af, sx, so, i_after);
          }
        }
      }
    } else {
      stamp(pi, 1);
      if (pi >= 2) {
        const unsigned need = (unsigned)(pi >> 1) * (unsigned)kE2Cons;
        for (unsigned spin = 0; e2_lds_ld(l_rdone + (pi & 1)) < need; ++spin) {
          if (bail(spin)) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      for (int o = lane; o < wk.nt * 16; o += 64) redp[o] = 0.f;
    }
    mark_free(wk.ntile);
    seq += (unsigned)nslots;
    stamp(pi, 4);

    // ---- every consumer of the CU has accumulated its share: each adds the waves' rows (in wave order) for ITS share of the columns and
    //      publishes {tag, fp32}.  (One wave doing it for all -- the last to arrive -- was measured: that wave enters the next phase 4 us
    //      behind its siblings, is the last to arrive again, and the whole chain runs at the pace of one wave.)
    {
      if (lane == 0) __hip_atomic_fetch_add(l_arrive + (pi & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned all = (unsigned)((pi >> 1) + 1) * (unsigned)kE2Cons;
      for (unsigned spin = 0; e2_lds_ld(l_arrive + (pi & 1)) < all; ++spin) {
        if (bail(spin)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      stamp(pi, 5);
      const float* r0 = red + (pi & 1) * kE2Cons * RED_FLOATS;
      for (int o = cw * 64 + lane; o < wk.nt * 16; o += kE2Cons * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kE2Cons; ++w) v += r0[w * RED_FLOATS + o];
        e2_st_gran(a.gran + ph.yoff + (long long)wk.s * ph.N + (unsigned)(wk.t0 * 16 + o), tag, __builtin_bit_cast(unsigned, v));
      }
      // rows of this parity are free again once EVERY wave has read its share: counted like the arrivals
      if (lane == 0) __hip_atomic_fetch_add(l_rdone + (pi & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      stamp(pi, 6);
    }
  }

  // ---- the last phase's outputs: completed like an edge's input (slots in order, bias, one rounding), 128 columns per task
  if (cw == 0) {
    const unsigned tag_in = epoch + (unsigned)a.n_phases;
    const unsigned long long* yin = a.gran + a.yoff_last;
    const int n_fin = (a.N_last + 127) / 128;
    for (int task = cu; task < n_fin; task += a.ncu) {
      const int c = task * 128 + 2 * lane;
      if (c < a.N_last) {
        unsigned long long g0[kE2MaxSplit], g1[kE2MaxSplit];
        bool ok = false;
        for (unsigned spin = 0; !ok; ++spin) {
          ok = true;
#pragma unroll
          for (int s = 0; s < kE2MaxSplit; ++s)
            if (s < a.S_last) {
              g0[s] = e2_ld_gran(yin + (long long)s * a.N_last + c);
              g1[s] = e2_ld_gran(yin + (long long)s * a.N_last + c + 1);
              ok = ok && (unsigned)(g0[s] >> 32) == tag_in && (unsigned)(g1[s] >> 32) == tag_in;
            }
          if (!ok) {
            if (spin >= limit) { __hip_atomic_store(a.ctl + 1, (unsigned)PARO_WS_STATUS_GIVEUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int s = 0; s < kE2MaxSplit; ++s)
          if (s < a.S_last) {
            v0 += __builtin_bit_cast(float, (unsigned)g0[s]);
            v1 += __builtin_bit_cast(float, (unsigned)g1[s]);
          }
        if (!ok) v0 = v1 = __builtin_nanf("");
        if (a.bias_last) {
          const unsigned bv = *(CGP<unsigned>)(a.bias_last + c);
          v0 += A::to_f32(bv & 0xffffu);
          v1 += A::to_f32(bv >> 16);
        }
#ifndef PARO_E2_DBG_DUMPX
        *(unsigned*)(a.y + c) = (unsigned)A::from_f32(v0) | ((unsigned)A::from_f32(v1) << 16);
#endif
      }
    }
    // the next launch's tags start above this launch's: CU 0 waits (bounded) until every CU has read the epoch, clears the count and
    // stores the new epoch; the kernel boundary publishes both
    if (cu == 0) {
      unsigned seen = 0;
      for (unsigned spin = 0; seen != (unsigned)a.ncu && spin <= limit; ++spin) {
        seen = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen != (unsigned)a.ncu) __builtin_amdgcn_s_sleep(8);
      }
      if (lane == 0) {
        if (seen != (unsigned)a.ncu) a.ctl[1] = PARO_WS_STATUS_GIVEUP;
        unsigned e2 = epoch + (unsigned)a.n_phases + 2u;
        if (e2 < epoch) e2 = 1u;
        __hip_atomic_store(a.ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctl, e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host: the plan
static int e2_env_int(const char* name, int dflt) {          // experiment knobs (tools/engine2_timeline.py); read per call, never cached
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

struct E2PhasePlan {
  int S = 1;
  int ng_max = 0;                    // groups of the deepest K-chunk (what a CU gathers per slot of its predecessor)
  std::vector<E2Work> work;          // per CU
  long long cost = 0;
};

static bool e2_plan_phase(const paro_linear_t* L, int ncu, int S, E2PhasePlan& out) {
  const int G = (int)(L->K / 128), P = L->n_parts;
  if (S < 1 || S > kE2MaxSplit || S > G || ncu / S < P) return false;
  const int per = ncu / S;                                   // CUs per K-chunk
  PartTable pt;
  if (!fill_part_table(pt, P, L->part_cols, 1)) return false;
  // CUs of a K-chunk over the partitions, proportional to their tiles (largest remainder), at least one each
  std::vector<int> cp(P, 1);
  {
    int left = per - P;
    std::vector<double> want(P);
    for (int p = 0; p < P; ++p) want[p] = (double)per * (pt.tile_start[p + 1] - pt.tile_start[p]) / pt.tiles;
    while (left > 0) {
      int bi = 0;
      double bd = -1e30;
      for (int p = 0; p < P; ++p) {
        const double d = want[p] - cp[p];
        if (d > bd) { bd = d; bi = p; }
      }
      cp[bi]++;
      left--;
    }
    for (int p = 0; p < P; ++p) cp[p] = std::min(cp[p], pt.tile_start[p + 1] - pt.tile_start[p]);   // never more CUs than tiles
  }
  out.S = S;
  {
    E2Work idle{};                                           // (a CU without work in a phase: ng = nt = ntile = 0 -- nothing streamed, nothing published)
    idle.inv_nt = 65536;
    out.work.assign(ncu, idle);
  }
  long long worst = 0, total = 0;
  // the cost model's constants (shader cycles; profiles/r05_engine2_timeline*.jsonl): PARO_E2_COST=rot,poll,pair,split overrides them
  long long c_rot = 2400, c_poll = 2400, c_pair = 300, c_split = 300;
  if (const char* v = getenv("PARO_E2_COST")) sscanf(v, "%lld,%lld,%lld,%lld", &c_rot, &c_poll, &c_pair, &c_split);
  for (int s = 0; s < S; ++s) {
    const int g0 = (int)((long long)G * s / S), ng = (int)((long long)G * (s + 1) / S) - g0;
    if (ng > kE2MaxNg || ng < 1) return false;
    int c = s * per;
    for (int p = 0; p < P; ++p) {
      const int T = pt.tile_start[p + 1] - pt.tile_start[p];
      for (int i = 0; i < cp[p]; ++i, ++c) {
        const int a0 = (int)((long long)T * i / cp[p]), a1 = (int)((long long)T * (i + 1) / cp[p]);
        E2Work& w = out.work[c];
        w.s = (short)s; w.p = (short)p; w.g0 = (short)g0; w.ng = (short)ng;
        w.t0 = pt.tile_start[p] + a0;
        w.tz0 = pt.szt_start[p] + a0;
        w.nt = (short)(a1 - a0);
        if (w.nt > kE2MaxNt || w.nt < 1) return false;
        w.inv_nt = (65536 + w.nt - 1) / w.nt;
        w.ntile = w.nt * w.ng;
        // shader cycles between "x is there" and "outputs published" for the CU's busiest wave (it owns every kE2Cons-th group): its
        // rotation batches (the first of up to three groups when their staged tiles fit the ring window, else one; then threes; every
        // batch after the first polls again), then nt tiles per group, two at a time (the first batch's first tiles come from registers)
        const int gpw = (ng + kE2Cons - 1) / kE2Cons;
        const int nb0 = (gpw >= 2 && 2 * kE2Cons * w.nt + kE2StageThree <= 4 * kE2SlotTiles) ? std::min(3, gpw) : 1;
        const int batches = 1 + (gpw - nb0 + 2) / 3;
        const long long rot = (long long)batches * c_rot + (long long)(batches - 1) * c_poll;
        const int staged = nb0 == 1 ? std::min((int)w.nt, kE2StageOne) : nb0 * std::min((int)w.nt, kE2StageThree);
        const long long eat = (long long)std::max(0, (gpw * w.nt - staged + 1) / 2) * c_pair + (long long)staged * (c_pair / 4);
        worst = std::max(worst, rot + eat);
        total += w.ntile;
      }
    }
  }
  // the stream is hidden behind the edges as long as the ring holds a phase's share; what stays visible is the slowest CU's chain
  // plus what a split costs the NEXT linear (S slots to gather per group) and the chip-wide imbalance of the stream (98 cycles / KiB at 25 GB/s)
  long long mx = 0;
  for (const E2Work& w : out.work) mx = std::max<long long>(mx, w.ntile);
  out.cost = worst + std::max(0ll, mx - total / ncu) * 98;
  out.ng_max = 0;
  for (const E2Work& w : out.work) out.ng_max = std::max(out.ng_max, (int)w.ng);
  (void)c_split;
  return true;
}

struct E2PlanHost {
  std::vector<E2Phase> phases;
  std::vector<E2Work> work;
  std::vector<int> shape_off;
  long long granules = 0, yoff_last = 0;
  int S_last = 1;
};

static int e2_build_plan(const paro_engine_phase_t* ph, int n, int ncu, E2PlanHost& H) {
  if (!ph || n < 1 || n > 4096) return fail(PARO_ERR_INVALID, "engine: 1..4096 phases");
  if (ncu < kE2MaxSplit * PARO_MAX_PARTS) return fail(PARO_ERR_UNSUPPORTED, "engine: needs at least %d compute units", kE2MaxSplit * PARO_MAX_PARTS);
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    int rc = validate_linear(L);
    if (rc != PARO_OK) return rc;
    if (L->krot > 8 || !L->rot) return fail(PARO_ERR_UNSUPPORTED, "engine: krot <= 8 (packed rotation schedule)");
    if (quant_group(L->group_size) != 128) return fail(PARO_ERR_UNSUPPORTED, "engine: quantisation group_size 128");
    if (L->act_dtype != ph[0].L->act_dtype) return fail(PARO_ERR_INVALID, "engine: one activation type per chain");
    if (i > 0) {
      const paro_linear_t* Lp = ph[i - 1].L;
      if (ph[i].in_col0 < 0 || (ph[i].in_col0 & 1) || ph[i].in_col0 + L->K > Lp->N)
        return fail(PARO_ERR_INVALID, "engine: phase %d reads columns %lld..%lld of a %lld-column predecessor", i, (long long)ph[i].in_col0,
                    (long long)(ph[i].in_col0 + L->K), (long long)Lp->N);
    } else if (ph[i].in_col0 != 0) {
      return fail(PARO_ERR_INVALID, "engine: phase 0 reads x from its first element");
    }
    if ((long long)L->K * L->N / 2 > 0x7fffffffll) return fail(PARO_ERR_UNSUPPORTED, "engine: packed weights of one linear must stay below 2 GiB");
  }
  H.phases.resize(n);
  H.work.clear();
  // ---- the distinct linear shapes of the chain, their candidate plans per K-chunk count, and the choice: a linear's K-chunks are
  //      what its SUCCESSOR gathers per group (ng_next * S granule rows of 1 KiB per CU, ~c_gath cycles each through a CU's memory
  //      pipeline beside the weight stream), so the counts are chosen together -- coordinate descent from the cheapest per-shape
  //      choice over the whole chain (chains repeat a layer's few shapes: a sweep is shapes x 4 evaluations of an O(n) sum)
  struct Shape { long long K, N; int P; int cols[PARO_MAX_PARTS]; int force; E2PhasePlan cand[kE2MaxSplit + 1]; bool ok[kE2MaxSplit + 1]; int S; int off; };
  std::vector<Shape> shapes;
  std::vector<int> shape_of(n);
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    const int force = ph[i].flags & 0xf;                     // 0: the planner's choice; 1..4: this many K-chunks (tuning, tests)
    int id = -1;
    for (size_t k = 0; k < shapes.size() && id < 0; ++k) {
      bool same = shapes[k].K == L->K && shapes[k].N == L->N && shapes[k].P == L->n_parts && shapes[k].force == force;
      for (int p = 0; same && p < L->n_parts; ++p) same = shapes[k].cols[p] == L->part_cols[p];
      if (same) id = (int)k;
    }
    if (id < 0) {
      if ((int)shapes.size() >= kE2MaxShapes) return fail(PARO_ERR_UNSUPPORTED, "engine: more than %d distinct linear shapes in one chain", kE2MaxShapes);
      shapes.emplace_back();
      Shape& sh = shapes.back();
      sh.K = L->K; sh.N = L->N; sh.P = L->n_parts; sh.force = force; sh.S = 0; sh.off = 0;
      for (int p = 0; p < PARO_MAX_PARTS; ++p) sh.cols[p] = p < L->n_parts ? L->part_cols[p] : 0;
      for (int sp = 1; sp <= kE2MaxSplit; ++sp) {
        sh.ok[sp] = (!force || sp == force) && e2_plan_phase(L, ncu, sp, sh.cand[sp]);
        if (sh.ok[sp] && (sh.S == 0 || sh.cand[sp].cost < sh.cand[sh.S].cost)) sh.S = sp;
      }
      if (sh.S == 0) return fail(PARO_ERR_UNSUPPORTED, "engine: no work split for a [%lld, %lld] linear on %d compute units", (long long)L->K, (long long)L->N, ncu);
      id = (int)shapes.size() - 1;
    }
    shape_of[i] = id;
  }
  {
    long long c_gath = 90;
    if (const char* v = getenv("PARO_E2_COST_GATHER")) c_gath = atoll(v);
    auto total = [&]() {
      long long t = 0;
      for (int i = 0; i < n; ++i) {
        const Shape& sh = shapes[shape_of[i]];
        t += sh.cand[sh.S].cost;
        if (i > 0) t += c_gath * sh.cand[sh.S].ng_max * shapes[shape_of[i - 1]].S;
      }
      return t;
    };
    for (int sweep = 0; sweep < 4; ++sweep) {
      bool moved = false;
      for (Shape& sh : shapes) {
        int best = sh.S;
        long long bc = total();
        const int keep = sh.S;
        for (int sp = 1; sp <= kE2MaxSplit; ++sp) {
          if (!sh.ok[sp] || sp == keep) continue;
          sh.S = sp;
          const long long c = total();
          if (c < bc) { bc = c; best = sp; }
        }
        sh.S = best;
        moved = moved || best != keep;
      }
      if (!moved) break;
    }
  }
  for (Shape& sh : shapes) {
    sh.off = (int)H.work.size();
    H.shape_off.push_back(sh.off);
    H.work.insert(H.work.end(), sh.cand[sh.S].work.begin(), sh.cand[sh.S].work.end());
  }
  int S_prev = 1;
  long long gran = 0, yoff_prev = 0;
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    const int G = (int)(L->K / 128);
    const int off = shapes[shape_of[i]].off, S = shapes[shape_of[i]].S;
    E2Phase& e = H.phases[i];
    memset(&e, 0, sizeof(e));
    PartTable pt;
    fill_part_table(pt, L->n_parts, L->part_cols, 1);
    e.h.wq = (const u32x4*)L->wq; e.h.sz = (const unsigned*)L->sz; e.rot = (const unsigned*)L->rot; e.cs = (const unsigned short*)L->channel_scales;
    e.bias_prev = i > 0 ? (const unsigned short*)ph[i - 1].L->bias : nullptr;
    e.G = G;
    e.h.tstride = L->wq_order ? 1 : G;
    e.h.gstride = L->wq_order ? pt.tiles : 1;
    e.h.szrow = (pt.tsz >> 2) * 64;
    e.P = L->n_parts; e.S = S; e.S_prev = S_prev;
    e.in_col0 = (int)ph[i].in_col0;
    e.N = (int)L->N; e.K = (int)L->K;
    e.work_off = off;
    e.N_prev = i > 0 ? (int)ph[i - 1].L->N : 0;
    e.yoff = gran;
    gran += (long long)S * L->N;
    e.yoff_prev = yoff_prev;
    yoff_prev = e.yoff;
    S_prev = S;
  }
  for (int i = 0; i + 1 < n; ++i) H.phases[i].h.work_off_next = H.phases[i + 1].work_off;
  H.S_last = S_prev;
  H.yoff_last = yoff_prev;
  H.granules = gran;
  return PARO_OK;
}

static long long e2_plan_bytes(const E2PlanHost& H) { return (long long)(H.phases.size() * sizeof(E2Phase) + H.work.size() * sizeof(E2Work)); }

static int e2_launch(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace, int64_t workspace_bytes,
                     unsigned long long* trace, void* stream) {
  if (!e || !plan_dev || !x || !y || !workspace) return fail(PARO_ERR_INVALID, "null pointer");
  if (workspace_bytes < e->workspace_bytes) return fail(PARO_ERR_INVALID, "engine workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)e->workspace_bytes);
  if (e->n_phases < 1 || e->last_split < 1 || e->last_split > kE2MaxSplit || e->plan_bytes < (int64_t)e->n_phases * (int64_t)sizeof(E2Phase))
    return fail(PARO_ERR_INVALID, "engine descriptor was not produced by paro_engine2_plan");
  if (e->act_dtype != PARO_DTYPE_F16 && e->act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  E2Args a;
  a.phases = (const E2Phase*)plan_dev;
  a.work = (const E2Work*)((const unsigned char*)plan_dev + (size_t)e->n_phases * sizeof(E2Phase));
  a.x = (const unsigned short*)x;
  a.y = (unsigned short*)y;
  a.ctl = (unsigned*)workspace;
  a.gran = (unsigned long long*)((unsigned char*)workspace + 256);
  a.bias_last = (const unsigned short*)e->last_bias;
  a.yoff_last = e->last_out_offset;
  a.n_phases = e->n_phases;
  a.ncu = e->n_cus;
  a.N_last = (int)e->out_features;
  a.S_last = e->last_split;
  a.trace = trace;
  a.work_off0 = 0;                               // (phase 0's shape is the first one planned)
  a.flags = (e2_env_int("PARO_E2_THIN", 0) & 3);
  // every workgroup of the grid must be resident at once (they wait for each other): one workgroup per CU -- checked per DEVICE
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(PARO_ERR_LAUNCH, "engine: no current device");
    static int per_cu[64][3];
    static bool known[64][3];
    if (!known[dev][e->act_dtype]) {
      int v = 0;
      hipError_t er = e->act_dtype == PARO_DTYPE_F16
                          ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, engine2_kernel<f16, false>, kE2Waves * 64, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, engine2_kernel<bf16, false>, kE2Waves * 64, 0);
      per_cu[dev][e->act_dtype] = (er == hipSuccess && v >= 1) ? v : 0;
      known[dev][e->act_dtype] = true;
    }
    if (per_cu[dev][e->act_dtype] < 1) return fail(PARO_ERR_UNSUPPORTED, "engine: the kernel does not fit a compute unit");
    if (e->n_cus > device_cu_count()) return fail(PARO_ERR_UNSUPPORTED, "engine: planned for %d compute units, the device has %d", e->n_cus, device_cu_count());
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)e->n_cus), block(kE2Waves * 64);
  if (trace) {
    if (e->act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL((engine2_kernel<f16, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((engine2_kernel<bf16, true>), grid, block, 0, st, a);
  } else {
    if (e->act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL((engine2_kernel<f16, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((engine2_kernel<bf16, false>), grid, block, 0, st, a);
  }
  return check_launch("paro_engine2_run");
}

}  // namespace paro

extern "C" int paro_engine2_plan(const paro_engine_phase_t* phases, int n_phases, int n_cus, paro_engine_t* out) {
  using namespace paro;
  if (!out) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_cus <= 0) n_cus = device_cu_count();
  E2PlanHost H;
  int rc = e2_build_plan(phases, n_phases, n_cus, H);
  if (rc != PARO_OK) return rc;
  memset(out, 0, sizeof(*out));
  out->n_phases = n_phases;
  out->n_cus = n_cus;
  out->act_dtype = phases[0].L->act_dtype;
  out->last_split = H.S_last;
  out->n_shapes = (int)H.shape_off.size();
  for (int i = 0; i < 8; ++i) out->shape_off[i] = i < (int)H.shape_off.size() ? H.shape_off[i] : 0;
  out->last_out_offset = H.yoff_last;
  out->last_bias = phases[n_phases - 1].L->bias;
  out->plan_bytes = e2_plan_bytes(H);
  out->workspace_bytes = 256 + H.granules * 8;
  out->in_features = phases[0].L->K;
  out->out_features = phases[n_phases - 1].L->N;
  return PARO_OK;
}

extern "C" int paro_engine2_build(const paro_engine_phase_t* phases, const paro_engine_t* e, void* plan_host) {
  using namespace paro;
  if (!e || !plan_host) return fail(PARO_ERR_INVALID, "null pointer");
  E2PlanHost H;
  int rc = e2_build_plan(phases, e->n_phases, e->n_cus, H);
  if (rc != PARO_OK) return rc;
  if (e2_plan_bytes(H) != e->plan_bytes) return fail(PARO_ERR_INVALID, "engine descriptor does not belong to these phases");
  unsigned char* dst = (unsigned char*)plan_host;
  memcpy(dst, H.phases.data(), H.phases.size() * sizeof(E2Phase));
  memcpy(dst + H.phases.size() * sizeof(E2Phase), H.work.data(), H.work.size() * sizeof(E2Work));
  return PARO_OK;
}

extern "C" int paro_engine2_describe(const paro_engine_phase_t* phases, const paro_engine_t* e, int phase, int32_t* out_split,
                                     int32_t* out_max_tiles, int32_t* out_min_tiles) {
  using namespace paro;
  if (!e || phase < 0 || phase >= e->n_phases) return fail(PARO_ERR_INVALID, "bad phase");
  E2PlanHost H;
  int rc = e2_build_plan(phases, e->n_phases, e->n_cus, H);
  if (rc != PARO_OK) return rc;
  const E2Phase& p = H.phases[phase];
  int mx = 0, mn = 1 << 30;
  for (int c = 0; c < e->n_cus; ++c) {
    const E2Work& w = H.work[p.work_off + c];
    mx = std::max(mx, (int)w.ntile);
    mn = std::min(mn, (int)w.ntile);
  }
  if (out_split) *out_split = p.S;
  if (out_max_tiles) *out_max_tiles = mx;
  if (out_min_tiles) *out_min_tiles = mn;
  return PARO_OK;
}

extern "C" int paro_engine2_run(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  return paro::e2_launch(e, plan_dev, x, y, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int paro_engine2_trace(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                                  int64_t workspace_bytes, void* trace, void* stream) {
  if (!trace) return paro::fail(PARO_ERR_INVALID, "null trace buffer");
  return paro::e2_launch(e, plan_dev, x, y, workspace, workspace_bytes, (unsigned long long*)trace, stream);
}

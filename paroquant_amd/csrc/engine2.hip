// Persistent decode engine, second build (round 5): a whole CHAIN of ParoQuant linears (batch 1) in ONE launch on the
// LDS-DMA loader / consumer geometry that /opt/skills/guides/MI355X_MICROARCH.md measured to beat captured launches
// (rows engine-vs-launches, prefetch-credit, ldsdma-fill, gather-pass).
//
// What the reference does per linear: rotate -> INT4 GEMM (transformers/modules.py:57-71, vllm/plugin.py:281-311); at batch 1 a
// chain of dependent launches.  Round 4's engine (engine.hip) kept one resident grid but lost to the launches: sixteen waves per CU
// that each request, unpack, poll and publish spend ~250 instructions of bookkeeping per phase against ~75 of arithmetic, and an
// edge was two dependent trips through memory (profiles/NOTES.md 4.2).  This build changes the geometry:
//
//   * one workgroup per CU = ONE LOADER wave + kE2Cons CONSUMER waves.  The loader does nothing but stream: INT4 tiles (1 KiB,
//     MFMA B-fragment order) and their scale / zero words go HBM -> LDS by LDS-DMA (`global_load_lds ... nt`) into a ring of
//     kE2Ring slots of 16 tiles; it runs ahead ACROSS phases (the weights never depend on the activations), bounded only by the
//     ring: up to 112 KiB of a CU's next tiles are in LDS when a phase's x arrives.  It owns the plan arithmetic (tile addresses);
//     the consumers never touch HBM for weights.
//   * per phase (= linear) a CU owns (K-chunk s of S) x (a run of <= 16 tiles inside ONE rotation partition), tile order
//     group-major.  Its consumers rotate the CU's own `ng` groups themselves -- ONE hop per edge: the K-chunks' fp32 partial sums
//     {tag, fp32} of the previous linear are polled straight by the wave that needs them, added in slot order (+ bias), rounded
//     once to the activation type (the value the reference's linear would have stored), scaled by channel_scales and run through
//     the eight Givens stages in registers (GivensRegs, up to three groups' stage chains interleaved per wave); the rotated group
//     and its two dequantisation sums go to LDS behind a per-group flag, so tiles of the first groups are consumed while the last
//     are still being rotated.
//   * a consumer's tile: one ds_read_b128 (the four B fragments), one scale / zero word, A = the rotated group broadcast to all
//     16 MFMA rows, 4 x v_mfma_f32_16x16x32, one `ds_add_f32` into the wave's own accumulator row.  The waves' rows are added in a
//     fixed order and published as {tag, fp32} granules (write-through stores; the data IS the flag).
//   * tags = epoch + phase; the epoch word is advanced on the device by CU 0 after an arrival count: a captured launch replays
//     without host work, nothing is ever re-armed; per-phase hop buffers.
//
// Numerics: per (K-chunk, column) the tiles are accumulated per wave in tile order, the waves in wave order, the K-chunks in slot
// order, one rounding per linear: deterministic run to run (tests/test_gpu_engine.py), within the oracle tolerance; not
// bit-identical to the per-call kernels (other K partition).
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "chain_impl.hpp"

#ifndef PARO_E2_CONS
#define PARO_E2_CONS 3
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
constexpr int kE2MaxNg = 64;                     // groups of one K-chunk
constexpr int kE2MaxNt = 16;                     // tiles of one CU's run
constexpr int kE2MaxSplit = 4;                   // K-chunks per linear
constexpr int kE2MaxShapes = 16;
constexpr int kE2XsStride = 136;                 // halves per rotated group in LDS (128 + 8: rows on different banks)
constexpr unsigned kE2Spin = 1u << 20;             // bound of every wait (a give-up is sticky: later waits of the launch poll once)
int validate_linear(const paro_linear_t* L);     // gemv.hip

struct alignas(16) E2Phase {                     // 128 bytes per phase; read with scalar loads (constant address space)
  const u32x4* wq;
  const unsigned* sz;
  const unsigned* rot;
  const unsigned short* cs;                      // [P][K]
  const unsigned short* bias_prev;               // bias of the linear that produced this phase's input (added where its sums are completed)
  long long yoff, yoff_prev;                     // granule index of this phase's partial sums (slot s at yoff + s * N) / of the previous phase's
  int tstride, gstride, szrow, work_off;         // tile (t, g) = t * tstride + g * gstride (1 KiB units); words per group row of sz; first E2Work
  int K, N, G, P;
  int S, S_prev, in_col0, N_prev;                // channel c of this phase = column in_col0 + c of the previous phase's output
  int pad[4];
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
  unsigned long long* trace;                     // TRACE builds: [n_phases][ncu][16] stamps of the 100 MHz counter
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

template <typename AT, bool TRACE = false>
__global__ __launch_bounds__(kE2Waves * 64) void engine2_kernel(const E2Args a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int RING_BYTES = kE2Ring * kE2SlotBytes;
  constexpr int XS_BYTES = kE2MaxNg * kE2XsStride * 2;
  constexpr int XSUM_BYTES = kE2MaxNg * 8;
  constexpr int GFLAG_BYTES = kE2MaxNg * 4;
  constexpr int RED_FLOATS = kE2MaxNt * 16;                               // one wave's accumulator row
  constexpr int RED_BYTES = 2 * kE2Cons * RED_FLOATS * 4;                 // double-buffered by the phase's parity
  constexpr int CTL_BYTES = 128;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RING_BYTES + XS_BYTES + XSUM_BYTES + GFLAG_BYTES + RED_BYTES + CTL_BYTES];
  unsigned char* ring = lds;
  unsigned short* xs = (unsigned short*)(lds + RING_BYTES);
  float* xsum = (float*)(lds + RING_BYTES + XS_BYTES);
  unsigned* gflag = (unsigned*)(lds + RING_BYTES + XS_BYTES + XSUM_BYTES);
  float* red = (float*)(lds + RING_BYTES + XS_BYTES + XSUM_BYTES + GFLAG_BYTES);
  unsigned* lctl = (unsigned*)(lds + RING_BYTES + XS_BYTES + XSUM_BYTES + GFLAG_BYTES + RED_BYTES);
  unsigned* l_landed = lctl;                     // slots whose DMA has landed (a count: slot q is readable once landed > q)
  unsigned* l_cdone = lctl + 4;                  // [kE2Cons] slots each consumer has finished reading
  unsigned* l_pdone = lctl + 16;                 // [kE2Cons] phases each consumer has finished accumulating

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cu = blockIdx.x;
  if (tid < 32) lctl[tid] = 0u;
  if (tid < kE2MaxNg) gflag[tid] = 0u;
  __syncthreads();

  auto stamp = [&](int pi, int slot) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      if (lane == 0) a.trace[((long long)pi * a.ncu + cu) * 16 + slot] = t;
    }
  };

  if (wave == 0) {
    // =================================================================== the LOADER: HBM -> LDS ring, ahead across the phases
    const unsigned lds_ring = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) unsigned char*)ring);
    const unsigned voff16 = (unsigned)lane * 16u;
    const int lq = lane >> 4, ln = lane & 15;
    unsigned limit = kE2Spin;                    // (0 once a wait of this launch was abandoned: nothing waits twice)
    unsigned seq = 0;                            // slots issued so far
    unsigned pub = 0;                            // the landed count the consumers have been told
    unsigned rslot = 0;                          // seq % kE2Ring
    int out1 = 0, out2 = 0;                      // VMEM instructions of the previous slot and of the one before (still possibly in flight)
    for (int pi = 0; pi < a.n_phases; ++pi) {
      const E2Phase ph = e2_sload(a.phases + pi);
      const E2Work wk = e2_sload(a.work + ph.work_off + cu);
      const int ntile = wk.ntile, nt = wk.nt;
      const unsigned long long wqb = (unsigned long long)ph.wq, szb = (unsigned long long)ph.sz;
      int gi = 0, jt = 0;
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
            for (unsigned spin = 0; !freed(); ++spin) {
              if (spin >= limit) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; limit = 0; break; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
        }
        if (i0 == 0) stamp(pi, 8);
        const unsigned slot = lds_ring + rslot * (unsigned)kE2SlotBytes;
        // ---- the tiles: tile j of the slot is (group gi, tile jt) of the CU's run, group-major
#pragma unroll
        for (int j = 0; j < kE2SlotTiles; ++j) {
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
        // everything older than the last kE2Depth slots has landed once at most their instructions are outstanding
        {
          const unsigned now = kE2Depth >= 2 ? seq - (seq ? 1u : 0u) : seq;        // slots 0 .. now - 1 have landed after this wait
          e2_wait_vm(kE2Depth >= 2 ? cur + out1 : cur);
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
      stamp(pi, 9);
      // a phase's tail must not wait for the NEXT phase's first slot to be issued before it counts as landed when the ring is about
      // to stall anyway; the common case (the next phase has slots) publishes it one slot later
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) e2_lds_st(l_landed, seq);
    (void)out2;
    (void)pub;
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
  const float off0 = A::to_f32((unsigned short)(A::kOffFrag0 & 0xffffu)), off1 = A::to_f32((unsigned short)(A::kOffFrag1 & 0xffffu));
  unsigned seq = 0, rslot = 0;
  unsigned limit = kE2Spin;                                  // (0 once any wait of this launch was abandoned, here or on another CU)

  for (int pi = 0; pi < a.n_phases; ++pi) {
    const E2Phase ph = e2_sload(a.phases + pi);
    const E2Work wk = e2_sload(a.work + ph.work_off + cu);
    if (__hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) limit = 0;
    const unsigned tag = epoch + (unsigned)pi + 1u;          // of this phase's output (and of its rotated groups in LDS)
    const unsigned tag_in = epoch + (unsigned)pi;            // of the previous phase's partial sums
    float* redp = red + ((pi & 1) * kE2Cons + cw) * RED_FLOATS;
    if (cw == 0) stamp(pi, 0);
    for (int o = lane; o < wk.nt * 16; o += 64) redp[o] = 0.f;

    // ---- the edge: this wave's groups (gi = cw, cw + C, ...) of the CU's K-chunk, up to three stage chains interleaved
    auto edge = [&](auto nbtag, int gi0) {
      constexpr int NB = decltype(nbtag)::value;
      GivensRegs<AT, 1> gr[NB];
      unsigned csv[NB];
      float x0[NB], x1[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int g = wk.g0 + gi0 + b * kE2Cons;
        gr[b].load((CGP<unsigned>)ph.rot, (unsigned)(wk.p * ph.G + g), lane);
        csv[b] = *(CGP<unsigned>)(ph.cs + (unsigned)(wk.p * ph.K + g * 128 + 2 * lane));
      }
      if (pi == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int g = wk.g0 + gi0 + b * kE2Cons;
          const unsigned xv = *(CGP<unsigned>)(a.x + (unsigned)(g * 128 + 2 * lane));
          x0[b] = A::to_f32(xv & 0xffffu);
          x1[b] = A::to_f32(xv >> 16);
        }
      } else {
        unsigned long long q0[NB][kE2MaxSplit], q1[NB][kE2MaxSplit];
        bool ok = false;
        for (unsigned spin = 0; !ok; ++spin) {
          ok = true;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const int g = wk.g0 + gi0 + b * kE2Cons;
            const unsigned long long* src = a.gran + ph.yoff_prev + (unsigned)(ph.in_col0 + g * 128 + 2 * lane);
            // (branch-free: the slots beyond S_prev re-read the last one -- a static request count keeps the waits counted, not drained)
#pragma unroll
            for (int s = 0; s < kE2MaxSplit; ++s) {
              const long long so = (long long)min(s, ph.S_prev - 1) * ph.N_prev;
              q0[b][s] = e2_ld_gran(src + so);
              q1[b][s] = e2_ld_gran(src + so + 1);
            }
          }
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < kE2MaxSplit; ++s) ok = ok && (unsigned)(q0[b][s] >> 32) == tag_in && (unsigned)(q1[b][s] >> 32) == tag_in;
          ok = __all(ok);
          if (!ok) {
            if (spin >= limit) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; limit = 0; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int s = 0; s < kE2MaxSplit; ++s) {
            v0 += s < ph.S_prev ? __builtin_bit_cast(float, (unsigned)q0[b][s]) : 0.f;
            v1 += s < ph.S_prev ? __builtin_bit_cast(float, (unsigned)q1[b][s]) : 0.f;
          }
          if (!ok) v0 = v1 = __builtin_nanf("");              // a hand-off that gave up is never a silently wrong number
          if (ph.bias_prev) {
            const int g = wk.g0 + gi0 + b * kE2Cons;
            const unsigned bv = *(CGP<unsigned>)(ph.bias_prev + (unsigned)(ph.in_col0 + g * 128 + 2 * lane));
            v0 += A::to_f32(bv & 0xffffu);
            v1 += A::to_f32(bv >> 16);
          }
          x0[b] = A::to_f32(A::from_f32(v0));                 // the one rounding of the producing linear
          x1[b] = A::to_f32(A::from_f32(v1));
        }
      }
      if (cw == 0 && gi0 == 0) stamp(pi, 1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        gr[b].prepare();
        gr[b].seed(0, x0[b], x1[b], csv[b]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) gr[b].stage(t);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int gi = gi0 + b * kE2Cons;
        unsigned short h1, h2;
        unsigned oa, ob;
        gr[b].finish_vals(h1, h2, oa, ob);
        unsigned char* row = (unsigned char*)(xs + gi * kE2XsStride);
        *(unsigned short*)(row + oa) = h1;
        *(unsigned short*)(row + ob) = h2;
        // the two sums the dequantisation needs per group: sum(x) for the zero points, sum(x * off) for the offsets the cheap unpack
        // leaves in (common.hpp: 1024 / 64 alternating per packed register for fp16, 128 for bf16); channel c sits in register (c / 2) & 1
        const float f1 = A::to_f32(h1), f2 = A::to_f32(h2);
        const float sx = wave_sum_dpp(f1 + f2);
        const float so = wave_sum_dpp(f1 * ((oa & 4u) ? off1 : off0) + f2 * ((ob & 4u) ? off1 : off0));
        if (lane == 63) *(f32x2*)(xsum + 2 * gi) = (f32x2){sx, so};
        if (lane == 63) e2_lds_st(gflag + gi, tag);           // (a wave's LDS operations stay in order: the flag is the youngest)
      }
    };
    for (int gi0 = cw; gi0 < wk.ng; gi0 += 3 * kE2Cons) {
      const int left = (wk.ng - gi0 + kE2Cons - 1) / kE2Cons;
      if (left >= 3) edge(std::integral_constant<int, 3>{}, gi0);
      else if (left == 2) edge(std::integral_constant<int, 2>{}, gi0);
      else edge(std::integral_constant<int, 1>{}, gi0);
    }
    if (cw == 0) stamp(pi, 2);

    // ---- the CU's tiles, slot by slot: this wave's contiguous share of every slot
    int cur_g = -1;
    vec8 af[4];
    f32x2 sums = {0.f, 0.f};
    bool first = true;
    for (int i0 = 0; i0 < wk.ntile; i0 += kE2SlotTiles) {
      const int n = min(kE2SlotTiles, wk.ntile - i0);
      for (unsigned spin = 0; e2_lds_ld(l_landed) <= seq; ++spin) {
        if (spin >= limit) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; limit = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      const int lo = (n * cw) / kE2Cons, hi = (n * (cw + 1)) / kE2Cons;
      const unsigned char* slot = ring + rslot * kE2SlotBytes;
      for (int idx = lo; idx < hi; ++idx) {
        const int i = i0 + idx;
        const int gi = (i * wk.inv_nt) >> 16, jt = i - gi * wk.nt;
        const u32x4 q = *(const u32x4*)(slot + idx * 1024 + lane * 16);
        const unsigned szw = *(const unsigned*)(slot + 16384 + idx * 64 + n16 * 4);
        if (gi != cur_g) {
          for (unsigned spin = 0; e2_lds_ld(gflag + gi) != tag; ++spin) {
            if (spin >= limit) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; limit = 0; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          const unsigned short* xr = xs + gi * kE2XsStride + 8 * mq;   // every MFMA row carries x (16 lanes read one address: a broadcast)
#pragma unroll
          for (int k = 0; k < 4; ++k) af[k] = *(const vec8*)(xr + 32 * k);
          sums = *(const f32x2*)(xsum + 2 * gi);
          cur_g = gi;
        }
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned w4[4];
          A::unpack_fast(q[k], w4, upk);
          const u32x4 wv4 = {w4[0], w4[1], w4[2], w4[3]};
          d = A::mfma(af[k], __builtin_bit_cast(vec8, wv4), d);
        }
        const float s = f16_bits_to_f32(szw & 0xffffu), zf = f16_bits_to_f32(szw >> 16);
        const float v = s * __builtin_fmaf(-zf, sums[0], d[0] - sums[1]);
        if (lane < 16) __hip_atomic_fetch_add(redp + jt * 16 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cw == 0 && first) { stamp(pi, 3); first = false; }
      }
      ++seq;
      if (lane == 0) e2_lds_st(l_cdone + cw, seq);           // (release: this wave's reads of the slot are complete)
      if (++rslot == (unsigned)kE2Ring) rslot = 0;
    }
    if (cw == 0) stamp(pi, 4);

    // ---- every consumer of the CU has accumulated its share: add the waves' rows in wave order, publish {tag, fp32}
    if (lane == 0) e2_lds_st(l_pdone + cw, (unsigned)pi + 1u);
    for (unsigned spin = 0;; ++spin) {
      unsigned m = e2_lds_ld(l_pdone);
#pragma unroll
      for (int w = 1; w < kE2Cons; ++w) m = min(m, e2_lds_ld(l_pdone + w));
      if (m >= (unsigned)pi + 1u) break;
      if (spin >= limit) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; limit = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (cw == 0) stamp(pi, 5);
    {
      const float* r0 = red + (pi & 1) * kE2Cons * RED_FLOATS;
      for (int o = cw * 64 + lane; o < wk.nt * 16; o += kE2Cons * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kE2Cons; ++w) v += r0[w * RED_FLOATS + o];
        e2_st_gran(a.gran + ph.yoff + (long long)wk.s * ph.N + (unsigned)(wk.t0 * 16 + o), tag, __builtin_bit_cast(unsigned, v));
      }
    }
    if (cw == 0) stamp(pi, 6);
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
            if (spin >= limit) { a.ctl[1] = PARO_WS_STATUS_GIVEUP; break; }
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
        *(unsigned*)(a.y + c) = (unsigned)A::from_f32(v0) | ((unsigned)A::from_f32(v1) << 16);
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
struct E2PhasePlan {
  int S = 1;
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
        // cycles between "x is there" and "outputs published": the wave's rotation batches (three groups per batch), then its share of the tiles
        const int gpw = (ng + kE2Cons - 1) / kE2Cons;
        const long long rot = (long long)((gpw + 2) / 3) * 1300;
        const long long eat = (long long)((w.ntile + kE2Cons - 1) / kE2Cons) * 135;
        worst = std::max(worst, rot + eat);
        total += w.ntile;
      }
    }
  }
  // the stream is hidden behind the edges as long as the ring holds a phase's share; what stays visible is the slowest CU's chain
  // plus what a split costs the NEXT linear (S slots to poll per group) and the chip-wide imbalance of the stream (98 cycles / KiB at 25 GB/s)
  long long mx = 0;
  for (const E2Work& w : out.work) mx = std::max<long long>(mx, w.ntile);
  out.cost = worst + 150ll * S + std::max(0ll, mx - total / ncu) * 98;
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
  struct Key { long long K, N; int P; int cols[PARO_MAX_PARTS]; int force; int off; int S; };
  std::vector<Key> seen;
  int S_prev = 1;
  long long gran = 0, yoff_prev = 0;
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    const int G = (int)(L->K / 128);
    const int force = ph[i].flags & 0xf;                     // 0: the planner's choice; 1..8: this many K-chunks (tuning, tests)
    int off = -1, S = 1;
    for (const Key& k : seen) {
      bool same = k.K == L->K && k.N == L->N && k.P == L->n_parts && k.force == force;
      for (int p = 0; same && p < L->n_parts; ++p) same = k.cols[p] == L->part_cols[p];
      if (same) { off = k.off; S = k.S; break; }
    }
    if (off < 0) {
      E2PhasePlan best;
      bool any = false;
      for (int s = 1; s <= kE2MaxSplit; ++s) {
        if (force && s != force) continue;
        E2PhasePlan cand;
        if (!e2_plan_phase(L, ncu, s, cand)) continue;
        if (!any || cand.cost < best.cost) { best = cand; any = true; }
      }
      if (!any) return fail(PARO_ERR_UNSUPPORTED, "engine: no work split for a [%lld, %lld] linear on %d compute units", (long long)L->K, (long long)L->N, ncu);
      if ((int)seen.size() >= kE2MaxShapes) return fail(PARO_ERR_UNSUPPORTED, "engine: more than %d distinct linear shapes in one chain", kE2MaxShapes);
      off = (int)H.work.size();
      S = best.S;
      H.shape_off.push_back(off);
      H.work.insert(H.work.end(), best.work.begin(), best.work.end());
      Key k{L->K, L->N, L->n_parts, {0}, force, off, S};
      for (int p = 0; p < L->n_parts; ++p) k.cols[p] = L->part_cols[p];
      seen.push_back(k);
    }
    E2Phase& e = H.phases[i];
    memset(&e, 0, sizeof(e));
    PartTable pt;
    fill_part_table(pt, L->n_parts, L->part_cols, 1);
    e.wq = (const u32x4*)L->wq; e.sz = (const unsigned*)L->sz; e.rot = (const unsigned*)L->rot; e.cs = (const unsigned short*)L->channel_scales;
    e.bias_prev = i > 0 ? (const unsigned short*)ph[i - 1].L->bias : nullptr;
    e.G = G;
    e.tstride = L->wq_order ? 1 : G;
    e.gstride = L->wq_order ? pt.tiles : 1;
    e.szrow = (pt.tsz >> 2) * 64;
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

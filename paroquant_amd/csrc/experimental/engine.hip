// Persistent decode engine for gfx950: a whole CHAIN of ParoQuant linears (batch 1) in ONE launch.
//
// The reference runs `rotate -> INT4 GEMM` per linear (transformers/modules.py:57-71, vllm/plugin.py:281-311): at batch 1 that is
// a chain of dependent launches, each of which pays the kernel boundary, its ramp and the first-byte latency before it streams a few
// MB (round 1..3: 2.4 us + bytes / 7.5 TB/s per launch even for kernels that only stream; profiles/r03_overlap_probe2.jsonl).  The
// weights never depend on the activations, so one RESIDENT grid can run ahead on them:
//
//   * grid = one 16-wave workgroup per CU, alive for the whole chain; phase = one linear;
//   * per phase a CU owns (K-chunk s of `S`) x (a run of 16-column tiles inside ONE rotation partition): it needs the rotated x of
//     its own few 128-channel groups only, and its tiles are requested one unit AHEAD -- across phase boundaries -- so that when a
//     phase's x arrives its first tiles are already in registers (the run-ahead credit of the guide's engine, with the VGPR file as
//     the ring: 15 waves x 2 units x 4 KiB per CU);
//   * an edge (linear i -> linear i + 1) is two small hand-offs through memory, both as data-tagged 8-byte granules
//     {tag, value} written with ONE write-through store and polled with relaxed agent-scope loads (cdna guide G16 recipe R2):
//       hop 1  every CU's K-chunk partial sums {fp32}  ->  the ONE wave (service wave of some CU) that owns group g of partition p
//              of the consumer: it adds the `S_prev` slots in order (+ bias), rounds once to the activation type -- the value the
//              reference's linear would have stored --, multiplies by channel_scales and runs the eight Givens stages in registers;
//       hop 2  that wave publishes the rotated group {2 activations}  ->  the CUs whose K-chunk contains g gather it into LDS.
//     So a group is rotated ONCE per partition per phase (the fused GEMV re-rotated all of x in every workgroup: 20 MB of schedule
//     words through the CUs next to 8 MB of weights on Qwen3-4B qkv), and nothing is all-gathered: a CU reads
//     ng x 512 B.
//   * tags count phases inside the launch on top of an epoch word in the workspace that the LAST phase's finisher advances: a
//     captured launch replays without host work and no granule is ever re-armed.
//
// Numerics: per (K-chunk, column) the groups are accumulated in a fixed order (wave order is static), the K-chunks are added in slot
// order, one rounding to the activation type per linear: deterministic, and within the parity tolerance of the oracle
// (tests/test_gpu_engine.py); NOT bit-identical to the per-call kernels, whose K partition differs.
#include <stddef.h>
#include <string.h>
#ifndef PARO_ENG_WAVES
#define PARO_ENG_WAVES 16
#endif

#include <algorithm>
#include <vector>

#include "chain_impl.hpp"
#include "paro_abi_experimental.h"

namespace paro {

constexpr int kEngWaves = PARO_ENG_WAVES;          // waves per workgroup: all but one compute, the last is the service wave
constexpr int kEngCompute = kEngWaves - 1;
constexpr int kEngTw = kEngWaves == 16 ? 4 : 8;   // tiles per unit (two units in flight per wave: 16 waves x 128 VGPRs, or 8 x 256)
constexpr int kEngMaxSplit = 4;        // K-chunks per linear
constexpr int kEngMaxGroups = 128;     // groups of one K-chunk (LDS: 272 B each)
constexpr int kEngMaxTiles = kEngCompute * 4;   // tiles of one CU and phase (publishing threads: 16 per tile, compute waves only)
constexpr unsigned kEngSpin = 1u << 21;
constexpr int kEngPlanCap = 448;       // phases whose records fit the LDS copy of the plan (Llama-3-70B: 320)
constexpr int kEngMaxShapes = 8;       // distinct linear shapes of a chain (their work rows are cached per CU)
constexpr int kEngXsStride = 136;      // halves per group row in LDS (128 + 8 pad: the fragment reads of different rows spread over banks)
int validate_linear(const paro_linear_t* L);   // gemv.hip

struct alignas(16) EngPhase {          // 144 bytes, one per phase (cached in LDS)
  // ---- what a compute wave needs (the first 64 bytes: read as one piece)
  const u32x4* wq;
  const unsigned* sz;
  unsigned wq_bytes, sz_bytes;
  int tstride, gstride;                // tile (t, g) = chunk t * tstride + g * gstride
  int szrow;                           // words per group row of the scale / zero array
  int shape;                           // index among the chain's distinct linears (work rows, wave plans)
  int K, N;
  // Every phase has its OWN hop buffers (granule offsets into the workspace): nothing is reused inside a launch, so a CU that lags
  // behind -- e.g. one whose outputs the next linear does not read (k / v columns in a chain without attention) -- can never find a
  // granule it still waits for overwritten by a later phase.
  long long xoff;                      // rotated input {tag, two activations}: partition p, group g at xoff + p * (K / 2) + g * 64
  long long yoff;                      // partial sums {tag, fp32}: slot s at yoff + s * N
  // ---- the service wave's
  const unsigned* rot;
  const unsigned short* cs;            // [P][K]
  const unsigned short* bias_prev;     // bias of the linear that produced this phase's input (added where its partial sums are completed)
  int G, P, S, S_prev;
  int in_col0;                         // channel c of this phase = column in_col0 + c of the previous phase's output
  int n_tasks, N_prev, T;
  long long yoff_prev;
  int work_off, pad[3];                // work_off: first EngWork of this phase's shape
};
struct EngPhaseC {                     // the compute waves' view: the first 64 bytes of an EngPhase
  const u32x4* wq;
  const unsigned* sz;
  unsigned wq_bytes, sz_bytes;
  int tstride, gstride, szrow, shape, K, N;
  long long xoff, yoff;
};
static_assert(sizeof(EngPhaseC) == 64 && offsetof(EngPhase, rot) == 64, "compute view");
// a compute wave's share of (shape, CU): written once per launch into LDS by the wave itself
struct alignas(16) EngWave {
  int nu;                              // its units in a phase of this shape (0: none)
  int r, nwb;                          // unit k is group g0 + r + k * nwb of the CU's K-chunk
  int g0;
  int nt_b, tb, tzb;                   // its column block: tiles, first tile, first padded scale / zero tile
  int pad;
};
static_assert(sizeof(EngWave) == 32, "wave plan");
static_assert(sizeof(EngPhase) == 144, "phase record");

struct alignas(16) EngWork {           // 32 bytes per (phase shape, CU)
  short s, p, g0, ng;                  // K-chunk, partition, first group, groups (0: no units in this phase)
  int t0;                              // first tile (global tile index)
  int tz0;                             // its padded scale / zero tile
  short nt, nb, tw, pad0;              // tiles; column blocks (1..15) of `tw` tiles each
  int inv_tw;                          // ceil(65536 / tw): j / tw == (j * inv_tw) >> 16 for j < 64
  int pad;
};
static_assert(sizeof(EngWork) == 32, "work record");

struct EngArgs {
  const EngPhase* phases;
  const EngWork* work;
  const unsigned short* x;             // [K of phase 0]
  unsigned short* y;                   // [N of the last phase]
  unsigned* ctl;                       // [0] epoch, [1] status, [2] CUs that have read the epoch
  unsigned long long* gran;            // the hop buffers (EngPhase::yoff / xoff)
  const unsigned short* bias_last;
  long long yoff_last;
  int n_phases, ncu, N_last, S_last;
  int n_shapes, shape_off[kEngMaxShapes];   // first EngWork of every distinct shape
  unsigned long long* trace;           // TRACE builds: [n_phases][ncu][8] stamps of the 100 MHz real-time counter (paro_engine_trace)
};

template <typename T>
__device__ __forceinline__ unsigned long long ld_gran(const T* p) {
  return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_gran(unsigned long long* p, unsigned tag, unsigned v) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The plan (phase records, this CU's work rows) is copied into LDS when the kernel starts and read from there: a record is needed
// two or three times per phase on the critical path (the wave that has just consumed a unit needs the NEXT phase's record to request
// its tiles), and a read from memory there measured ~1500 cycles each -- scalar loads through the constant cache, 3500 cycles between
// "unit consumed" and "partial sums staged" (profiles/r04_engine_timeline_v3.jsonl) -- while vector loads would queue behind the
// wave's outstanding HBM tile requests.  The values are wave-uniform: pulled back into scalar registers after the LDS read.
template <typename T>
__device__ __forceinline__ T lds_uniform(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  unsigned w[sizeof(T) / 4];
  __builtin_memcpy(w, p, sizeof(T));
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)w[i]);
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}

template <typename AT, bool TRACE = false>
__global__ __launch_bounds__(kEngWaves * 64) void engine_kernel(const EngArgs a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int XS_STRIDE = kEngXsStride;
  constexpr int XS_BYTES = (kEngMaxGroups + 1) * XS_STRIDE * 2;         // + the zero row
  constexpr int RED_BYTES = kEngCompute * kEngTw * 16 * 4;
  constexpr int SVC_BYTES = 256;
  constexpr int XSUM_BYTES = kEngMaxGroups * 8;                          // per gathered group: sum(x), sum(x * unpack offset)
  constexpr int PLAN_BYTES = kEngPlanCap * (int)sizeof(EngPhase) + kEngMaxShapes * (int)sizeof(EngWork) + kEngMaxShapes * kEngWaves * (int)sizeof(EngWave);
  __shared__ __attribute__((aligned(16))) unsigned char lds[XS_BYTES + RED_BYTES + SVC_BYTES + PLAN_BYTES + XSUM_BYTES];
  float* xsum = (float*)(lds + XS_BYTES + RED_BYTES + SVC_BYTES + PLAN_BYTES);
  unsigned short* xs = (unsigned short*)lds;
  unsigned short* zrow = xs + kEngMaxGroups * XS_STRIDE;
  float* red = (float*)(lds + XS_BYTES);
  unsigned short* svc = (unsigned short*)(lds + XS_BYTES + RED_BYTES);
  const EngPhase* lphase = (const EngPhase*)(lds + XS_BYTES + RED_BYTES + SVC_BYTES);
  const EngWork* lwork = (const EngWork*)(lds + XS_BYTES + RED_BYTES + SVC_BYTES + kEngPlanCap * (int)sizeof(EngPhase));
  EngWave* lwave = (EngWave*)(lds + XS_BYTES + RED_BYTES + SVC_BYTES + kEngPlanCap * (int)sizeof(EngPhase) + kEngMaxShapes * (int)sizeof(EngWork));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cu = blockIdx.x;
  {
    // the plan into LDS: every thread moves 16-byte pieces (phase records: n_phases x 144 B; this CU's work row of every shape)
    const u32x4* src = (const u32x4*)a.phases;
    u32x4* dst = (u32x4*)(lds + XS_BYTES + RED_BYTES + SVC_BYTES);
    const int n16 = a.n_phases * (int)(sizeof(EngPhase) / 16);
    for (int i = tid; i < n16; i += kEngWaves * 64) dst[i] = src[i];
    if (tid < a.n_shapes * 2) {
      const int sh = tid >> 1, half = tid & 1;
      ((u32x4*)lwork)[tid] = ((const u32x4*)(a.work + a.shape_off[sh] + cu))[half];
    }
    __syncthreads();
  }
  // TRACE: stamp `slot` (of 16; + 16: the shader-clock counter at the same event) of (phase, CU) with the chip-wide 100 MHz counter -- service wave: 0 phase entered, 1 partial sums
  // arrived, 2 rotated group published; wave 0: 3 gather entered, 4 gathered, 5 past B1, 8 its units consumed, 6 partial sums staged,
  // 9 past B2, 7 outputs published
  auto stamp = [&](int pi, int slot) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime(), c = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        a.trace[((long long)pi * a.ncu + blockIdx.x) * 32 + slot] = t;
        a.trace[((long long)pi * a.ncu + blockIdx.x) * 32 + 16 + slot] = c;   // the shader clock at the same event
      }
    }
  };
  // the launch's base tag: advanced by the previous launch's finisher, which first waits until every CU has read it (ctl[2]; the add's
  // operand depends on the value read, so it cannot overtake the read)
  const unsigned epoch = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == kEngCompute * 64) {
    unsigned one = 1u;
    asm volatile("" : "+v"(one) : "v"(epoch));             // a data dependency the compiler cannot fold away
    __hip_atomic_fetch_add(a.ctl + 2, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned nan2 = (unsigned)A::from_f32(__builtin_nanf("")) * 0x10001u;

  // hop 2, consumer side: the rotated groups of this CU's K-chunk into LDS (every wave takes groups wave, wave + 16, ...)
  auto gather = [&](long long xoff, int K, const EngWork& w, unsigned tag) {
    if (w.ng <= 0) return;
    const unsigned long long* src = a.gran + xoff + (long long)w.p * (K / 2) + (unsigned)(w.g0 * 64 + lane);
    for (int i0 = wave; i0 < w.ng; i0 += kEngWaves * 4) {
      unsigned long long gq[4];
      bool ok = false;
      for (unsigned spin = 0; !ok; ++spin) {
        ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + q * kEngWaves;
          if (i < w.ng) gq[q] = ld_gran(src + i * 64);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + q * kEngWaves;
          if (i < w.ng) ok = ok && (unsigned)(gq[q] >> 32) == tag;
        }
        ok = __all(ok);
        if (!ok) {
          if (spin > kEngSpin) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * kEngWaves;
        if (i < w.ng) {
          const unsigned xv = ok ? (unsigned)gq[q] : nan2;                                          // (gave up: NaN)
          *(unsigned*)(xs + i * XS_STRIDE + 2 * lane) = xv;
          // The two sums the dequantisation needs per group -- sum(x) for the zero points, sum(x * off) for the offsets the cheap unpack
          // leaves in (common.hpp: 1024 / 64 alternating per register for fp16, 128 for bf16) -- ONCE per group and CU, on the VALU,
          // by the wave that gathered it: the units used to spend 8 of their 24 matrix instructions on them, and the burst after a
          // hand-off is matrix-pipe bound (profiles/r04_engine_timeline_v5.jsonl)
          const float x01 = A::to_f32(xv & 0xffffu) + A::to_f32(xv >> 16);
          const float offl = A::to_f32((unsigned short)((lane & 1) ? (A::kOffFrag1 & 0xffffu) : (A::kOffFrag0 & 0xffffu)));
          const float sx = wave_sum_dpp(x01), so = wave_sum_dpp(x01 * offl);                         // totals in lane 63
          if (lane == 63) *(f32x2*)(xsum + 2 * i) = (f32x2){sx, so};
        }
      }
    }
  };

  if (wave == kEngCompute) {
    // =================================================================== the SERVICE wave: rotation tasks, gathers, the final outputs
    // (its own loop, so that no register of the compute waves' tile buffers is live across the rotation: the first build shared one
    // loop and spilled the just-requested tiles to scratch -- a wait for HBM in front of everything)
    for (int pi = 0; pi < a.n_phases; ++pi) {
      const EngPhase ph = lds_uniform(lphase + pi);
      const EngWork w = lds_uniform(lwork + ph.shape);
      const unsigned tag = epoch + (unsigned)pi + 1u;      // of this phase's rotated input (and of its output)
      const unsigned tag_in = epoch + (unsigned)pi;        // of the previous phase's partial sums
      stamp(pi, 0);
      // the rotation tasks of this phase that live on this CU (task = (partition, group); task id = CU, CU + ncu, ...)
      for (int task = cu; task < ph.n_tasks; task += a.ncu) {
        const int p = task / ph.G, g = task - p * ph.G;
        GivensRegs<AT, 1> gr;
        gr.load((CGP<unsigned>)ph.rot, (unsigned)(p * ph.G + g), lane);
        const unsigned csv = *(CGP<unsigned>)(ph.cs + (unsigned)(p * ph.K + g * 128 + 2 * lane));
        float x0, x1;
        if (pi == 0) {
          const unsigned xv = *(CGP<unsigned>)(a.x + (unsigned)(g * 128 + 2 * lane));
          x0 = A::to_f32(xv & 0xffffu);
          x1 = A::to_f32(xv >> 16);
        } else {
          // hop 1: the S_prev partial sums of this lane's two channels, every slot polled in one batch
          const unsigned long long* src = a.gran + ph.yoff_prev + (unsigned)(ph.in_col0 + g * 128 + 2 * lane);
          unsigned long long g0[kEngMaxSplit], g1[kEngMaxSplit];
          bool ok = false;
          for (unsigned spin = 0; !ok; ++spin) {
            ok = true;
#pragma unroll
            for (int s = 0; s < kEngMaxSplit; ++s)
              if (s < ph.S_prev) {
                g0[s] = ld_gran(src + (long long)s * ph.N_prev);
                g1[s] = ld_gran(src + (long long)s * ph.N_prev + 1);
              }
#pragma unroll
            for (int s = 0; s < kEngMaxSplit; ++s)
              if (s < ph.S_prev) ok = ok && (unsigned)(g0[s] >> 32) == tag_in && (unsigned)(g1[s] >> 32) == tag_in;
            ok = __all(ok);
            if (!ok) {
              if (spin > kEngSpin) { if (lane == 0) a.ctl[1] = PARO_WS_STATUS_GIVEUP; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int s = 0; s < kEngMaxSplit; ++s)
            if (s < ph.S_prev) {
              v0 += __builtin_bit_cast(float, (unsigned)g0[s]);
              v1 += __builtin_bit_cast(float, (unsigned)g1[s]);
            }
          if (!ok) v0 = v1 = __builtin_nanf("");            // a hand-off that gave up is never a silently wrong number
          if (ph.bias_prev) {
            const unsigned bv = *(CGP<unsigned>)(ph.bias_prev + (unsigned)(ph.in_col0 + g * 128 + 2 * lane));
            v0 += A::to_f32(bv & 0xffffu);
            v1 += A::to_f32(bv >> 16);
          }
          x0 = A::to_f32(A::from_f32(v0));                  // the one rounding of the producing linear
          x1 = A::to_f32(A::from_f32(v1));
        }
        if (task == cu) stamp(pi, 1);
        gr.prepare();
        gr.seed(0, x0, x1, csv);
        gr.stages();
        gr.finish(svc);                                     // natural channel order, 128 halves
        __builtin_amdgcn_wave_barrier();
        const unsigned pr = *(const unsigned*)(svc + 2 * lane);
        __builtin_amdgcn_wave_barrier();
        st_gran(a.gran + ph.xoff + (long long)p * (ph.K / 2) + (unsigned)(g * 64 + lane), tag, pr);   // hop 2: the data IS the flag
        if (task == cu) stamp(pi, 2);
      }
      gather(ph.xoff, ph.K, w, tag);
      lds_barrier();                                        // B1
      lds_barrier();                                        // B2
    }
    // ---- the last phase's outputs: completed like a rotation task's input (slots in order, bias, one rounding), 128 columns per task
    const unsigned tag_in = epoch + (unsigned)a.n_phases;
    const unsigned long long* yin = a.gran + a.yoff_last;
    const int n_fin = (a.N_last + 127) / 128;
    for (int task = cu; task < n_fin; task += a.ncu) {
      const int c = task * 128 + 2 * lane;
      if (c < a.N_last) {
        unsigned long long g0[kEngMaxSplit], g1[kEngMaxSplit];
        bool ok = false;
        for (unsigned spin = 0; !ok; ++spin) {
          ok = true;
#pragma unroll
          for (int s = 0; s < kEngMaxSplit; ++s)
            if (s < a.S_last) {
              g0[s] = ld_gran(yin + (long long)s * a.N_last + c);
              g1[s] = ld_gran(yin + (long long)s * a.N_last + c + 1);
              ok = ok && (unsigned)(g0[s] >> 32) == tag_in && (unsigned)(g1[s] >> 32) == tag_in;
            }
          if (!ok) {
            if (spin > kEngSpin) { a.ctl[1] = PARO_WS_STATUS_GIVEUP; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int s = 0; s < kEngMaxSplit; ++s)
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
    // The next launch's tags start above this launch's.  Whoever advances the word must know that every CU has read it: CU 0's
    // service wave waits for the arrival count (bounded), clears it and stores the new epoch; the kernel boundary publishes both.
    if (cu == 0) {
      unsigned seen = 0;
      for (unsigned spin = 0; seen != (unsigned)a.ncu && spin < kEngSpin; ++spin) {
        seen = __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen != (unsigned)a.ncu) __builtin_amdgcn_s_sleep(8);
      }
      if (lane == 0) {
        if (seen != (unsigned)a.ncu) a.ctl[1] = PARO_WS_STATUS_GIVEUP;
        unsigned e2 = epoch + (unsigned)a.n_phases + 2u;
        if (e2 < epoch) e2 = 1u;                            // wrap (tags of older launches are long overwritten)
        __hip_atomic_store(a.ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctl, e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // ======================================================================= the fifteen COMPUTE waves
  // Bookkeeping is what a phase costs here, not arithmetic: sixteen waves share ONE scalar unit per CU, and the first builds spent
  // 3300 cycles between "unit consumed" and "partial sums staged" on cursor arithmetic, integer divisions, 36-dword record reads and
  // waterfall loops around buffer descriptors that had ended up in vector registers (profiles/r04_engine_timeline_v3.jsonl).  So: the
  // wave's share of every SHAPE is worked out once per launch (EngWave, in LDS), a phase needs 16 + 8 dwords from LDS, the descriptors
  // are pinned into scalar registers, and nothing divides.
  if (tid < XS_STRIDE / 4) *(u32x2*)(zrow + 4 * tid) = (u32x2){0u, 0u};   // (first read behind B1)
  for (int sh = 0; sh < a.n_shapes; ++sh) {
    const EngWork w = lds_uniform(lwork + sh);
    const int nb = w.nb;
    const int r = wave / nb, b = wave - r * nb;
    const int nwb = (kEngCompute - b + nb - 1) / nb;
    const int nt_b = min((int)w.tw, (int)w.nt - b * (int)w.tw);
    EngWave e;
    e.nu = (nt_b > 0 && r < w.ng) ? (w.ng - r + nwb - 1) / nwb : 0;
    e.r = r; e.nwb = nwb; e.g0 = w.g0;
    e.nt_b = nt_b; e.tb = w.t0 + b * w.tw; e.tzb = w.tz0 + b * w.tw;
    e.pad = 0;
    if (lane == 0) lwave[sh * kEngWaves + wave] = e;        // read back by this wave only (a wave's LDS operations stay in order)
  }
  // A-fragment addressing of a one-row product: MFMA row 0 carries x, the other fifteen rows read the zero row
  const int mq = lane >> 4, n16 = lane & 15;
  const bool avalid = n16 == 0;
  const typename A::Unpack upk = A::unpack_consts();

  struct TBuf {
    u32x4 q[kEngTw];
    unsigned szw[kEngTw];
  };
  // tile requests of one unit: `n` tiles of group g from tile t (the unused slots of a narrower unit point OUTSIDE the buffer: a
  // buffer load beyond num_records returns zeros and fetches nothing, so the request count stays static for the vmcnt bookkeeping)
  auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto load_unit = [&](TBuf& b, const EngPhaseC& ph, int g_, int t_, int tz_, int n_) {
    // every operand pinned into a scalar register HERE: a cursor that lives across loop iterations ends up in vector registers, and a
    // buffer descriptor in vector registers makes the compiler wrap each load in a waterfall loop
    const unsigned long long wqp = (unsigned long long)ph.wq, szp = (unsigned long long)ph.sz;
    const unsigned long long wq_s = ((unsigned long long)(unsigned)rfl((int)(wqp >> 32)) << 32) | (unsigned)rfl((int)wqp);
    const unsigned long long sz_s = ((unsigned long long)(unsigned)rfl((int)(szp >> 32)) << 32) | (unsigned)rfl((int)szp);
    const int g = rfl(g_), t = rfl(t_), tz = rfl(tz_), n = rfl(n_), tstride = rfl(ph.tstride);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wq_s, 0, rfl((int)ph.wq_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sz_s, 0, rfl((int)ph.sz_bytes), 0x00020000);
    const unsigned base = (unsigned)(t * tstride + g * rfl(ph.gstride)) * 1024u + (unsigned)lane * 16u;
    const unsigned step = (unsigned)tstride * 1024u;
#pragma unroll
    for (int j = 0; j < kEngTw; ++j)
      b.q[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, j < n ? base + (unsigned)j * step : 0xfffffff0u, 0, 2));   // aux 2 = nt
    const unsigned zbase = (unsigned)(g * rfl(ph.szrow)) * 4u + (unsigned)n16 * 16u;
#pragma unroll
    for (int j = 0; j < kEngTw; ++j) {
      const int ts = tz + j;
      b.szw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, j < n ? zbase + (unsigned)((ts >> 2) * 256 + (ts & 3) * 4) : 0xfffffff0u, 0, 0);
    }
  };
  // A wave's units form ONE stream across the phases: (phase 0: groups g0 + r, + nwb, ...), (phase 1: ...), ...  The REQUEST cursor
  // runs two units ahead of the consuming loop -- unit k lives in buffer k & 1, and the buffer a unit has just been consumed from is
  // re-requested at once with unit k + 2, whatever phase that is in.  So when a phase's x arrives, the wave's next two units (all a
  // Qwen3-4B phase has for it) are in registers or in flight.
  struct Cur {
    int pi, k;
    EngPhaseC ph;
    EngWave wv;
  };
  auto seek = [&](Cur& c) {                                 // onto the next phase with a unit for this wave (pi == n_phases: exhausted)
    while (c.pi < a.n_phases) {
      c.ph = lds_uniform((const EngPhaseC*)(lphase + c.pi));
      c.wv = lds_uniform(lwave + c.ph.shape * kEngWaves + wave);
      if (c.wv.nu > 0) break;
      ++c.pi;
    }
    c.k = 0;
  };
  auto request = [&](TBuf& b, Cur& c) {                     // the cursor's unit into `b`, then the cursor moves on
    const bool live = c.pi < a.n_phases;
    load_unit(b, c.ph, c.wv.g0 + c.wv.r + c.k * c.wv.nwb, c.wv.tb, c.wv.tzb, live ? c.wv.nt_b : 0);
    if (live && ++c.k >= c.wv.nu) {
      ++c.pi;
      seek(c);
    }
  };

  TBuf tA, tB;
  Cur rq;
  rq.pi = 0;
  seek(rq);
  request(tA, rq);                                          // before anything is waited for
  request(tB, rq);
  int par = 0;                                              // buffer of the next unit to consume

  for (int pi = 0; pi < a.n_phases; ++pi) {
    const EngPhaseC ph = lds_uniform((const EngPhaseC*)(lphase + pi));
    const EngWork w = lds_uniform(lwork + ph.shape);
    const EngWave wv = lds_uniform(lwave + ph.shape * kEngWaves + wave);
    const unsigned tag = epoch + (unsigned)pi + 1u;
    if (wave == 0) stamp(pi, 3);
    gather(ph.xoff, ph.K, w, tag);
    if (wave == 0) stamp(pi, 4);
    lds_barrier();                                          // B1: x of this phase is staged
    if (wave == 0) stamp(pi, 5);

    float acc[kEngTw] = {0.f, 0.f, 0.f, 0.f};
    // one unit: fragments of the rotated group, the two sums, unpack -> MFMA -> scale / zero (gemv_impl.hpp's unit, one row), then the
    // buffer is handed back to the request cursor
    const int nt_u = wv.nt_b;
    auto unit = [&](TBuf& tc, int gi) {
      vec8 af[4];
      {
        const unsigned short* afrag = (avalid ? xs + gi * XS_STRIDE : zrow) + 8 * mq;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const vec8*)(afrag + 32 * i);
      }
      const f32x2 sums = *(const f32x2*)(xsum + 2 * gi);      // {sum(x), sum(x * off)} of the group (gather)
#pragma unroll
      for (int j = 0; j < kEngTw; ++j) {
        if (j < nt_u) {                                      // (uniform: the unit's tile count; empty slots cost no matrix instructions)
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned w4[4];
            A::unpack_fast(tc.q[j][i], w4, upk);
            const u32x4 wv4 = {w4[0], w4[1], w4[2], w4[3]};
            d = A::mfma(af[i], __builtin_bit_cast(vec8, wv4), d);
          }
          const float s = f16_bits_to_f32(tc.szw[j] & 0xffffu), zf = f16_bits_to_f32(tc.szw[j] >> 16);
          acc[j] = __builtin_fmaf(s, __builtin_fmaf(-zf, sums[0], d[0] - sums[1]), acc[j]);
        }
      }
      if (wave == 0) stamp(pi, 11);
      request(tc, rq);
    };
    if (wave == 0) stamp(pi, 10);
    for (int k = 0; k < wv.nu; ++k) {
      const int gi = wv.r + k * wv.nwb;
      if (par) unit(tB, gi); else unit(tA, gi);
      par ^= 1;
    }
    if (wave == 0) stamp(pi, 8);
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < kEngTw; ++j) red[(wave * kEngTw + j) * 16 + lane] = acc[j];
    }
    if (wave == 0) stamp(pi, 6);
    lds_barrier();                                          // B2: every wave's partial sums are staged
    if (wave == 0) stamp(pi, 9);

    // ---- hop 1, producer side: this CU's outputs over its K-chunk, one {tag, fp32} granule each (wave order = group order: static)
    if (tid < w.nt * 16) {
      const int j = tid >> 4, n = tid & 15;
      const int bj = (j * w.inv_tw) >> 16, jj = j - bj * w.tw;      // j / tw without a division (inv_tw = ceil(65536 / tw), j < 64)
      const int nw = (kEngCompute - bj + w.nb - 1) / w.nb;
      float part[kEngCompute];
#pragma unroll
      for (int rr = 0; rr < kEngCompute; ++rr) part[rr] = red[((min(rr, nw - 1) * w.nb + bj) * kEngTw + jj) * 16 + n];   // all reads in flight at once
      float v = 0.f;
#pragma unroll
      for (int rr = 0; rr < kEngCompute; ++rr) v += rr < nw ? part[rr] : 0.f;
      st_gran(a.gran + ph.yoff + (long long)w.s * ph.N + (unsigned)((w.t0 + j) * 16 + n), tag, __builtin_bit_cast(unsigned, v));
    }
    if (wave == 0) stamp(pi, 7);
    // (red is rewritten only behind the next phase's B1, which every publishing thread reaches after its reads)
  }
}

// ------------------------------------------------------------------------------------------------ host: the plan
struct PhasePlan {
  int S = 1;
  std::vector<int> g0, ng;           // per K-chunk
  std::vector<EngWork> work;         // per CU
  long long cost = 0;
};

static int pick_blocks(int nt, int ng, int& nb, int& tw) {
  // column blocks (1..15; block b is shared by the waves b, b + nb, ...) of `tw` <= 4 tiles: the slowest wave's unit steps, in cycles
  long long best = -1;
  for (int cand = 1; cand <= kEngCompute; ++cand) {
    const int t = (nt + cand - 1) / cand;
    if (t > kEngTw || t < 1 || (cand - 1) * t >= nt) continue;           // (no empty blocks)
    const int nwb_min = kEngCompute / cand;                               // the blocks with the fewest waves
    const int steps = (ng + nwb_min - 1) / nwb_min;
    const long long c = (long long)steps * (150 + 130 * t);              // a unit step: fragments + the two sums, then ~130 per tile
    if (best < 0 || c < best) { best = c; nb = cand; tw = t; }
  }
  return best < 0 ? -1 : 0;
}

static bool plan_phase(const paro_linear_t* L, int ncu, int S, PhasePlan& out) {
  const int G = (int)(L->K / 128), P = L->n_parts;
  if (S < 1 || S > kEngMaxSplit || S > G || ncu / S < P) return false;
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
  out.g0.resize(S);
  out.ng.resize(S);
  {
    EngWork idle{};
    idle.nb = 1;
    idle.tw = 1;                                             // (a CU without work in a phase: ng = nt = 0; the divisors stay sane)
    idle.inv_tw = 65536;
    out.work.assign(ncu, idle);
  }
  long long worst = 0;
  for (int s = 0; s < S; ++s) {
    out.g0[s] = (int)((long long)G * s / S);
    out.ng[s] = (int)((long long)G * (s + 1) / S) - out.g0[s];
    if (out.ng[s] > kEngMaxGroups) return false;
    int c = s * per;
    for (int p = 0; p < P; ++p) {
      const int T = pt.tile_start[p + 1] - pt.tile_start[p];
      for (int i = 0; i < cp[p]; ++i, ++c) {
        const int a0 = (int)((long long)T * i / cp[p]), a1 = (int)((long long)T * (i + 1) / cp[p]);
        EngWork& w = out.work[c];
        w.s = (short)s; w.p = (short)p; w.g0 = (short)out.g0[s]; w.ng = (short)out.ng[s];
        w.t0 = pt.tile_start[p] + a0;
        w.tz0 = pt.szt_start[p] + a0;
        w.nt = (short)(a1 - a0);
        if (w.nt > kEngMaxTiles || w.nt < 1) return false;
        int nb = 1, tw = 1;
        if (pick_blocks(w.nt, w.ng, nb, tw) != 0) return false;
        w.nb = (short)nb; w.tw = (short)tw;
        w.inv_tw = (65536 + tw - 1) / tw;
        const int nwb = kEngCompute / nb;                    // (the blocks with the fewest waves)
        const long long steps = (w.ng + nwb - 1) / nwb;
        // cycles: a CU ingests a 1 KiB tile in ~96 cycles (25 GB/s); its slowest wave then runs `steps` unit steps
        worst = std::max(worst, 96ll * w.nt * w.ng + steps * (150 + 130 * tw));
      }
    }
  }
  out.cost = worst + 50ll * S;                                // (a split costs its hand-off volume: ties go to fewer chunks)
  return true;
}

struct EnginePlanHost {
  std::vector<EngPhase> phases;
  std::vector<EngWork> work;
  std::vector<int> shape_off;
  long long granules = 0, yoff_last = 0;
  int S_last = 1;
};

static int build_plan(const paro_engine_phase_t* ph, int n, int ncu, EnginePlanHost& H) {
  if (!ph || n < 1 || n > kEngPlanCap) return fail(PARO_ERR_INVALID, "engine: 1..%d phases (the plan is cached in LDS; longer chains: one engine per part)", kEngPlanCap);
  if (ncu < kEngMaxSplit * PARO_MAX_PARTS) return fail(PARO_ERR_UNSUPPORTED, "engine: needs at least %d compute units", kEngMaxSplit * PARO_MAX_PARTS);
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    int rc = validate_linear(L);
    if (rc != PARO_OK) return rc;
    if (L->krot > 8) return fail(PARO_ERR_UNSUPPORTED, "engine: krot <= 8 (packed rotation schedule)");
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
  // identical linears (same shape) share one work table
  struct Key { long long K, N; int P; int cols[PARO_MAX_PARTS]; int off; int S; int shape; };
  std::vector<Key> seen;
  int S_prev = 1;
  long long gran = 0, yoff_prev = 0;
  for (int i = 0; i < n; ++i) {
    const paro_linear_t* L = ph[i].L;
    const int G = (int)(L->K / 128);
    int off = -1, S = 1, shape = -1;
    for (const Key& k : seen) {
      bool same = k.K == L->K && k.N == L->N && k.P == L->n_parts;
      for (int p = 0; same && p < L->n_parts; ++p) same = k.cols[p] == L->part_cols[p];
      if (same) { off = k.off; S = k.S; shape = k.shape; break; }
    }
    if (off < 0) {
      PhasePlan best;
      bool any = false;
      for (int s = 1; s <= kEngMaxSplit; ++s) {
        PhasePlan cand;
        if (!plan_phase(L, ncu, s, cand)) continue;
        if (!any || cand.cost < best.cost) { best = cand; any = true; }
      }
      if (!any) return fail(PARO_ERR_UNSUPPORTED, "engine: no work split for a [%lld, %lld] linear on %d compute units", (long long)L->K, (long long)L->N, ncu);
      if ((int)seen.size() >= kEngMaxShapes) return fail(PARO_ERR_UNSUPPORTED, "engine: more than %d distinct linear shapes in one chain", kEngMaxShapes);
      off = (int)H.work.size();
      S = best.S;
      shape = (int)seen.size();
      H.shape_off.push_back(off);
      H.work.insert(H.work.end(), best.work.begin(), best.work.end());
      Key k{L->K, L->N, L->n_parts, {0}, off, S, shape};
      for (int p = 0; p < L->n_parts; ++p) k.cols[p] = L->part_cols[p];
      seen.push_back(k);
    }
    EngPhase& e = H.phases[i];
    memset(&e, 0, sizeof(e));
    PartTable pt;
    fill_part_table(pt, L->n_parts, L->part_cols, 1);
    e.wq = (const u32x4*)L->wq; e.sz = (const unsigned*)L->sz; e.rot = (const unsigned*)L->rot; e.cs = (const unsigned short*)L->channel_scales;
    e.bias_prev = i > 0 ? (const unsigned short*)ph[i - 1].L->bias : nullptr;
    e.wq_bytes = (unsigned)(L->K * L->N / 2);
    e.sz_bytes = (unsigned)((long long)G * pt.tsz * 16 * 4);
    e.G = G; e.T = pt.tiles;
    e.tstride = L->wq_order ? 1 : G;
    e.gstride = L->wq_order ? pt.tiles : 1;
    e.szrow = (pt.tsz >> 2) * 64;
    e.P = L->n_parts; e.S = S; e.S_prev = S_prev;
    e.in_col0 = (int)ph[i].in_col0;
    e.N = (int)L->N; e.K = (int)L->K;
    e.n_tasks = G * L->n_parts;
    e.work_off = off;
    e.shape = shape;
    e.N_prev = i > 0 ? (int)ph[i - 1].L->N : 0;
    e.xoff = gran;
    gran += (long long)L->n_parts * (L->K / 2);
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

static long long plan_bytes_of(const EnginePlanHost& H) { return (long long)(H.phases.size() * sizeof(EngPhase) + H.work.size() * sizeof(EngWork)); }
static long long ws_bytes_of(const EnginePlanHost& H) { return 256 + H.granules * 8; }

}  // namespace paro

extern "C" int paro_engine_plan(const paro_engine_phase_t* phases, int n_phases, int n_cus, paro_engine_t* out) {
  using namespace paro;
  if (!out) return fail(PARO_ERR_INVALID, "null pointer");
  if (n_cus <= 0) n_cus = device_cu_count();
  EnginePlanHost H;
  int rc = build_plan(phases, n_phases, n_cus, H);
  if (rc != PARO_OK) return rc;
  out->n_phases = n_phases;
  out->n_cus = n_cus;
  out->act_dtype = phases[0].L->act_dtype;
  out->last_split = H.S_last;
  out->n_shapes = (int)H.shape_off.size();
  for (int i = 0; i < 8; ++i) out->shape_off[i] = i < (int)H.shape_off.size() ? H.shape_off[i] : 0;
  out->last_out_offset = H.yoff_last;
  out->last_bias = phases[n_phases - 1].L->bias;
  out->plan_bytes = plan_bytes_of(H);
  out->workspace_bytes = ws_bytes_of(H);
  out->in_features = phases[0].L->K;
  out->out_features = phases[n_phases - 1].L->N;
  return PARO_OK;
}

extern "C" int paro_engine_build(const paro_engine_phase_t* phases, const paro_engine_t* e, void* plan_host) {
  using namespace paro;
  if (!e || !plan_host) return fail(PARO_ERR_INVALID, "null pointer");
  EnginePlanHost H;
  int rc = build_plan(phases, e->n_phases, e->n_cus, H);
  if (rc != PARO_OK) return rc;
  if (plan_bytes_of(H) != e->plan_bytes) return fail(PARO_ERR_INVALID, "engine descriptor does not belong to these phases");
  unsigned char* dst = (unsigned char*)plan_host;
  memcpy(dst, H.phases.data(), H.phases.size() * sizeof(EngPhase));
  memcpy(dst + H.phases.size() * sizeof(EngPhase), H.work.data(), H.work.size() * sizeof(EngWork));
  return PARO_OK;
}

extern "C" int paro_engine_describe(const paro_engine_phase_t* phases, const paro_engine_t* e, int phase, int32_t* out_split,
                                    int32_t* out_max_tiles, int32_t* out_min_tiles) {
  using namespace paro;
  if (!e || phase < 0 || phase >= e->n_phases) return fail(PARO_ERR_INVALID, "bad phase");
  EnginePlanHost H;
  int rc = build_plan(phases, e->n_phases, e->n_cus, H);
  if (rc != PARO_OK) return rc;
  const EngPhase& p = H.phases[phase];
  int mx = 0, mn = 1 << 30;
  for (int c = 0; c < e->n_cus; ++c) {
    const EngWork& w = H.work[p.work_off + c];
    const int t = w.nt * w.ng;
    mx = std::max(mx, t);
    mn = std::min(mn, t);
  }
  if (out_split) *out_split = p.S;
  if (out_max_tiles) *out_max_tiles = mx;
  if (out_min_tiles) *out_min_tiles = mn;
  return PARO_OK;
}

namespace paro {
static int engine_launch(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace, int64_t workspace_bytes,
                         unsigned long long* trace, void* stream) {
  if (!e || !plan_dev || !x || !y || !workspace) return fail(PARO_ERR_INVALID, "null pointer");
  if (workspace_bytes < e->workspace_bytes) return fail(PARO_ERR_INVALID, "engine workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)e->workspace_bytes);
  if (e->n_phases < 1 || e->last_split < 1 || e->last_split > kEngMaxSplit || e->plan_bytes < (int64_t)e->n_phases * (int64_t)sizeof(EngPhase))
    return fail(PARO_ERR_INVALID, "engine descriptor was not produced by paro_engine_plan");
  if (e->act_dtype != PARO_DTYPE_F16 && e->act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  EngArgs a;
  a.phases = (const EngPhase*)plan_dev;
  a.work = (const EngWork*)((const unsigned char*)plan_dev + (size_t)e->n_phases * sizeof(EngPhase));
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
  if (e->n_shapes < 1 || e->n_shapes > kEngMaxShapes || e->n_phases > kEngPlanCap) return fail(PARO_ERR_INVALID, "engine descriptor was not produced by paro_engine_plan");
  a.n_shapes = e->n_shapes;
  for (int i = 0; i < kEngMaxShapes; ++i) a.shape_off[i] = e->shape_off[i];
  // every workgroup of the grid must be resident at once (they wait for each other): one 16-wave workgroup per CU
  {
    // (per DEVICE: another GPU of the process may differ; the worst a racing first call does is ask twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(PARO_ERR_LAUNCH, "engine: no current device");
    static int per_cu[64][3];
    static bool known[64][3];
    if (!known[dev][e->act_dtype]) {
      per_cu[dev][e->act_dtype] = -1;
      known[dev][e->act_dtype] = true;
    }
    int& per = per_cu[dev][e->act_dtype];
    if (per < 0) {
      int v = 0;
      hipError_t er = e->act_dtype == PARO_DTYPE_F16
                          ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, engine_kernel<f16, false>, kEngWaves * 64, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, engine_kernel<bf16, false>, kEngWaves * 64, 0);
      per = (er == hipSuccess && v >= 1) ? v : 0;
    }
    if (per < 1) return fail(PARO_ERR_UNSUPPORTED, "engine: the kernel does not fit a compute unit");
    if (e->n_cus > device_cu_count()) return fail(PARO_ERR_UNSUPPORTED, "engine: planned for %d compute units, the device has %d", e->n_cus, device_cu_count());
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)e->n_cus), block(kEngWaves * 64);
  if (trace) {
    if (e->act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL((engine_kernel<f16, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((engine_kernel<bf16, true>), grid, block, 0, st, a);
  } else {
    if (e->act_dtype == PARO_DTYPE_F16) hipLaunchKernelGGL((engine_kernel<f16, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((engine_kernel<bf16, false>), grid, block, 0, st, a);
  }
  return check_launch("paro_engine_run");
}
}  // namespace paro

extern "C" int paro_engine_run(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  return paro::engine_launch(e, plan_dev, x, y, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int paro_engine_trace(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                                 int64_t workspace_bytes, void* trace, void* stream) {
  if (!trace) return paro::fail(PARO_ERR_INVALID, "null trace buffer");
  return paro::engine_launch(e, plan_dev, x, y, workspace, workspace_bytes, (unsigned long long*)trace, stream);
}

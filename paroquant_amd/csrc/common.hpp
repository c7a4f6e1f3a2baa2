// Shared device/host helpers for libparo_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "paro_abi.h"

namespace paro {

// ---- error reporting (thread-local, no exceptions across the ABI) -------------
char* error_buffer();
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PARO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PARO_OK;
}

// ---- merged-partition bookkeeping (passed by value to kernels) ------------------------
// tile_start : first 16-column tile of each partition (prefix sums of part_cols / 16)
// szt_start  : the same in the padded tile space of the scale/zero words (each partition rounded up to 8 tiles)
// cb_start   : first column block (= workgroup x-index) of each partition for a given tiles-per-block
struct PartTable {
  int nparts, tiles, tsz, cbs;
  int tile_start[PARO_MAX_PARTS + 1];
  int szt_start[PARO_MAX_PARTS + 1];
  int cb_start[PARO_MAX_PARTS + 1];
  __host__ __device__ int part_of_tile(int t) const {
    int p = 0;
#pragma unroll
    for (int q = 1; q < PARO_MAX_PARTS; ++q)
      if (q < nparts && t >= tile_start[q]) p = q;
    return p;
  }
  __host__ __device__ int part_of_cb(int cb) const {
    int p = 0;
#pragma unroll
    for (int q = 1; q < PARO_MAX_PARTS; ++q)
      if (q < nparts && cb >= cb_start[q]) p = q;
    return p;
  }
};

// ---- the one-shot all-reduce buffer (allreduce.hip; shared with the GEMV's all-reduce epilogue, gemv_impl.hpp):
//   u32 words: [1] status   [16 + wg] epoch of workgroup wg of the standalone kernel (the GEMV epilogue keeps its give-up
//   flag and one epoch per 16-column tile in ORDINARY device memory: paro_fusion_t.ar_state, kArStateTiles + max_elems / 16 words);   region A at 4096: (set, rank) slots of {two activations, tag} granules,
//   max_elems / 2 per slot;   region B behind it: (set, rank) slots of {fp32 partial, tag} granules, max_elems per slot.
//   The two users keep separate regions AND separate epochs: a tag is only ever compared with its own user's sequence.
constexpr int kArEpochOff = 64, kArDataOff = 4096, kArMaxWorld = 16, kArThreads = 1024;
constexpr int kArMaxWgs = (kArDataOff - kArEpochOff) / 4;
constexpr int kArStateTiles = 16;   // first tile epoch word of paro_fusion_t.ar_state
inline long long ar_slot_a(long long max_elems) { return ((max_elems / 2 + 31) / 32) * 32; }
inline long long ar_slot_b(long long max_elems) { return ((max_elems + 31) / 32) * 32; }
inline long long ar_region_b_off(int world, long long max_elems) { return kArDataOff + 2ll * world * ar_slot_a(max_elems) * 8; }
inline long long ar_buffer_bytes(int world, long long max_elems) { return ar_region_b_off(world, max_elems) + 2ll * world * ar_slot_b(max_elems) * 8; }

// quantisation group of a layer: 128 (also for 0 = unset) or 64; -1 = unsupported
inline int quant_group(int group_size) { return (group_size == 0 || group_size == 128) ? 128 : (group_size == 64 ? 64 : -1); }

inline bool fill_part_table(PartTable& pt, int nparts, const int32_t* part_cols, int tiles_per_cb) {
  if (nparts < 1 || nparts > PARO_MAX_PARTS || !part_cols || tiles_per_cb < 1) return false;
  pt.nparts = nparts;
  int tiles = 0, tsz = 0, cbs = 0;
  for (int i = 0; i <= PARO_MAX_PARTS; ++i) {
    pt.tile_start[i] = tiles;
    pt.szt_start[i] = tsz;
    pt.cb_start[i] = cbs;
    if (i < nparts) {
      if (part_cols[i] <= 0 || part_cols[i] % 16 != 0) return false;
      const int t = part_cols[i] / 16;
      tiles += t;
      tsz += (t + 7) / 8 * 8;
      cbs += (t + tiles_per_cb - 1) / tiles_per_cb;
    }
  }
  pt.tiles = tiles;
  pt.tsz = tsz;
  pt.cbs = cbs;
  return true;
}

// The AQL packet index of the running dispatch on its queue: the same in every workgroup of a launch, new for every launch and for every
// replay of a captured graph (tools/probes/dispatch_id_probe.hip) -- a launch tag that costs no memory access.  clang has no builtin
// for it; the LLVM intrinsic is reachable by name.
extern "C" __device__ unsigned long long paro_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");

// ---- vector types ---------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- scalar conversions -----------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {  // round-to-nearest-even
  unsigned u = __builtin_bit_cast(unsigned, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) {
  return (float)__builtin_bit_cast(f16, b);
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (f16)f);
}

// Activation-type traits: 16-bit storage, MFMA flavour, INT4 -> (16 + q) magic unpack.
template <typename T>
struct Act;

template <>
struct Act<f16> {
  typedef f16x8 vec8;
  static constexpr int kDtype = PARO_DTYPE_F16;
  static constexpr unsigned kMask = 0x03C003C0u;   // nibble lands on mantissa bits 6..9
  static constexpr unsigned kMagic = 0x4C004C00u;  // 16.0h | nibble<<6  == 16 + q exactly
  static constexpr unsigned kOnes = 0x3C003C00u;   // (1.0h, 1.0h)
  __device__ static __forceinline__ float to_f32(unsigned short b) { return f16_bits_to_f32(b); }
  __device__ static __forceinline__ unsigned short from_f32(float f) { return f32_to_f16_bits(f); }
  // word w holds 8 nibbles; out[v] = (16 + nibble v, 16 + nibble v+4) as a packed pair
  __device__ static __forceinline__ void unpack(unsigned w, unsigned (&o)[4]) {
    o[0] = ((w << 6) & kMask) | kMagic;
    o[1] = ((w << 2) & kMask) | kMagic;
    o[2] = ((w >> 2) & kMask) | kMagic;
    o[3] = ((w >> 6) & kMask) | kMagic;
  }
  // Cheaper unpack for the VALU-bound GEMV: no shift for the nibbles that already sit on mantissa
  // bits 0..3 (| 1024.0h -> 1024 + q) and 4..7 (| 64.0h -> 64 + q); one shift brings the other two
  // pairs there.  5 VALU per word (1 shift + 4 v_and_or_b32); the per-element offsets (1024, 64, 1024, 64 per
  // register) are removed with one extra MFMA against kOffFrag:  sum_k x_k off_k.
  // The masks / magic numbers as OPAQUE register values (laundered once per kernel through an empty asm):
  // gfx9 VOP3 takes no literals, so with compile-time constants the compiler emits v_and + v_or; with
  // register operands it selects v_and_or_b32 -- 5 VALU per word instead of 9.
  struct Unpack {
    unsigned m0, m1, k0, k1;
  };
  __device__ static __forceinline__ Unpack unpack_consts() {
    Unpack u = {0x000F000Fu, 0x00F000F0u, 0x64006400u, 0x54005400u};
    asm volatile("" : "+s"(u.m0), "+s"(u.m1), "+v"(u.k0), "+v"(u.k1));
    return u;
  }
  __device__ static __forceinline__ void unpack_fast(unsigned w, unsigned (&o)[4], const Unpack& u) {
    const unsigned t = w >> 8;
    o[0] = (w & u.m0) | u.k0;
    o[1] = (w & u.m1) | u.k1;
    o[2] = (t & u.m0) | u.k0;
    o[3] = (t & u.m1) | u.k1;
  }
  static constexpr unsigned kOffFrag0 = 0x64006400u, kOffFrag1 = 0x54005400u;  // registers 0/2 and 1/3
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <>
struct Act<bf16> {
  typedef bf16x8 vec8;
  static constexpr int kDtype = PARO_DTYPE_BF16;
  static constexpr unsigned kMask = 0x00780078u;   // mantissa bits 3..6
  static constexpr unsigned kMagic = 0x41804180u;  // 16.0bf16 | nibble<<3 == 16 + q exactly
  static constexpr unsigned kOnes = 0x3F803F80u;
  __device__ static __forceinline__ float to_f32(unsigned short b) { return bf16_bits_to_f32(b); }
  __device__ static __forceinline__ unsigned short from_f32(float f) { return f32_to_bf16_bits(f); }
  __device__ static __forceinline__ void unpack(unsigned w, unsigned (&o)[4]) {
    o[0] = ((w << 3) & kMask) | kMagic;
    o[1] = ((w >> 1) & kMask) | kMagic;
    o[2] = ((w >> 5) & kMask) | kMagic;
    o[3] = ((w >> 9) & kMask) | kMagic;
  }
  // bf16 has 7 mantissa bits: only bits 0..3 can hold a nibble without touching the exponent
  // (| 128.0 -> 128 + q, exact in 8 significant bits); uniform offset 128, 7 VALU per word (v_and_or_b32).
  struct Unpack {
    unsigned m0, k0;
  };
  __device__ static __forceinline__ Unpack unpack_consts() {
    Unpack u = {0x000F000Fu, 0x43004300u};
    asm volatile("" : "+s"(u.m0), "+v"(u.k0));
    return u;
  }
  __device__ static __forceinline__ void unpack_fast(unsigned w, unsigned (&o)[4], const Unpack& u) {
    o[0] = (w & u.m0) | u.k0;
    o[1] = ((w >> 4) & u.m0) | u.k0;
    o[2] = ((w >> 8) & u.m0) | u.k0;
    o[3] = ((w >> 12) & u.m0) | u.k0;
  }
  static constexpr unsigned kOffFrag0 = 0x43004300u, kOffFrag1 = 0x43004300u;
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// Load a rotation/quantisation parameter of dtype code `dt` as fp32.
__device__ __forceinline__ float load_param(const void* p, int64_t i, int dt) {
  if (dt == PARO_DTYPE_F16) return f16_bits_to_f32(((const unsigned short*)p)[i]);
  if (dt == PARO_DTYPE_BF16) return bf16_bits_to_f32(((const unsigned short*)p)[i]);
  return ((const float*)p)[i];
}

// sin/cos of an angle in radians on the transcendental unit (v_sin_f32 / v_cos_f32 take revolutions).
__device__ __forceinline__ void fast_sincos(float theta, float& s, float& c) {
  const float rev = theta * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
}

// ---------------------------------------------------------------------------------
// One wavefront applies all `krot` Givens stages to ONE 128-channel span held in
// wave-private LDS.  Lane l owns pair l of every stage: (i, j) = the l-th int16 pair
// of the span, theta = the l-th angle.  Restates rotation.cu:36-39 +
// rotation.cuh:53-56/:143-153 with fp32 state and no per-stage rounding.
//
// LDS layout: xr[(chunk * 128 + channel) * VW + v], rows = NCH * VW, row = chunk * VW + v.
// A span is 128 channels = one GS=128 group or two GS=64 groups (lanes 32..63 then
// index the upper 64 channels: `sub` = 64 for those lanes).
//
// No workgroup barrier is needed: the span is private to this wave and a wave's DS
// operations execute in order; the wavefront fence stops the compiler reordering.
// ---------------------------------------------------------------------------------
template <int VW, int NCH>
__device__ __forceinline__ void rotate_stage(float* xr, unsigned ij, float th, int sub) {
  typedef float V __attribute__((ext_vector_type(VW)));
  const int i = (int)(ij & 0xffffu) + sub;
  const int j = (int)(ij >> 16) + sub;
  float s, c;
  fast_sincos(th, s, c);
  V a[NCH], b[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    a[ch] = *(const V*)(xr + (ch * 128 + i) * VW);
    b[ch] = *(const V*)(xr + (ch * 128 + j) * VW);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    *(V*)(xr + (ch * 128 + i) * VW) = a[ch] * c + b[ch] * s;
    *(V*)(xr + (ch * 128 + j) * VW) = b[ch] * c - a[ch] * s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Runtime-krot form: the next stage's coefficients are fetched under the current stage's LDS work.
template <int VW, int NCH>
__device__ __forceinline__ void rotate_span_lds(float* xr, const int16_t* __restrict__ idx_span,
                                                int64_t idx_stride, const void* __restrict__ theta,
                                                int64_t theta_off, int64_t theta_stride, int theta_dt,
                                                int krot, int lane, int sub) {
  unsigned ij = *(const unsigned*)(idx_span + 2 * lane);
  float th = load_param(theta, theta_off + lane, theta_dt);
  for (int r = 0; r < krot; ++r) {
    unsigned ij_next = 0;
    float th_next = 0.f;
    if (r + 1 < krot) {
      ij_next = *(const unsigned*)(idx_span + (int64_t)(r + 1) * idx_stride + 2 * lane);
      th_next = load_param(theta, theta_off + (int64_t)(r + 1) * theta_stride + lane, theta_dt);
    }
    rotate_stage<VW, NCH>(xr, ij, th, sub);
    ij = ij_next;
    th = th_next;
  }
}

// Compile-time-krot form: all coefficients are already in registers (loaded BEFORE the weight
// tiles were requested, so waiting for them does not wait for the tiles: vmcnt retires in order).
template <int VW, int NCH, int KROT>
__device__ __forceinline__ void rotate_span_regs(float* xr, const unsigned (&ij)[KROT], const float (&th)[KROT],
                                                 int sub) {
#pragma unroll
  for (int r = 0; r < KROT; ++r) rotate_stage<VW, NCH>(xr, ij[r], th[r], sub);
}

// The merge of a split attention launch's four slots (attn.hip: paro_attn_decode_split): un-normalised outputs o, maxima m, sums l.
// ONE definition for the fused GEMV's attn_in prologue (gemv_impl.hpp) and paro_attn_finish (attn.hip): both give the same bits.
__device__ __forceinline__ float attn_merge(const f32x4& o, const f32x4& m, const f32x4& l) {
  constexpr float kLog2e = 1.4426950408889634f;
  const float M = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float w = __builtin_amdgcn_exp2f((m[c] - M) * kLog2e);
    den = __builtin_fmaf(w, l[c], den);
    num = __builtin_fmaf(w, l[c] > 0.f ? o[c] : 0.f, num);     // a slot nobody filled may hold anything
  }
  return num * __builtin_amdgcn_rcpf(den);
}

}  // namespace paro

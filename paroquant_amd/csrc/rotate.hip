// Standalone pairwise-rotation kernel for gfx950:  out = (prod_r Givens_r)(x * scales).
//
// MI355X-native restatement of the operator behind torch.ops.rotation.rotate
// (reference: paroquant/kernels/cuda/rotation.cu:10-43,62-124; rotation.cuh:16-173).
// Design differences, on purpose:
//   * one 64-lane wavefront owns one 128-channel span x R rows: 64 lanes == 64 pairs, so
//     the krot stages need no workgroup barrier at all (the reference issues 8
//     __syncthreads per block) -- only in-order DS traffic inside the wave;
//   * the state stays fp32 in LDS for all stages and is rounded once on the way out
//     (the reference re-rounds to half after every stage, rotation.cuh:152-153);
//   * theta / scales are consumed in their own dtype (no per-call cast kernels,
//     cf. rotation.cu:75-78).
#include "common.hpp"

namespace paro {

template <int VW, int NCH>
__global__ __launch_bounds__(64) void rotate_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                   const int16_t* __restrict__ idx,
                                                   const void* __restrict__ theta,
                                                   const void* __restrict__ scales, int rows, int hidden,
                                                   int krot, int gs, int x_dt, int p_dt) {
  // blockIdx.z = merged partition: the same x is rotated with partition z's parameters into out[z]
  // (parameters are [P, krot, hidden] / [P, krot, hidden/2] / [P, hidden]; 2-byte parameter types only)
  if (blockIdx.z > 0) {
    const int64_t pz = blockIdx.z;
    idx += pz * krot * hidden;
    theta = (const unsigned short*)theta + pz * krot * (hidden / 2);
    if (scales) scales = (const unsigned short*)scales + pz * hidden;
    out = (char*)out + pz * rows * hidden * (x_dt == PARO_DTYPE_F32 ? 4 : 2);
  }
  constexpr int R = VW * NCH;
  __shared__ __attribute__((aligned(16))) float xr[R * 128];
  const int lane = threadIdx.x;
  const int span = blockIdx.y;
  const int row0 = blockIdx.x * R;
  const int64_t c0 = (int64_t)span * 128 + 2 * lane;  // this lane's two channels (load / store side)
  if (c0 >= hidden) return;                           // gs == 64 with an odd number of groups
  const int sub = (gs == 64) ? (lane >> 5) * 64 : 0;

  float s0 = 1.f, s1 = 1.f;
  if (scales) {
    s0 = load_param(scales, c0, p_dt);
    s1 = load_param(scales, c0 + 1, p_dt);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    float v0 = 0.f, v1 = 0.f;
    if (row < rows) {
      const int64_t o = (int64_t)row * hidden + c0;
      if (x_dt == PARO_DTYPE_F32) {
        const f32x2 t = *(const f32x2*)((const float*)x + o);
        v0 = t.x;
        v1 = t.y;
      } else {
        const unsigned t = *(const unsigned*)((const unsigned short*)x + o);
        if (x_dt == PARO_DTYPE_F16) {
          v0 = f16_bits_to_f32(t & 0xffffu);
          v1 = f16_bits_to_f32(t >> 16);
        } else {
          v0 = bf16_bits_to_f32(t & 0xffffu);
          v1 = bf16_bits_to_f32(t >> 16);
        }
      }
    }
    const int ch = r / VW, v = r % VW;
    xr[(ch * 128 + 2 * lane) * VW + v] = v0 * s0;
    xr[(ch * 128 + 2 * lane + 1) * VW + v] = v1 * s1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  rotate_span_lds<VW, NCH>(xr, idx + (int64_t)span * 128, hidden, theta, (int64_t)span * 64, hidden / 2, p_dt,
                           krot, lane, sub);

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    if (row >= rows) break;
    const int ch = r / VW, v = r % VW;
    const float v0 = xr[(ch * 128 + 2 * lane) * VW + v];
    const float v1 = xr[(ch * 128 + 2 * lane + 1) * VW + v];
    const int64_t o = (int64_t)row * hidden + c0;
    if (x_dt == PARO_DTYPE_F32) {
      f32x2 t;
      t.x = v0;
      t.y = v1;
      *(f32x2*)((float*)out + o) = t;
    } else if (x_dt == PARO_DTYPE_F16) {
      *(unsigned*)((unsigned short*)out + o) = (unsigned)f32_to_f16_bits(v0) | ((unsigned)f32_to_f16_bits(v1) << 16);
    } else {
      *(unsigned*)((unsigned short*)out + o) = (unsigned)f32_to_bf16_bits(v0) | ((unsigned)f32_to_bf16_bits(v1) << 16);
    }
  }
}

int launch_rotate(const void* x, void* out, const int16_t* idx, const void* theta, const void* scales,
                  int64_t rows, int64_t hidden, int krot, int gs, int x_dt, int p_dt, hipStream_t st,
                  int nparts) {
  if (rows == 0) return PARO_OK;
  const unsigned spans = (unsigned)((hidden + 127) / 128);
  const unsigned P = (unsigned)(nparts < 1 ? 1 : nparts);
  if (rows <= 1) {
    hipLaunchKernelGGL((rotate_kernel<1, 1>), dim3(1, spans, P), dim3(64), 0, st, x, out, idx, theta, scales,
                       (int)rows, (int)hidden, krot, gs, x_dt, p_dt);
  } else if (rows <= 4096) {
    hipLaunchKernelGGL((rotate_kernel<4, 1>), dim3((unsigned)((rows + 3) / 4), spans, P), dim3(64), 0, st, x, out, idx,
                       theta, scales, (int)rows, (int)hidden, krot, gs, x_dt, p_dt);
  } else {
    hipLaunchKernelGGL((rotate_kernel<4, 2>), dim3((unsigned)((rows + 7) / 8), spans, P), dim3(64), 0, st, x, out, idx,
                       theta, scales, (int)rows, (int)hidden, krot, gs, x_dt, p_dt);
  }
  return check_launch("paro_rotate");
}

}  // namespace paro

extern "C" int paro_rotate(const void* x, void* out, const int16_t* idx_ij, const void* theta, const void* scales,
                           int64_t rows, int64_t hidden, int krot, int group_size, int x_dtype, int param_dtype,
                           void* stream) {
  using namespace paro;
  // validation order follows rotate_dynamic / rotate_launcher (rotation.cu:111-124, :62-66)
  if (group_size != 64 && group_size != 128)
    return fail(PARO_ERR_UNSUPPORTED, "Unsupported group_size: %d; expected 64 or 128", group_size);
  if (krot < 1 || krot > 16) return fail(PARO_ERR_UNSUPPORTED, "Unsupported KROT = %d; supported: 1..16", krot);
  if (hidden <= 0 || hidden % group_size != 0) return fail(PARO_ERR_INVALID, "h must be divisible by GROUP_SIZE");
  if (rows < 0 || rows > 0x7fffffff || hidden > 0x7fffffff) return fail(PARO_ERR_INVALID, "rows/hidden out of range");
  if (x_dtype < 0 || x_dtype > 2 || param_dtype < 0 || param_dtype > 2)
    return fail(PARO_ERR_INVALID, "rotate supports Float, Half, and BFloat16");
  if (!x || !out || !idx_ij || !theta) return fail(PARO_ERR_INVALID, "null pointer");
  return launch_rotate(x, out, idx_ij, theta, scales, rows, hidden, krot, group_size, x_dtype, param_dtype,
                       (hipStream_t)stream, 1);
}

// ---------------------------------------------------------------------------------------------
// Pre-pass for 9..255 rows on the PACKED exchange schedule (paro_pack_rotation: 3 KiB per group, the words the fused GEMV
// consumes).  The stage kernel above derives cos / sin from theta per stage and keeps its state in LDS: 6.5 .. 7 us per call
// at 17..32 rows (profiles/r06_rows_17_64.jsonl), a launch boundary plus ~4 us of its own.  Here one WAVE rotates one
// (partition, group, 4 rows) task in REGISTERS -- the seed / stage / finish arithmetic of the GEMV's in-kernel rotation
// (gemv_impl.hpp, the producer of mode 3), so the rotated halves are BIT-IDENTICAL to what modes 0 and 3 multiply by -- and
// stores either plain rows out[p][row][K] (the MFMA GEMM's LDS-DMA source, mode 2's layout) or, for the GEMV on 17..64 rows,
// MFMA-FRAGMENT order: [p][g][row tile][k-step][lane = 16 mq + (row & 15)] x 16 bytes, so that a wave's A operand of one
// (row tile, k-step) is ONE contiguous 1-KiB load instead of sixteen 64-byte row pieces (every column block re-reads the
// rows: Qwen3-4B qkv at 32 rows pulls 30 MB of x through L2 beside 8.4 MB of weights).
// Same semantics as rotate<T,4,128,8> per partition (rotation.cu:10-43; plugin.py:288-306 for the merged partitions).
// ---------------------------------------------------------------------------------------------
namespace paro {

constexpr int kPrerotRows = 4;       // rows per task
constexpr int kPrerotStride = 136;   // halves per staged row (128 + 8 pad)

template <typename AT, bool FRAG>
__global__ __launch_bounds__(256) void prerot_sched_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                                          const unsigned* __restrict__ rot, const unsigned short* __restrict__ cs,
                                                          int rows, int K, int krot, int units, int row_tiles) {
  typedef Act<AT> A;
  constexpr int PRR = kPrerotRows;
  __shared__ __attribute__((aligned(16))) unsigned short lds[4 * PRR * kPrerotStride];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = blockIdx.x * 4 + wave;
  if (u >= units) return;
  const int G = K >> 7;
  const int nq = (rows + PRR - 1) / PRR;
  const int pg = u / nq, r0 = (u - pg * nq) * PRR;     // the four waves of a workgroup share (partition, group) when nq >= 4
  const int pu = pg / G, g = pg - pu * G;
  unsigned short* xq = lds + wave * (PRR * kPrerotStride);
  const u32x4* rp = (const u32x4*)rot + (int64_t)pg * 192 + lane;
  u32x4 rc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) rc[q] = rp[q * 64];
  const unsigned csv = *(const unsigned*)(cs + (int64_t)pu * K + g * 128 + 2 * lane);
  unsigned xv[PRR];
#pragma unroll
  for (int r = 0; r < PRR; ++r) {
    const int rr = min(r0 + r, rows - 1);
    xv[r] = *(const unsigned*)(x + (int64_t)rr * K + g * 128 + 2 * lane);
  }
  const float fscale = __builtin_ldexpf(1.0f, 49 - 14 * krot);
  const float c0 = f16_bits_to_f32(csv & 0xffffu) * 0x1p-63f, c1 = f16_bits_to_f32(csv >> 16) * 0x1p-63f;
  float sa[PRR], sb[PRR];
#pragma unroll
  for (int r = 0; r < PRR; ++r) {
    const unsigned v = (r0 + r < rows) ? xv[r] : 0u;
    sa[r] = A::to_f32(v & 0xffffu) * c0;
    sb[r] = A::to_f32(v >> 16) * c1;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < krot) {
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
      *(unsigned short*)((unsigned char*)(xq + r * kPrerotStride) + oa) = A::from_f32(o1);
      *(unsigned short*)((unsigned char*)(xq + r * kPrerotStride) + ob) = A::from_f32(o2);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if constexpr (FRAG) {
    // lane -> (row r0 + (lane & 3), 16-byte chunk c = lane >> 2 = 4 i + mq): four lanes write 64 contiguous bytes
    const int r = lane & 3, c = lane >> 2;
    const int row = r0 + r;
    if (row < rows) {
      const u32x4 v = *(const u32x4*)(xq + r * kPrerotStride + 8 * c);
      const int64_t frag = ((int64_t)pg * row_tiles + (row >> 4)) * 4 + (c >> 2);
      *(u32x4*)(out + (frag * 64 + (c & 3) * 16 + (row & 15)) * 8) = v;
    }
  } else {
    // lane -> (row r0 + (lane >> 4), chunk lane & 15): 256 contiguous bytes per row
    const int r = lane >> 4, c = lane & 15;
    const int row = r0 + r;
    if (row < rows) {
      const u32x4 v = *(const u32x4*)(xq + r * kPrerotStride + 8 * c);
      *(u32x4*)(out + ((int64_t)pu * rows + row) * K + g * 128 + 8 * c) = v;
    }
  }
}

// frag_row_tiles = 0: plain rows out[p][row][K]; > 0: fragment order for a GEMV instantiation of that many 16-row tiles
int launch_prerot_sched(const void* x, void* out, const void* rot, const void* cs, int64_t rows, int64_t K, int krot, int nparts,
                        int dt, int frag_row_tiles, hipStream_t st) {
  if (rows == 0) return PARO_OK;
  if (krot < 1 || krot > 8 || !rot || !cs) return fail(PARO_ERR_INVALID, "schedule pre-pass: needs the packed schedule (krot <= 8)");
  const int64_t units = (int64_t)nparts * (K / 128) * ((rows + kPrerotRows - 1) / kPrerotRows);
  if (units > 0x7fffffff / 4) return fail(PARO_ERR_UNSUPPORTED, "schedule pre-pass: too many tasks");
  const dim3 grid((unsigned)((units + 3) / 4));
  const bool h = dt == PARO_DTYPE_F16;
#define PARO_PREROT_LAUNCH(AT, FR)                                                                                                    \
  hipLaunchKernelGGL((prerot_sched_kernel<AT, FR>), grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)out,           \
                     (const unsigned*)rot, (const unsigned short*)cs, (int)rows, (int)K, krot, (int)units, frag_row_tiles)
  if (frag_row_tiles > 0) { if (h) PARO_PREROT_LAUNCH(f16, true); else PARO_PREROT_LAUNCH(bf16, true); }
  else { if (h) PARO_PREROT_LAUNCH(f16, false); else PARO_PREROT_LAUNCH(bf16, false); }
#undef PARO_PREROT_LAUNCH
  return check_launch("paro_rotate (schedule pre-pass)");
}

}  // namespace paro

// ---------------------------------------------------------------------------------------------
// Prefill pre-pass on the matrix cores.  For many rows the 8 sparse Givens stages are cheaper as ONE
// dense product per 128-channel group:  x_rot[:, g] = x[:, g] @ R'_g,  R'_g = diag(cs_g) * G_1 ... G_8
// (128 x 128, built once at load time by running the stage kernel above on the scaled identity and
// stored transposed/k-contiguous as rmat[p][g][n][k] in the activation dtype).  The stage kernel is
// LDS-write bound (~2.3 TB/s of input); this one is HBM bound: 2*M*K*2 bytes per partition.
//   grid (ceil(rows / 256), K / 128, P), 256 threads = 4 waves x 64 rows.
//   X tile: global -> LDS by global_load_lds with the source-side XOR swizzle (as in gemm.hip);
//   every wave pulls its 64 x 128 A fragments into registers, then walks the 8 column tiles of R'_g
//   (B fragments straight from L2) and writes the fp16/bf16 result back through the same LDS tile
//   so that the global stores are 16-byte, full-row coalesced.
// ---------------------------------------------------------------------------------------------
namespace paro {

template <typename AT>
__global__ __launch_bounds__(256) void rotate_mfma_kernel(const unsigned short* __restrict__ x,
                                                         unsigned short* __restrict__ out,
                                                         const unsigned short* __restrict__ rmat, int rows, int K,
                                                         int nparts) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  // What bounds this kernel is what a CU can take in through its vector-memory path (10..17 B / clock in this code base), not HBM:
  // with every wave fetching its own copy of the group's 128 x 128 rotation matrix as B fragments, a workgroup pulled 128 KB of
  // matrix per 64 KB of output (T = 80 + 200 P us at M = 65536; twice the waves per CU with half the rows each: 1.5x SLOWER).
  // Now the matrix of (partition, group) comes into LDS ONCE per workgroup (LDS-DMA, 32 KB, 16-byte chunks XOR-swizzled by the row
  // so that the 16 rows of a B fragment read conflict-free) and the four waves read their fragments from there.
  // LDS: [0, 64 KB) the x tile (256 rows x 256 B) at first; then [0, 36 KB) output staging (per wave 64 rows x 64 columns, pitch
  // 144 B), [36 KB, 68 KB) the rotation matrix.
  constexpr int OP = 144;                       // staging pitch: rows 4 mq + r of a D fragment fall into different banks
  constexpr int STG = 64 * OP;                  // bytes of staging per wave
  constexpr int RM0 = 4 * STG;                  // 36864: rotation matrix tile
  __shared__ __attribute__((aligned(16))) unsigned char lds[RM0 + 128 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x = group: the workgroups that run together cover whole rows of x (and of every rotated copy) -- consecutive DRAM
  // pages -- instead of one 256-byte column stripe of 256 x (resident workgroups) different rows (pre-pass 757 -> 578 us at
  // M = 65536, Llama-3-8B qkv + gate_up, profiles/r03_prepass_order.txt)
  const int row0 = blockIdx.y * 256, g = blockIdx.x;
  const int G = K >> 7;
  const int n = lane & 15, mq = lane >> 4;

  // stage X[row0 .. row0+255][g*128 .. +127]: wave w fills rows 64w .. 64w+63 (its own rows)
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int row = wave * 64 + c * 4 + (lane >> 4);
    const int grow = min(row0 + row, rows - 1);
    const unsigned short* src = x + (int64_t)grow * K + g * 128 + (((lane & 15) ^ (row & 15)) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + (wave * 64 + c * 4) * 256), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (a wave reads only the rows it staged itself)
  __builtin_amdgcn_wave_barrier();
  // A fragments of this wave's 64 rows (4 row tiles x 4 k-steps)
  vec8 af[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int row = wave * 64 + rt * 16 + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[rt][i] = *(const vec8*)(lds + row * 256 + (((4 * i + mq) ^ (row & 15)) << 4));
  }
  __syncthreads();  // every wave has its rows in registers: the tile becomes staging + matrix space

  // matrix of (p, g): rmat[p][g][n][k] (k contiguous, 256 B per output column n) -> LDS row n, chunk (k / 8) ^ (n & 15)
  auto issue_rmat = [&](int p) {
    const unsigned short* rp = rmat + (((int64_t)p * G + g) * 128) * 128;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int row = (wave * 8 + c) * 4 + (lane >> 4);        // 4 rows of 16 chunks per wave instruction
      const unsigned short* src = rp + row * 128 + (((lane & 15) ^ (row & 15)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + RM0 + (wave * 8 + c) * 1024), 16, 0, 0);
    }
  };
  issue_rmat(0);
  unsigned char* stg = lds + wave * STG;
  for (int p = 0; p < nparts; ++p) {
    // (the only requests younger than the matrix pieces are none at p = 0 and the previous partition's last 8 row stores after:
    // vmcnt retires in order, so waiting for the pieces does not wait for those stores)
    // (a partial last tile may skip store instructions whose rows are all out of range: count nothing there)
    if (p == 0 || row0 + 256 > rows) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();                                      // the whole matrix has landed
    unsigned short* op = out + (int64_t)p * rows * K;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {                      // 64 output columns at a time through the staging buffer
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int ct = hh * 4 + c4;
        vec8 bf[4];
        const int brow = ct * 16 + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const vec8*)(lds + RM0 + brow * 256 + (((4 * i + mq) ^ (brow & 15)) << 4));
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i) d = A::mfma(af[rt][i], bf[i], d);
          // D: row = 4*mq + r, col = n  ->  staging[row][c4*16 + n]
#pragma unroll
          for (int r = 0; r < 4; ++r) *(unsigned short*)(stg + (rt * 16 + 4 * mq + r) * OP + (c4 * 16 + n) * 2) = A::from_f32(d[r]);
        }
      }
      if (hh == 1) {
        // every wave is done with this partition's matrix: the next one is requested BEFORE the stores below
        __syncthreads();
        if (p + 1 < nparts) issue_rmat(p + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int row = (lane >> 3) + 8 * c;                 // the wave's own rows: 128 contiguous bytes per 8 lanes
        if (row0 + wave * 64 + row < rows)
          *(u32x4*)(op + (int64_t)(row0 + wave * 64 + row) * K + g * 128 + hh * 64 + (lane & 7) * 8) = *(const u32x4*)(stg + row * OP + (lane & 7) * 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staged rows are in registers before the next half overwrites them
      __builtin_amdgcn_wave_barrier();
    }
  }
}

int launch_rotate_mfma(const void* x, void* out, const void* rmat, int64_t rows, int64_t K, int nparts, int dt,
                       hipStream_t st) {
  if (rows == 0) return PARO_OK;
  const int64_t rb = (rows + 255) / 256;
  if (rb > 65535) return fail(PARO_ERR_UNSUPPORTED, "rotate pre-pass: more than 65535 x 256 rows");
  dim3 grid((unsigned)(K / 128), (unsigned)rb);
  if (dt == PARO_DTYPE_F16)
    hipLaunchKernelGGL(rotate_mfma_kernel<f16>, grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)out,
                       (const unsigned short*)rmat, (int)rows, (int)K, nparts);
  else
    hipLaunchKernelGGL(rotate_mfma_kernel<bf16>, grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)out,
                       (const unsigned short*)rmat, (int)rows, (int)K, nparts);
  return check_launch("paro_rotate (mfma pre-pass)");
}

}  // namespace paro

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

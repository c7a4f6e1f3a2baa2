// One-time weight repack: AWQ checkpoint layout -> CDNA4 tile layout (see include/paro_abi.h).
//
// Takes the place of the per-partition AWQ -> Marlin conversion the reference performs in
// ParoQuantLinearMethod.process_weights_after_loading (vllm/plugin.py:208-279).  The source
// format is the one written by paroquant/cli/convert.py:149-155,194-203.
#include "common.hpp"

namespace paro {

// AWQ nibble p of a word holds column 8c + (0,2,4,6,1,3,5,7)[p]; the inverse map (column -> nibble)
// is (0,4,1,5,2,6,3,7)  (cli/convert.py:19; mlx/load.py:18).
__device__ __forceinline__ unsigned awq_nibble(unsigned word, int col_in_word) {
  const int p = ((col_in_word & 1) << 2) | (col_in_word >> 1);
  return (word >> (4 * p)) & 0xFu;
}

// one thread per output word: out[((t * G + g) * 64 + lane) * 4 + i]
__global__ __launch_bounds__(256) void repack_qweight_kernel(const unsigned* __restrict__ qw,
                                                            unsigned* __restrict__ out, int K, int N) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int G = K / 128;
  const int64_t total = (int64_t)(N / 16) * G * 256;
  if (gid >= total) return;
  const int i = (int)(gid & 3);
  const int lane = (int)((gid >> 2) & 63);
  const int64_t tg = gid >> 8;
  const int g = (int)(tg % G);
  const int t = (int)(tg / G);
  const int n = lane & 15, kb = lane >> 4;
  const int col = t * 16 + n;
  const int k0 = g * 128 + i * 32 + kb * 8;
  const int wcol = col >> 3, cin = col & 7;
  const int NW = N / 8;
  unsigned o = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned q = awq_nibble(qw[(int64_t)(k0 + e) * NW + wcol], cin);
    o |= q << (4 * ((e >> 1) + 4 * (e & 1)));
  }
  out[gid] = o;
}

__global__ __launch_bounds__(256) void repack_qzeros_kernel(const unsigned* __restrict__ qz,
                                                           unsigned* __restrict__ out, int64_t words) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= words) return;
  const unsigned w = qz[gid];
  unsigned o = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) o |= awq_nibble(w, j) << (4 * j);
  out[gid] = o;
}

// Debug / verification: dense W[k, n] = (q - z) * s from the PACKED buffers, rounded once.
template <typename AT>
__global__ __launch_bounds__(256) void dequant_packed_kernel(const unsigned* __restrict__ wq,
                                                            const unsigned* __restrict__ zq,
                                                            const unsigned short* __restrict__ scales,
                                                            unsigned short* __restrict__ out, int K, int N) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)K * N) return;
  const int k = (int)(gid / N), n = (int)(gid % N);
  const int G = K / 128;
  const int g = k >> 7, kk = k & 127;
  const int i = kk >> 5, kb = (kk >> 3) & 3, e = kk & 7;
  const int t = n >> 4, lane = (kb << 4) | (n & 15);
  const unsigned w = wq[(((int64_t)t * G + g) * 64 + lane) * 4 + i];
  const int q = (int)((w >> (4 * ((e >> 1) + 4 * (e & 1)))) & 0xF);
  const int z = (int)((zq[(int64_t)g * (N / 8) + (n >> 3)] >> (4 * (n & 7))) & 0xF);
  const float s = f16_bits_to_f32(scales[(int64_t)g * N + n]);
  out[gid] = Act<AT>::from_f32((float)(q - z) * s);
}

}  // namespace paro

extern "C" int64_t paro_packed_qweight_bytes(int64_t K, int64_t N) {
  if (K <= 0 || N <= 0 || K % 128 != 0 || N % 16 != 0) return -1;
  return K * N / 2;
}

extern "C" int64_t paro_packed_qzeros_bytes(int64_t K, int64_t N) {
  if (K <= 0 || N <= 0 || K % 128 != 0 || N % 16 != 0) return -1;
  return (K / 128) * (N / 8) * 4;
}

extern "C" int paro_repack_awq(const int32_t* qweight, const int32_t* qzeros, int64_t K, int64_t N, void* out_wq,
                               void* out_zq, void* stream) {
  using namespace paro;
  if (K <= 0 || N <= 0 || K % 128 != 0) return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128 (got %lld)", (long long)K);
  if (N % 16 != 0) return fail(PARO_ERR_INVALID, "out_features must be a multiple of 16 (got %lld)", (long long)N);
  if (K > 0x7fffffff || N > 0x7fffffff) return fail(PARO_ERR_INVALID, "shape out of range");
  if (!qweight || !qzeros || !out_wq || !out_zq) return fail(PARO_ERR_INVALID, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t words = K * N / 8;
  hipLaunchKernelGGL(repack_qweight_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                     (const unsigned*)qweight, (unsigned*)out_wq, (int)K, (int)N);
  const int64_t zwords = (K / 128) * (N / 8);
  hipLaunchKernelGGL(repack_qzeros_kernel, dim3((unsigned)((zwords + 255) / 256)), dim3(256), 0, st,
                     (const unsigned*)qzeros, (unsigned*)out_zq, zwords);
  return check_launch("paro_repack_awq");
}

extern "C" int paro_dequant_packed(const paro_linear_t* L, void* out_w, void* stream) {
  using namespace paro;
  if (!L || !out_w || !L->wq || !L->zq || !L->scales) return fail(PARO_ERR_INVALID, "null pointer");
  if (L->K % 128 != 0 || L->N % 16 != 0) return fail(PARO_ERR_INVALID, "bad shape");
  const int64_t total = L->K * L->N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (L->act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(dequant_packed_kernel<f16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->zq, (const unsigned short*)L->scales, (unsigned short*)out_w, (int)L->K,
                       (int)L->N);
  else if (L->act_dtype == PARO_DTYPE_BF16)
    hipLaunchKernelGGL(dequant_packed_kernel<bf16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->zq, (const unsigned short*)L->scales, (unsigned short*)out_w, (int)L->K,
                       (int)L->N);
  else
    return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_dequant_packed");
}

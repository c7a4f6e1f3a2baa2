// One-time repack: checkpoint layout -> CDNA4 kernel layout (see include/paro_abi.h).
//
// Takes the place of the per-partition AWQ -> Marlin conversion the reference performs in
// ParoQuantLinearMethod.process_weights_after_loading (vllm/plugin.py:208-279).  The source
// format is the one written by paroquant/cli/convert.py:149-155,194-203,264-277.
#include "common.hpp"

namespace paro {

// AWQ nibble p of a word holds column 8c + (0,2,4,6,1,3,5,7)[p]; the inverse map (column -> nibble)
// is (0,4,1,5,2,6,3,7)  (cli/convert.py:19; mlx/load.py:18).
__device__ __forceinline__ unsigned awq_nibble(unsigned word, int col_in_word) {
  const int p = ((col_in_word & 1) << 2) | (col_in_word >> 1);
  return (word >> (4 * p)) & 0xFu;
}

// one thread per output word: out[(chunk(t, g) * 64 + lane) * 4 + i], chunk = t*G+g (order 0) or g*T+t (order 1)
__global__ __launch_bounds__(256) void repack_qweight_kernel(const unsigned* __restrict__ qw,
                                                            unsigned* __restrict__ out, int K, int N, int order) {
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
  const int64_t chunk = order ? (int64_t)g * (N / 16) + t : tg;
  out[(chunk << 8) | (gid & 255)] = o;
}

// one thread per (group, real column): scale + zero point -> one word in the padded tile space
__global__ __launch_bounds__(256) void pack_sz_kernel(const unsigned* __restrict__ qz,
                                                     const unsigned short* __restrict__ scales,
                                                     unsigned* __restrict__ out, int G, int N, PartTable pt) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)G * N) return;
  const int g = (int)(gid / N), col = (int)(gid % N);
  const int t = col >> 4, n = col & 15;
  const int p = pt.part_of_tile(t);
  const int ts = pt.szt_start[p] + (t - pt.tile_start[p]);
  const unsigned z = awq_nibble(qz[(int64_t)g * (N / 8) + (col >> 3)], col & 7);
  const unsigned zf = f32_to_f16_bits((float)z);
  const unsigned s = scales[(int64_t)g * N + col];
  out[(((int64_t)g * (pt.tsz / 4) + (ts >> 2)) * 16 + n) * 4 + (ts & 3)] = s | (zf << 16);
}

// One rotation coefficient word: the fp32 value NEAREST to `v` whose low 9 bits equal `4 * channel`
// (the LDS byte offset of the channel's fp32 state).  The GEMV uses the word as the float directly
// (relative error <= 2^-15, below the single final rounding to fp16 / bf16) and masks 0x1fc for the address.
__device__ __forceinline__ unsigned coef_word(float v, unsigned channel) {
  const unsigned lo = (channel & 127u) * 4u;
  const unsigned b = (__builtin_bit_cast(unsigned, v) & ~0x1ffu) | lo;
  unsigned best = b;
  float err = fabsf(__builtin_bit_cast(float, b) - v);
  if ((b & 0x7fffffffu) >= 0x200u) {
    const float e = fabsf(__builtin_bit_cast(float, b - 0x200u) - v);
    if (e < err) { err = e; best = b - 0x200u; }
  }
  {
    const float e = fabsf(__builtin_bit_cast(float, b + 0x200u) - v);
    if (e < err) { err = e; best = b + 0x200u; }
  }
  return best;
}

// One thread per (partition, group, stage): pack that stage's 64 Givens pairs into lane words
//   i | j << 8 | theta_fp16 << 16
// and, because WHICH lane applies a pair and the pair's orientation are free
// ((i, j, theta) == (j, i, -theta): xi' = c xi + s xj, xj' = c xj - s xi), choose both so that the
// stage's LDS traffic is bank-conflict free: ds_read/write_b32 is serviced per 32-lane half over 32
// banks (bank = channel mod 32), so within each half all `i` channels must differ mod 32 and all `j`
// channels must differ mod 32.  Construction: the 64 pairs are the edges of a 4-regular multigraph
// on the 32 bank classes; an Euler orientation gives every class out-degree 2 (as `i`) and in-degree
// 2 (as `j`); the resulting 2-regular bipartite graph (i-classes x j-classes) splits into two perfect
// matchings by alternating along its cycles -- one matching per half-wave.  Always succeeds when the
// stage is a perfect matching of the 128 channels (every valid checkpoint: optim/rotation.py:37-54);
// otherwise the input order is kept (still correct, just not conflict free).
__global__ __launch_bounds__(64) void pack_rot_kernel(const int16_t* __restrict__ pairs,
                                                     const unsigned short* __restrict__ theta,
                                                     unsigned* __restrict__ out, int K, int nparts, int krot) {
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int G = K / 128;
  if (gid >= (int64_t)nparts * G * 8) return;
  const int r = (int)(gid & 7);
  const int g = (int)((gid >> 3) % G);
  const int p = (int)((gid >> 3) / G);
  // [p][g][stage pair q = r >> 1][lane][4 words]: words 2 * (r & 1) + {0, 1} of the lane's 16-byte chunk
  unsigned* o = out + (((int64_t)p * G + g) * 4 + (r >> 1)) * 256 + 2 * (r & 1);  // + lane * 4
  if (r >= krot) {
    for (int l = 0; l < 64; ++l) {  // identity stage (never executed: the kernels stop at krot)
      o[l * 4] = coef_word(1.0f, 2 * l);
      o[l * 4 + 1] = coef_word(0.0f, 2 * l + 1);
    }
    return;
  }
  const int64_t pb = (int64_t)p * krot + r;
  const int16_t* pr = pairs + pb * K + g * 128;
  const unsigned short* th = theta + pb * (K / 2) + g * 64;

  unsigned char ei[64], ej[64], flip[64], half[64];
  signed char adj[32][4];
  unsigned char deg[32], seen[128];
  bool valid = true;
  for (int c = 0; c < 32; ++c) deg[c] = 0;
  for (int c = 0; c < 128; ++c) seen[c] = 0;
  for (int e = 0; e < 64; ++e) {
    const int i = pr[2 * e], j = pr[2 * e + 1];
    if (i < 0 || i > 127 || j < 0 || j > 127 || i == j || seen[i & 127] || seen[j & 127]) {
      valid = false;
      break;
    }
    seen[i] = seen[j] = 1;
    ei[e] = (unsigned char)i;
    ej[e] = (unsigned char)j;
    flip[e] = 0;
    half[e] = (unsigned char)(e >> 5);
    adj[i & 31][deg[i & 31]++] = (signed char)e;
    adj[j & 31][deg[j & 31]++] = (signed char)e;  // a loop (i == j mod 32) appears twice in its class
  }
  if (valid) {
    // --- Euler orientation: walk closed trails, orienting every edge away from the vertex it is left by
    unsigned char used[64], tail[64], head[64];
    for (int e = 0; e < 64; ++e) used[e] = 0;
    for (int start = 0; start < 32; ++start) {
      int v = start;
      for (;;) {
        int e = -1;
        for (int s = 0; s < 4; ++s)
          if (!used[adj[v][s]]) {
            e = adj[v][s];
            break;
          }
        if (e < 0) break;
        used[e] = 1;
        const int ci = ei[e] & 31, cj = ej[e] & 31;
        int w;
        if (ci == v) {
          flip[e] = 0;
          w = cj;
        } else {
          flip[e] = 1;
          w = ci;
        }
        tail[e] = (unsigned char)v;
        head[e] = (unsigned char)w;
        v = w;
      }
    }
    // --- 2-colour the 2-regular bipartite graph (tails x heads) by alternating along its cycles
    signed char outE[32][2], inE[32][2];
    unsigned char no[32], ni[32];
    for (int c = 0; c < 32; ++c) no[c] = ni[c] = 0;
    for (int e = 0; e < 64; ++e) {
      if (no[tail[e]] >= 2 || ni[head[e]] >= 2) {
        valid = false;
        break;
      }
      outE[tail[e]][no[tail[e]]++] = (signed char)e;
      inE[head[e]][ni[head[e]]++] = (signed char)e;
    }
    if (valid) {
      unsigned char col[64];
      for (int e = 0; e < 64; ++e) col[e] = 2;
      for (int e0 = 0; e0 < 64; ++e0) {
        if (col[e0] != 2) continue;
        int e = e0;
        for (;;) {
          col[e] = 0;
          const int hv = head[e];
          const int f = (inE[hv][0] == e) ? inE[hv][1] : inE[hv][0];  // the other edge into this head class
          if (col[f] != 2) break;
          col[f] = 1;
          const int tv = tail[f];
          const int gnext = (outE[tv][0] == f) ? outE[tv][1] : outE[tv][0];  // the other edge out of that tail class
          if (col[gnext] != 2) break;
          e = gnext;
        }
      }
      for (int e = 0; e < 64; ++e) half[e] = col[e];
    }
  }
  int next[2] = {0, 32};
  if (!valid) {
    for (int e = 0; e < 64; ++e) {
      flip[e] = 0;
      half[e] = (unsigned char)(e >> 5);
    }
  }
  for (int e = 0; e < 64; ++e) {
    const unsigned i = (unsigned short)pr[2 * e] & 0xffu, j = (unsigned short)pr[2 * e + 1] & 0xffu;
    const int lane = next[half[e]]++;
    float sn, cs;
    sincosf(f16_bits_to_f32(th[e]), &sn, &cs);   // accurate libm sincos, once, at load time
    // (i, j, theta) == (j, i, -theta): the orientation chosen above only swaps the roles and the sign of sin
    o[lane * 4] = coef_word(cs, flip[e] ? j : i);
    o[lane * 4 + 1] = coef_word(flip[e] ? -sn : sn, flip[e] ? i : j);
  }
}

// Debug / verification: dense W[k, n] = (q - z) * s from the PACKED buffers, rounded once.
template <typename AT>
__global__ __launch_bounds__(256) void dequant_packed_kernel(const unsigned* __restrict__ wq,
                                                            const unsigned* __restrict__ sz,
                                                            unsigned short* __restrict__ out, int K, int N,
                                                            PartTable pt, int order) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)K * N) return;
  const int k = (int)(gid / N), n = (int)(gid % N);
  const int G = K / 128;
  const int g = k >> 7, kk = k & 127;
  const int i = kk >> 5, kb = (kk >> 3) & 3, e = kk & 7;
  const int t = n >> 4, lane = (kb << 4) | (n & 15);
  const int64_t chunk = order ? (int64_t)g * (N / 16) + t : (int64_t)t * G + g;
  const unsigned w = wq[(chunk * 64 + lane) * 4 + i];
  const int q = (int)((w >> (4 * ((e >> 1) + 4 * (e & 1)))) & 0xF);
  const int p = pt.part_of_tile(t);
  const int ts = pt.szt_start[p] + (t - pt.tile_start[p]);
  const unsigned word = sz[(((int64_t)g * (pt.tsz / 4) + (ts >> 2)) * 16 + (n & 15)) * 4 + (ts & 3)];
  const float s = f16_bits_to_f32(word & 0xffffu);
  const float zf = f16_bits_to_f32(word >> 16);  // zero point
  out[gid] = Act<AT>::from_f32(((float)q - zf) * s);
}

}  // namespace paro

extern "C" int64_t paro_packed_qweight_bytes(int64_t K, int64_t N) {
  if (K <= 0 || N <= 0 || K % 128 != 0 || N % 16 != 0) return -1;
  return K * N / 2;
}

extern "C" int64_t paro_packed_sz_bytes(int64_t K, int n_parts, const int32_t* part_cols) {
  paro::PartTable pt;
  if (K <= 0 || K % 128 != 0 || !paro::fill_part_table(pt, n_parts, part_cols, 1)) return -1;
  return (K / 128) * (int64_t)pt.tsz * 16 * 4;
}

extern "C" int64_t paro_packed_rot_bytes(int64_t K, int n_parts) {
  if (K <= 0 || K % 128 != 0 || n_parts < 1 || n_parts > PARO_MAX_PARTS) return -1;
  return (int64_t)n_parts * (K / 128) * 64 * 8 * 8;
}

extern "C" int paro_repack_awq(const int32_t* qweight, const int32_t* qzeros, const void* scales, int64_t K, int64_t N,
                               int n_parts, const int32_t* part_cols, int wq_order, void* out_wq, void* out_sz,
                               void* stream) {
  using namespace paro;
  if (K <= 0 || N <= 0 || K % 128 != 0)
    return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128 (got %lld)", (long long)K);
  if (N % 16 != 0) return fail(PARO_ERR_INVALID, "out_features must be a multiple of 16 (got %lld)", (long long)N);
  if (K > (1 << 24) || N > (1 << 24)) return fail(PARO_ERR_INVALID, "shape out of range");
  if (!qweight || !qzeros || !scales || !out_wq || !out_sz) return fail(PARO_ERR_INVALID, "null pointer");
  PartTable pt;
  if (!fill_part_table(pt, n_parts, part_cols, 1) || (int64_t)pt.tiles * 16 != N)
    return fail(PARO_ERR_INVALID, "partition sizes must be positive multiples of 16 summing to out_features");
  hipStream_t st = (hipStream_t)stream;
  const int64_t words = K * N / 8;
  hipLaunchKernelGGL(repack_qweight_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                     (const unsigned*)qweight, (unsigned*)out_wq, (int)K, (int)N, wq_order ? 1 : 0);
  const int G = (int)(K / 128);
  (void)hipMemsetAsync(out_sz, 0, (size_t)G * pt.tsz * 64, st);
  const int64_t cols = (int64_t)G * N;
  hipLaunchKernelGGL(pack_sz_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, (const unsigned*)qzeros,
                     (const unsigned short*)scales, (unsigned*)out_sz, G, (int)N, pt);
  return check_launch("paro_repack_awq");
}

extern "C" int paro_pack_rotation(const int16_t* pairs, const void* theta, int64_t K, int n_parts, int krot,
                                  void* out_rot, void* stream) {
  using namespace paro;
  if (K <= 0 || K % 128 != 0) return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128");
  if (n_parts < 1 || n_parts > PARO_MAX_PARTS) return fail(PARO_ERR_INVALID, "n_parts must be in 1..%d", PARO_MAX_PARTS);
  if (krot < 1 || krot > 8) return fail(PARO_ERR_UNSUPPORTED, "packed rotation supports krot 1..8 (got %d)", krot);
  if (!pairs || !theta || !out_rot) return fail(PARO_ERR_INVALID, "null pointer");
  const int64_t n = (int64_t)n_parts * (K / 128) * 8;
  hipLaunchKernelGGL(pack_rot_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pairs,
                     (const unsigned short*)theta, (unsigned*)out_rot, (int)K, n_parts, krot);
  return check_launch("paro_pack_rotation");
}

extern "C" int paro_dequant_packed(const paro_linear_t* L, void* out_w, void* stream) {
  using namespace paro;
  if (!L || !out_w || !L->wq || !L->sz) return fail(PARO_ERR_INVALID, "null pointer");
  PartTable pt;
  if (L->K % 128 != 0 || !fill_part_table(pt, L->n_parts, L->part_cols, 1) || (int64_t)pt.tiles * 16 != L->N)
    return fail(PARO_ERR_INVALID, "bad shape");
  const int64_t total = L->K * L->N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (L->act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(dequant_packed_kernel<f16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->sz, (unsigned short*)out_w, (int)L->K, (int)L->N, pt, L->wq_order);
  else if (L->act_dtype == PARO_DTYPE_BF16)
    hipLaunchKernelGGL(dequant_packed_kernel<bf16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->sz, (unsigned short*)out_w, (int)L->K, (int)L->N, pt, L->wq_order);
  else
    return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_dequant_packed");
}

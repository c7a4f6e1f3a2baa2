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

// Rotation parameters of one (partition, group) -> the EXCHANGE SCHEDULE the fused GEMV executes with the
// rotation state in registers (one workgroup per (partition, group); stages depend on each other).
//
// In the kernel lane l always holds BOTH members (A, B) of one pair of the current stage, so a stage is
// four FMA-class ops on registers.  Between two stages every lane keeps one of its two channels and
// fetches ONE value from another lane (a single ds_bpermute_b32): the union of two perfect matchings is a
// set of alternating cycles, and walking each cycle in one direction gives every next-stage pair a lane
// that already holds one of its members.  Which member is kept and the (i, j) orientation are absorbed
// into the coefficients:
//     keep' = P A + Q B      give' = P B - Q A      B <- give' of lane src, A <- keep'
// with per-channel signs tracked here (the stored value of a channel may be the negated true value; the
// last stage restores them).  (P, Q) is always (cos a, sin a) of an angle a in {+-theta + k pi/2}; both are
// stored as signed 16-bit fixed point in units of 2^-14 (absolute error <= 3.1e-5, +-1 exact, so the
// identity and the zero-angle dummy pairs of optim/rotation.py:53 are exact): one word per lane and
// stage, decoded by two converts and used as integers (the kernel tracks the power-of-two scale) -- no
// v_sin / v_cos in the launch (they cost 32 of the stage's 68 issue cycles when tried) and half the bytes
// of ready-made fp32 pairs.  Every workgroup re-reads this stream
// for every group it covers, so its size is L2 -> L1 traffic on every CU: 3 KiB per group, the size of
// the checkpoint's own int16 pairs + fp16 theta.
// Stage t = 0 is the identity on the natural layout (channels 2l, 2l+1 from one coalesced load); stage
// t >= 1 is checkpoint stage t - 1; the last checkpoint stage produces both outputs in place,
//     out[a] = P A + Q B      out[b] = sigma (P B - Q A).
// Output, uint32 [p][g][3][64 lanes][4]:
//   chunk 0 / 1: stage words t = 0..3 / 4..7:  Q << 16 | P
//   chunk 2:     {4 * src of stages 0..3 (one byte each), 4 * src of stages 4..7, final Q << 16 | P,
//                 2a | 2b << 8 | (sigma < 0) << 31}
// Update of a pair follows rotation.cuh:53-56: xi' = c xi + s xj, xj' = c xj - s xi.
// A stage that is not a perfect matching of the 128 channels (the reference's converter raises
// "illegal pair", optim/rotation.py:36-37) sets *bad.
__device__ __forceinline__ unsigned coef_word(double P, double Q) {
  const int pi = (int)llrint(P * 16384.0), qi = (int)llrint(Q * 16384.0);
  return ((unsigned)qi << 16) | ((unsigned)pi & 0xffffu);
}

// One 64-lane workgroup per (partition, group): thread l plays lane l of the GEMV (its pair, its
// coefficients); only the walk along the alternating cycles is serial (thread 0, 64 steps per stage).
__global__ __launch_bounds__(64) void pack_rot_kernel(const int16_t* __restrict__ pairs,
                                                     const unsigned short* __restrict__ theta,
                                                     unsigned* __restrict__ out, int K, int nparts, int krot,
                                                     int* __restrict__ bad) {
  const int G = K / 128;
  const int g = (int)(blockIdx.x % G), p = (int)(blockIdx.x / G);
  const int l = threadIdx.x;
  unsigned* o = out + (int64_t)blockIdx.x * 768;

  __shared__ unsigned char chA[64], chB[64], lane_of[128], partner[128], pe_cur[128], pe_next[128], isi_cur[128],
      isi_next[128], keepA[64], src[64], done[64];
  __shared__ signed char tau[128];
  __shared__ int seen[128];
  __shared__ int invalid;

  chA[l] = (unsigned char)(2 * l);
  chB[l] = (unsigned char)(2 * l + 1);
  tau[2 * l] = tau[2 * l + 1] = 1;
  pe_cur[2 * l] = pe_cur[2 * l + 1] = 0;
  isi_cur[2 * l] = isi_cur[2 * l + 1] = 1;
  if (l == 0) invalid = 0;
  unsigned srcw[2] = {0u, 0u};
  __syncthreads();

  // rotation matrix entries of the CURRENT stage for the lane holding (a, b):  x_a' = c x_a + m x_b,  x_b' = c x_b - m x_a
  auto coeffs = [&](int t, int a, double& c, double& m) {
    if (t == 0) {
      c = 1.0;
      m = 0.0;
      return;
    }
    const unsigned short th = theta[((int64_t)p * krot + (t - 1)) * (K / 2) + g * 64 + pe_cur[a]];
    double sn, cs;
    sincos((double)f16_bits_to_f32(th), &sn, &cs);
    c = cs;
    m = isi_cur[a] ? sn : -sn;
  };

  for (int t = 0; t < krot; ++t) {     // 2-word stages: identity (t = 0) and checkpoint stages 0 .. krot-2
    // checkpoint stage t = the matching the lanes must hold AFTER this stage: thread l validates pair l
    {
      const int16_t* pr = pairs + ((int64_t)p * krot + t) * K + g * 128;
      seen[2 * l] = seen[2 * l + 1] = 0;
      __syncthreads();
      const int i = pr[2 * l], j = pr[2 * l + 1];
      const bool ok = i >= 0 && i <= 127 && j >= 0 && j <= 127 && i != j;
      if (ok) {
        atomicAdd(&seen[i], 1);
        atomicAdd(&seen[j], 1);
        partner[i] = (unsigned char)j;
        partner[j] = (unsigned char)i;
        pe_next[i] = pe_next[j] = (unsigned char)l;
        isi_next[i] = 1;
        isi_next[j] = 0;
      } else {
        invalid = 1;
      }
      __syncthreads();
      if (seen[2 * l] != 1 || seen[2 * l + 1] != 1) invalid = 1;
      lane_of[chA[l]] = lane_of[chB[l]] = (unsigned char)l;
      done[l] = 0;
      __syncthreads();
      if (invalid) {
        if (l == 0) atomicOr(bad, 1);
        return;
      }
    }
    // walk the alternating cycles: lane x keeps channel k and receives partner[k] from the lane holding it,
    // which therefore gives that channel away and keeps its other one
    if (l == 0) {
#pragma clang loop unroll(disable)
      for (int l0 = 0; l0 < 64; ++l0) {
        if (done[l0]) continue;
        int x = l0, k = chA[l0];
#pragma clang loop unroll(disable)
        do {
          done[x] = 1;
          keepA[x] = (unsigned char)(k == chA[x]);
          const int want = partner[k];
          const int x2 = lane_of[want];
          src[x] = (unsigned char)x2;
          k = (want == chA[x2]) ? chB[x2] : chA[x2];
          x = x2;
        } while (x != l0);
      }
    }
    __syncthreads();
    const int a = chA[l], b = chB[l];
    {
      double c, m;
      coeffs(t, a, c, m);
      // keep = a:  x_a' =  c x_a + m x_b ;  keep = b:  x_b' = -m x_a + c x_b
      const double alpha = keepA[l] ? c : -m, beta = keepA[l] ? m : c;
      o[((t >> 2) * 64 + l) * 4 + (t & 3)] = coef_word(alpha * tau[a], beta * tau[b]);
      srcw[t >> 2] |= (4u * src[l]) << (8 * (t & 3));
    }
    // new signs and the new layout (every thread has read the old ones above)
    const int k = keepA[l] ? a : b, gv = keepA[l] ? b : a;
    const signed char ntau = (signed char)((keepA[l] ? 1 : -1) * tau[a] * tau[b]);
    __syncthreads();
    tau[k] = 1;
    tau[gv] = ntau;
    chA[l] = (unsigned char)k;
    chB[l] = partner[k];
    pe_cur[2 * l] = pe_next[2 * l];
    pe_cur[2 * l + 1] = pe_next[2 * l + 1];
    isi_cur[2 * l] = isi_next[2 * l];
    isi_cur[2 * l + 1] = isi_next[2 * l + 1];
    __syncthreads();
  }
  // unused stage slots (krot < 8): identity, never executed
  for (int t = krot; t < 8; ++t) {
    o[((t >> 2) * 64 + l) * 4 + (t & 3)] = coef_word(1.0, 0.0);
    srcw[t >> 2] |= (4u * l) << (8 * (t & 3));
  }
  // last checkpoint stage: out[a] = c ta A + m tb B,  out[b] = -m ta A + c tb B = sigma (P B - Q A), sigma = ta tb
  {
    const int a = chA[l], b = chB[l];
    double c, m;
    coeffs(krot, a, c, m);
    unsigned* w = o + (128 + l) * 4;
    w[0] = srcw[0];
    w[1] = srcw[1];
    w[2] = coef_word(c * tau[a], m * tau[b]);
    w[3] = (2u * a) | ((2u * b) << 8) | ((tau[a] * tau[b] < 0) ? 0x80000000u : 0u);
  }
}

// Debug / verification: dense W[k, n] = (q - z) * s from the PACKED buffers, rounded once.
template <typename AT>
__global__ __launch_bounds__(256) void dequant_packed_kernel(const unsigned* __restrict__ wq,
                                                            const unsigned* __restrict__ sz,
                                                            unsigned short* __restrict__ out, int K, int N,
                                                            PartTable pt, int order, int gs) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)K * N) return;
  const int k = (int)(gid / N), n = (int)(gid % N);
  const int G = K / 128;
  const int g = k >> 7, kk = k & 127;
  const int gq = k / gs;   // quantisation group (row of the scale/zero array)
  const int i = kk >> 5, kb = (kk >> 3) & 3, e = kk & 7;
  const int t = n >> 4, lane = (kb << 4) | (n & 15);
  const int64_t chunk = order ? (int64_t)g * (N / 16) + t : (int64_t)t * G + g;
  const unsigned w = wq[(chunk * 64 + lane) * 4 + i];
  const int q = (int)((w >> (4 * ((e >> 1) + 4 * (e & 1)))) & 0xF);
  const int p = pt.part_of_tile(t);
  const int ts = pt.szt_start[p] + (t - pt.tile_start[p]);
  const unsigned word = sz[(((int64_t)gq * (pt.tsz / 4) + (ts >> 2)) * 16 + (n & 15)) * 4 + (ts & 3)];
  const float s = f16_bits_to_f32(word & 0xffffu);
  const float zf = f16_bits_to_f32(word >> 16);  // zero point
  out[gid] = Act<AT>::from_f32(((float)q - zf) * s);
}

}  // namespace paro

extern "C" int64_t paro_packed_qweight_bytes(int64_t K, int64_t N) {
  if (K <= 0 || N <= 0 || K % 128 != 0 || N % 16 != 0) return -1;
  return K * N / 2;
}

extern "C" int64_t paro_packed_sz_bytes(int64_t K, int group_size, int n_parts, const int32_t* part_cols) {
  paro::PartTable pt;
  const int gs = paro::quant_group(group_size);
  if (K <= 0 || K % 128 != 0 || gs < 0 || !paro::fill_part_table(pt, n_parts, part_cols, 1)) return -1;
  return (K / gs) * (int64_t)pt.tsz * 16 * 4;
}

extern "C" int64_t paro_packed_rot_bytes(int64_t K, int n_parts) {
  if (K <= 0 || K % 128 != 0 || n_parts < 1 || n_parts > PARO_MAX_PARTS) return -1;
  return (int64_t)n_parts * (K / 128) * 3072;
}

extern "C" int paro_repack_awq(const int32_t* qweight, const int32_t* qzeros, const void* scales, int64_t K, int64_t N,
                               int group_size, int n_parts, const int32_t* part_cols, int wq_order, void* out_wq, void* out_sz,
                               void* stream) {
  using namespace paro;
  if (K <= 0 || N <= 0 || K % 128 != 0)
    return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128 (got %lld)", (long long)K);
  if (N % 16 != 0) return fail(PARO_ERR_INVALID, "out_features must be a multiple of 16 (got %lld)", (long long)N);
  if (K > (1 << 24) || N > (1 << 24)) return fail(PARO_ERR_INVALID, "shape out of range");
  if (!qweight || !qzeros || !scales || !out_wq || !out_sz) return fail(PARO_ERR_INVALID, "null pointer");
  const int gs = quant_group(group_size);
  if (gs < 0) return fail(PARO_ERR_UNSUPPORTED, "Unsupported group_size: %d; expected 64 or 128", group_size);
  PartTable pt;
  if (!fill_part_table(pt, n_parts, part_cols, 1) || (int64_t)pt.tiles * 16 != N)
    return fail(PARO_ERR_INVALID, "partition sizes must be positive multiples of 16 summing to out_features");
  hipStream_t st = (hipStream_t)stream;
  const int64_t words = K * N / 8;
  hipLaunchKernelGGL(repack_qweight_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                     (const unsigned*)qweight, (unsigned*)out_wq, (int)K, (int)N, wq_order ? 1 : 0);
  const int G = (int)(K / gs);   // rows of the scale/zero array: quantisation groups
  (void)hipMemsetAsync(out_sz, 0, (size_t)G * pt.tsz * 64, st);
  const int64_t cols = (int64_t)G * N;
  hipLaunchKernelGGL(pack_sz_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, (const unsigned*)qzeros,
                     (const unsigned short*)scales, (unsigned*)out_sz, G, (int)N, pt);
  return check_launch("paro_repack_awq");
}

extern "C" int paro_pack_rotation(const int16_t* pairs, const void* theta, int64_t K, int n_parts, int krot,
                                  void* out_rot, int32_t* status, void* stream) {
  using namespace paro;
  if (K <= 0 || K % 128 != 0) return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128");
  if (n_parts < 1 || n_parts > PARO_MAX_PARTS) return fail(PARO_ERR_INVALID, "n_parts must be in 1..%d", PARO_MAX_PARTS);
  if (krot < 1 || krot > 8) return fail(PARO_ERR_UNSUPPORTED, "packed rotation supports krot 1..8 (got %d)", krot);
  if (!pairs || !theta || !out_rot || !status) return fail(PARO_ERR_INVALID, "null pointer");
  // The caller-owned status word reports stages that are not perfect matchings; nothing is allocated and
  // nothing synchronises here (the caller reads the word once the stream has drained).
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(status, 0, sizeof(int32_t), st);
  const int64_t n = (int64_t)n_parts * (K / 128);
  hipLaunchKernelGGL(pack_rot_kernel, dim3((unsigned)n), dim3(64), 0, st, pairs,
                     (const unsigned short*)theta, (unsigned*)out_rot, (int)K, n_parts, krot, (int*)status);
  return check_launch("paro_pack_rotation");
}

extern "C" int paro_dequant_packed(const paro_linear_t* L, void* out_w, void* stream) {
  using namespace paro;
  if (!L || !out_w || !L->wq || !L->sz) return fail(PARO_ERR_INVALID, "null pointer");
  PartTable pt;
  const int gs = quant_group(L->group_size);
  if (L->K % 128 != 0 || gs < 0 || !fill_part_table(pt, L->n_parts, L->part_cols, 1) || (int64_t)pt.tiles * 16 != L->N)
    return fail(PARO_ERR_INVALID, "bad shape");
  const int64_t total = L->K * L->N;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (L->act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(dequant_packed_kernel<f16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->sz, (unsigned short*)out_w, (int)L->K, (int)L->N, pt, L->wq_order, gs);
  else if (L->act_dtype == PARO_DTYPE_BF16)
    hipLaunchKernelGGL(dequant_packed_kernel<bf16>, grid, dim3(256), 0, st, (const unsigned*)L->wq,
                       (const unsigned*)L->sz, (unsigned short*)out_w, (int)L->K, (int)L->N, pt, L->wq_order, gs);
  else
    return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  return check_launch("paro_dequant_packed");
}

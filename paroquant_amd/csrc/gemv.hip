// Fused pairwise-rotation + INT4 dequant + GEMV/small-batch GEMM for gfx950 (decode path).
//
// One launch computes  y[b, :] = rotate_p(x[b, :] * cs_p) @ dequant(W)  for every merged
// partition p (qkv = 3 rotations, gate_up = 2) -- the work the reference spreads over
// `rotate` + AWQ/Marlin GEMM launches per partition + torch.cat + bias
// (transformers/modules.py:57-71, vllm/plugin.py:281-311).
//
// Mapping (CDNA4-first, not a warp-tiled port):
//   * work unit = (128-channel quantisation group g) x (TPW column tiles of 16 outputs).
//     A 64-lane wavefront owns one unit at a time: 64 lanes == the 64 Givens pairs of
//     the group, so the wave rotates ITS OWN slice of x in wave-private LDS with no
//     workgroup barrier, while its INT4 tiles (issued first, non-temporal, 1 KiB per
//     wave-load, straight to VGPRs) are still in flight from HBM.
//   * the INT4 tile is stored in MFMA B-fragment order (paro_repack_awq), so a lane's
//     16-byte load IS its four v_mfma_f32_16x16x32 B operands after a shift/and/or
//     unpack to (16 + q) halves; scales and zero points are applied once per
//     (group, column) on the fp32 MFMA result:  acc += s * (D - (16 + z) * sum_k x_k),
//     with sum_k x_k obtained from one extra MFMA against a ones fragment.
//   * batch rows (<= 16) ride in the MFMA M dimension: rows <= 4 occupy MFMA rows
//     0,4,8,12 so a single accumulator register per tile suffices.
//   * 4 waves of a workgroup take 4 different groups of the same columns and reduce
//     through LDS; a K-split across workgroups (grid.y) is combined in-launch by the
//     last-arriving workgroup (agent-scope release / acquire + arrival counter), so
//     the output is written exactly once, as fp16/bf16, with the bias.
#include "common.hpp"

namespace paro {

struct GemvArgs {
  const u32x4* wq;
  const unsigned* zq;
  const unsigned short* scales;
  const int16_t* pairs;
  const unsigned short* theta;
  const unsigned short* cs;
  const unsigned short* bias;
  const unsigned short* x;
  unsigned short* y;
  float* slabs;
  unsigned* counters;
  int K, N, G, rows, krot, nparts, ksplit, gps;  // gps = groups per K-split
  int part_tile_start[PARO_MAX_PARTS + 1];
  int part_cb_start[PARO_MAX_PARTS + 1];
};

constexpr int kXhStride = 136;  // halves per fragment row in LDS (128 + 8 pad -> b128 reads of 16 rows spread over banks)

template <typename AT, int TPW, int MB, int KROT>   // KROT = 8: coefficients preloaded; 0: runtime krot
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
  typedef Act<AT> A;
  typedef typename A::vec8 vec8;
  constexpr int MR = MB <= 4 ? 1 : (MB <= 8 ? 2 : 4);   // accumulator registers kept per tile
  constexpr int VW = MB >= 4 ? 4 : MB;                   // LDS vector width of the rotation state
  constexpr int NCH = MB / VW;
  constexpr int XR_FLOATS = MB * 128;
  constexpr int XH_HALVES = (MB + 1) * kXhStride;        // + one all-zero row for unused MFMA rows
  constexpr int WAVE_BYTES = XR_FLOATS * 4 + ((XH_HALVES * 2 + 15) / 16) * 16;
  constexpr int RED_FLOATS = 4 * TPW * MR * 64;
  constexpr int LDS_BYTES = (4 * WAVE_BYTES > RED_FLOATS * 4 ? 4 * WAVE_BYTES : RED_FLOATS * 4) + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.x, ks = blockIdx.y;

  int p = 0;
#pragma unroll
  for (int q = 1; q < PARO_MAX_PARTS; ++q)
    if (q < a.nparts && cb >= a.part_cb_start[q]) p = q;
  const int tile0 = a.part_tile_start[p] + (cb - a.part_cb_start[p]) * TPW;
  const int nt = min(TPW, a.part_tile_start[p + 1] - tile0);
  const int g_begin = ks * a.gps;
  const int g_end = min(a.G, g_begin + a.gps);

  float* xr = (float*)(lds + wave * WAVE_BYTES);
  unsigned short* xh = (unsigned short*)(lds + wave * WAVE_BYTES + XR_FLOATS * 4);

  // zero row for MFMA rows that carry no batch row
  for (int c = lane; c < kXhStride; c += 64) xh[MB * kXhStride + c] = 0;

  const int n = lane & 15, mq = lane >> 4;
  // A-fragment source row for this lane: MFMA row m' = lane & 15 carries batch row (m'>>2)*MR + (m'&3)
  const int mrow = lane & 15;
  const int brow = (mrow >> 2) * MR + (mrow & 3);
  const bool avalid = ((mrow & 3) < MR) && (brow < a.rows);
  const unsigned short* afrag = xh + (avalid ? brow : MB) * kXhStride + 8 * mq;

  float acc[TPW][MR];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[j][r] = 0.f;

  const int64_t pbase = (int64_t)p * a.krot;
  const int NW = a.N >> 3;

  for (int g = g_begin + wave; g < g_end; g += 4) {
    // ---- 1. rotation inputs for this group (small, issued first so their wait does not cover the tiles)
    float xs[MB][2];
    {
      const unsigned csv = *(const unsigned*)(a.cs + (int64_t)p * a.K + g * 128 + 2 * lane);
      const float c0 = f16_bits_to_f32(csv & 0xffffu), c1 = f16_bits_to_f32(csv >> 16);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        unsigned xv = 0;
        if (b < a.rows) xv = *(const unsigned*)(a.x + (int64_t)b * a.K + g * 128 + 2 * lane);
        xs[b][0] = A::to_f32(xv & 0xffffu) * c0;
        xs[b][1] = A::to_f32(xv >> 16) * c1;
      }
    }
    constexpr int KR = KROT > 0 ? KROT : 1;
    unsigned ijr[KR];
    float thr[KR];
    if constexpr (KROT > 0) {
#pragma unroll
      for (int r = 0; r < KROT; ++r) {
        ijr[r] = *(const unsigned*)(a.pairs + (pbase + r) * a.K + g * 128 + 2 * lane);
        thr[r] = f16_bits_to_f32(a.theta[(pbase + r) * (a.K / 2) + g * 64 + lane]);
      }
    }
    // ---- 2. stream this unit's INT4 tiles + their scales / zero points
    u32x4 qv[TPW];
    unsigned short sraw[TPW];
    unsigned zw[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (j < nt) {
        const int t = tile0 + j;
        qv[j] = __builtin_nontemporal_load(a.wq + ((int64_t)t * a.G + g) * 64 + lane);
        sraw[j] = a.scales[(int64_t)g * a.N + t * 16 + n];
        zw[j] = a.zq[(int64_t)g * NW + t * 2 + (n >> 3)];
      }
    }
    // ---- 3. rotate this group's slice of x in wave-private LDS
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int ch = b / VW, v = b % VW;
      xr[(ch * 128 + 2 * lane) * VW + v] = xs[b][0];
      xr[(ch * 128 + 2 * lane + 1) * VW + v] = xs[b][1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (KROT > 0)
      rotate_span_regs<VW, NCH, KR>(xr, ijr, thr, 0);
    else
      rotate_span_lds<VW, NCH>(xr, a.pairs + pbase * a.K + g * 128, a.K, a.theta, pbase * (a.K / 2) + g * 64,
                               a.K / 2, PARO_DTYPE_F16, a.krot, lane, 0);
    // fp32 state -> activation-dtype fragment rows
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int ch = b / VW, v = b % VW;
      const float v0 = xr[(ch * 128 + 2 * lane) * VW + v];
      const float v1 = xr[(ch * 128 + 2 * lane + 1) * VW + v];
      *(unsigned*)(xh + b * kXhStride + 2 * lane) = (unsigned)A::from_f32(v0) | ((unsigned)A::from_f32(v1) << 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 4. A fragments (4 x K=32) + per-row sums via a ones fragment
    vec8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const vec8*)(afrag + 32 * i);
    f32x4 sx = {0.f, 0.f, 0.f, 0.f};
    {
      const u32x4 ones = {A::kOnes, A::kOnes, A::kOnes, A::kOnes};
      const vec8 ob = __builtin_bit_cast(vec8, ones);
#pragma unroll
      for (int i = 0; i < 4; ++i) sx = A::mfma(af[i], ob, sx);
    }
    // ---- 5. per tile: unpack -> 4 MFMA -> scale / zero-point on the fp32 result
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (j < nt) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned w4[4];
          A::unpack(qv[j][i], w4);
          const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
          d = A::mfma(af[i], __builtin_bit_cast(vec8, wv), d);
        }
        const float s = f16_bits_to_f32(sraw[j]);
        const float zf = (float)(16 + ((zw[j] >> (4 * (n & 7))) & 0xFu));
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[j][r] = __builtin_fmaf(s, __builtin_fmaf(-zf, sx[r], d[r]), acc[j][r]);
      }
    }
  }

  // ---- reduce the 4 waves (4 groups in flight) through LDS
  __syncthreads();
  float* red = (float*)lds;
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) red[((wave * TPW + j) * MR + r) * 64 + lane] = acc[j][r];
  __syncthreads();

  const bool direct = (a.ksplit == 1);
  for (int e = tid; e < TPW * MR * 64; e += 256) {
    const int el = e & 63, r = (e >> 6) % MR, j = e / (MR * 64);
    const int b = (el >> 4) * MR + r;
    if (j >= nt || b >= a.rows) continue;
    float v = red[e] + red[e + TPW * MR * 64] + red[e + 2 * TPW * MR * 64] + red[e + 3 * TPW * MR * 64];
    const int col = (tile0 + j) * 16 + (el & 15);
    if (direct) {
      if (a.bias) v += A::to_f32(a.bias[col]);
      a.y[(int64_t)b * a.N + col] = A::from_f32(v);
    } else {
      a.slabs[((int64_t)ks * a.rows + b) * a.N + col] = v;
    }
  }
  if (direct) return;

  // ---- in-launch K-split combine: publish the slab, take a ticket, last arriver reduces.
  // Placement-independent hand-off: plain stores -> every wave drains -> barrier -> one lane
  // agent-scope release (+ asm wait the compiler cannot drop) -> relaxed agent atomic ticket;
  // the last arriver does ONE agent-scope acquire, then plain loads.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* flag = (unsigned*)(lds + LDS_BYTES - 16);
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *flag = __hip_atomic_fetch_add(a.counters + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*flag != (unsigned)(a.ksplit - 1)) return;
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  const int ncols = nt * 16;
  for (int e = tid; e < a.rows * ncols; e += 256) {
    const int b = e / ncols, c = e % ncols;
    const int col = tile0 * 16 + c;
    float v = 0.f;
    for (int s = 0; s < a.ksplit; ++s) v += a.slabs[((int64_t)s * a.rows + b) * a.N + col];
    if (a.bias) v += A::to_f32(a.bias[col]);
    a.y[(int64_t)b * a.N + col] = A::from_f32(v);
  }
  if (tid == 0) __hip_atomic_store(a.counters + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename AT, int TPW, int KROT>
static void launch_mb(const GemvArgs& a, dim3 grid, hipStream_t st) {
  if (a.rows <= 1)
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, 1, KROT>), grid, dim3(256), 0, st, a);
  else if (a.rows <= 4)
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, 4, KROT>), grid, dim3(256), 0, st, a);
  else if (a.rows <= 8)
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, 8, KROT>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gemv_kernel<AT, TPW, 16, KROT>), grid, dim3(256), 0, st, a);
}

template <typename AT>
static int launch_tpw(const GemvArgs& a, int tpw, dim3 grid, hipStream_t st) {
  if (a.krot != 8) {  // generic-krot build exists for TPW = 4 only (the caller forces tpw = 4)
    launch_mb<AT, 4, 0>(a, grid, st);
    return check_launch("paro_w4a16_gemv");
  }
  switch (tpw) {
    case 1: launch_mb<AT, 1, 8>(a, grid, st); break;
    case 2: launch_mb<AT, 2, 8>(a, grid, st); break;
    case 4: launch_mb<AT, 4, 8>(a, grid, st); break;
    case 8: launch_mb<AT, 8, 8>(a, grid, st); break;
    default: return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave must be 1, 2, 4 or 8 (got %d)", tpw);
  }
  return check_launch("paro_w4a16_gemv");
}

int validate_linear(const paro_linear_t* L) {
  if (!L) return fail(PARO_ERR_INVALID, "null layer descriptor");
  if (L->K <= 0 || L->K % 128 != 0) return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128 (got %lld)", (long long)L->K);
  if (L->n_parts < 1 || L->n_parts > PARO_MAX_PARTS) return fail(PARO_ERR_INVALID, "n_parts must be in 1..%d", PARO_MAX_PARTS);
  int64_t sum = 0;
  for (int i = 0; i < L->n_parts; ++i) {
    if (L->part_cols[i] <= 0 || L->part_cols[i] % 16 != 0)
      return fail(PARO_ERR_INVALID, "partition %d has %d columns; must be a positive multiple of 16", i, L->part_cols[i]);
    sum += L->part_cols[i];
  }
  if (sum != L->N) return fail(PARO_ERR_INVALID, "sum(part_cols) = %lld != N = %lld", (long long)sum, (long long)L->N);
  if (L->K > (1 << 24) || L->N > (1 << 24)) return fail(PARO_ERR_INVALID, "shape out of range");
  if (L->krot < 1 || L->krot > 16) return fail(PARO_ERR_UNSUPPORTED, "Unsupported KROT = %d; supported: 1..16", L->krot);
  if (L->act_dtype != PARO_DTYPE_F16 && L->act_dtype != PARO_DTYPE_BF16)
    return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  if (!L->wq || !L->zq || !L->scales || !L->pairs || !L->theta || !L->channel_scales)
    return fail(PARO_ERR_INVALID, "null parameter pointer");
  return PARO_OK;
}

// Heuristic launch shape (calibrated on MI355X): enough workgroups to cover the
// 256 CUs a few times over, >= 4 groups (one per wave) per workgroup.
void gemv_autotune(const paro_linear_t* L, int64_t rows, int& tpw, int& ksplit) {
  const int G = (int)(L->K / 128);
  const int64_t tiles = L->N / 16;
  if (tpw <= 0) {
    tpw = 4;
    if (tiles * G >= 64 * 1024) tpw = 8;   // >= 64 MiB of tiles: longer streams per wave
    if (rows > 8 && tpw > 4) tpw = 4;      // MR = 4 accumulators per tile
  }
  int64_t ncb = 0;
  for (int i = 0; i < L->n_parts; ++i) ncb += (L->part_cols[i] / 16 + tpw - 1) / tpw;
  if (ksplit <= 0) {
    const int64_t target = 1024;           // ~4 workgroups per CU
    int64_t ksp = (target + ncb - 1) / ncb;
    const int max_split = (G + 3) / 4;     // keep >= 4 groups per workgroup
    if (ksp > max_split) ksp = max_split;
    if (ksp < 1) ksp = 1;
    ksplit = (int)ksp;
  }
  if (ksplit > G) ksplit = G;
}

}  // namespace paro

extern "C" int64_t paro_linear_workspace_bytes(const paro_linear_t* L, int64_t rows) {
  using namespace paro;
  if (validate_linear(L) != PARO_OK) return -1;
  if (rows < 0) return -1;
  const int G = (int)(L->K / 128);
  const int64_t r = rows < 1 ? 1 : rows;
  // GEMV: up to G slabs of [rows<=16, N] fp32.  GEMM: n_parts rotated copies of x [rows, K] (16-bit).
  const int64_t gemv = (int64_t)G * (r < 16 ? r : 16) * L->N * 4;
  const int64_t gemm = r > 16 ? (int64_t)L->n_parts * r * L->K * 2 : 0;
  return PARO_WS_COUNTER_BYTES + (gemv > gemm ? gemv : gemm);
}

extern "C" int paro_w4a16_gemv(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                               int64_t workspace_bytes, int tiles_per_wave, int ksplit, void* stream) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows == 0) return PARO_OK;
  if (rows < 0 || rows > 16) return fail(PARO_ERR_INVALID, "paro_w4a16_gemv handles 1..16 rows (got %lld)", (long long)rows);
  if (!x || !y) return fail(PARO_ERR_INVALID, "null pointer");
  int tpw = tiles_per_wave, ksp = ksplit;
  if (tpw != 0 && tpw != 1 && tpw != 2 && tpw != 4 && tpw != 8)
    return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave must be 0 (auto), 1, 2, 4 or 8 (got %d)", tpw);
  if (L->krot != 8) tpw = 4;
  gemv_autotune(L, rows, tpw, ksp);
  const int G = (int)(L->K / 128);

  GemvArgs a;
  a.wq = (const u32x4*)L->wq;
  a.zq = (const unsigned*)L->zq;
  a.scales = (const unsigned short*)L->scales;
  a.pairs = L->pairs;
  a.theta = (const unsigned short*)L->theta;
  a.cs = (const unsigned short*)L->channel_scales;
  a.bias = (const unsigned short*)L->bias;
  a.x = (const unsigned short*)x;
  a.y = (unsigned short*)y;
  a.K = (int)L->K;
  a.N = (int)L->N;
  a.G = G;
  a.rows = (int)rows;
  a.krot = L->krot;
  a.nparts = L->n_parts;
  a.ksplit = ksp;
  a.gps = (G + ksp - 1) / ksp;
  a.ksplit = (G + a.gps - 1) / a.gps;  // drop empty splits
  int tiles = 0, cbs = 0;
  for (int i = 0; i < PARO_MAX_PARTS; ++i) {
    a.part_tile_start[i] = tiles;
    a.part_cb_start[i] = cbs;
    if (i < L->n_parts) {
      const int pt = L->part_cols[i] / 16;
      tiles += pt;
      cbs += (pt + tpw - 1) / tpw;
    }
  }
  a.part_tile_start[PARO_MAX_PARTS] = tiles;
  a.part_cb_start[PARO_MAX_PARTS] = cbs;
  a.slabs = nullptr;
  a.counters = nullptr;
  if (a.ksplit > 1) {
    const int64_t need = PARO_WS_COUNTER_BYTES + (int64_t)a.ksplit * rows * L->N * 4;
    if (!workspace || workspace_bytes < need)
      return fail(PARO_ERR_INVALID, "workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
    if (cbs * 4 > PARO_WS_COUNTER_BYTES) return fail(PARO_ERR_INVALID, "too many column blocks for the counter area");
    a.counters = (unsigned*)workspace;
    a.slabs = (float*)((char*)workspace + PARO_WS_COUNTER_BYTES);
  }
  dim3 grid((unsigned)cbs, (unsigned)a.ksplit);
  hipStream_t st = (hipStream_t)stream;
  if (L->act_dtype == PARO_DTYPE_F16) return launch_tpw<f16>(a, tpw, grid, st);
  return launch_tpw<bf16>(a, tpw, grid, st);
}

// Host side of the decode-chain GEMV (chain_impl.hpp): validation, launch shape, ABI entry points.
#include <limits.h>
#include <stdlib.h>

#include "chain_impl.hpp"

namespace paro {
int validate_linear(const paro_linear_t* L);   // gemv.hip
int launch_rotate(const void* x, void* out, const int16_t* idx, const void* theta, const void* scales, int64_t rows,
                  int64_t hidden, int krot, int gs, int x_dt, int p_dt, hipStream_t st, int nparts);   // rotate.hip
int launch_rotate_mfma(const void* x, void* out, const void* rmat, int64_t rows, int64_t K, int nparts, int dt, hipStream_t st);   // rotate.hip
int launch_prerot_sched(const void* x, void* out, const void* rot, const void* cs, int64_t rows, int64_t K, int krot, int nparts,
                        int dt, int frag_row_tiles, hipStream_t st);   // rotate.hip

// Launch shape: one 128-column block (PAIR: one gate + one up block) per workgroup, K cut so that the grid fills the
// chip once (~1 workgroup per CU); 4 waves when a K-slice has <= 4 groups per block, else 8.
static void chain_shape(int nblocks, int G, bool pair, int rows, int& ksplit, int& waves) {
  const int cus = device_cu_count();
  const bool auto_ks = ksplit <= 0;
  if (ksplit <= 0) {
    int ks = (cus + nblocks / 2) / nblocks;           // round(cus / blocks)
    if (ks < 1) ks = 1;
    if (ks > 16) ks = 16;
    if (ks > G) ks = G;
    // more rows: the reducer's polls grow with rows x slices; keep the split small -- but not as small as 4 everywhere
    // (CHAIN_SWEEP at 8 / 16 rows, profiles/r03_chain_shape_sweep_rows.jsonl: deep K (down_proj) wants 8 slices at <= 8 rows
    // -- Qwen3-4B 11.6 -> 9.8 us, Llama-3-8B 13.8 -> 12.4 -- and 5 at <= 16; the others 5 / 4: Qwen3-4B qkv at 8 rows 8.0 -> 7.2)
    if (rows > 4) {
      const int cap = rows <= 8 ? (G >= 64 ? 8 : 5) : (G >= 64 ? 5 : 4);
      if (ks > cap) ks = cap;
    }
    ksplit = ks;
  }
  if (ksplit > G) ksplit = G;
  int gps = (G + ksplit - 1) / ksplit;
  // three groups per slice leave one of the four waves idle and cost more slices to poll: four, when the grid still covers half the
  // chip (Qwen3-4B o_proj at 2..4 rows: 11 slices of 3 groups 7.5 us, 8 slices of 4 groups 6.5; profiles/r03_chain_shape_sweep_rows.jsonl)
  if (auto_ks && !pair && gps == 3 && nblocks * ((G + 3) / 4) >= cus / 2) gps = 4;
  ksplit = (G + gps - 1) / gps;
  if (waves <= 0) waves = pair ? (gps <= 2 ? 4 : 8) : (gps <= 4 ? 4 : 8);
}
}  // namespace paro

namespace paro { static void* g_chain_dbg = nullptr; }
#ifdef PARO_CHAIN_DIAG
// diagnostic builds only (make EXTRA=-DPARO_CHAIN_DIAG): 16 phase stamps per workgroup of the NEXT chain launches go to `buf`
extern "C" void paro_chain_set_debug(void* buf) { paro::g_chain_dbg = buf; }
#endif

extern "C" int64_t paro_chain_workspace_bytes(const paro_linear_t* L, int64_t rows) {
  using namespace paro;
  if (validate_linear(L) != PARO_OK || rows < 1 || rows > kChainMaxRows) return -1;
  return PARO_WS_COUNTER_BYTES + (int64_t)15 * rows * L->N * 8;
}

extern "C" int paro_rotate_parts(const paro_linear_t* L, const void* x, void* x_rot, int64_t rows, void* stream) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows == 0) return PARO_OK;
  if (!x || !x_rot || rows < 0) return fail(PARO_ERR_INVALID, "null pointer / bad row count");
  static const int env_sched = getenv("PARO_PREROT_SCHED") ? atoi(getenv("PARO_PREROT_SCHED")) : 1;   // 0: the stage kernel (A/B)
  if (L->rmat && rows >= 256)     // many rows: the dense per-group product on the matrix cores -- the pre-pass paro_w4a16_gemm runs (gemm.hip)
    rc = launch_rotate_mfma(x, x_rot, L->rmat, rows, L->K, L->n_parts, L->act_dtype, (hipStream_t)stream);
  else if (L->rot && L->krot <= 8 && env_sched != 0)   // the schedule pre-pass (rotate.hip): the in-kernel rotation's bits, plain rows
    rc = launch_prerot_sched(x, x_rot, L->rot, L->channel_scales, rows, L->K, L->krot, L->n_parts, L->act_dtype, 0, (hipStream_t)stream);
  else
    rc = launch_rotate(x, x_rot, L->pairs, L->theta, L->channel_scales, rows, L->K, L->krot, 128, L->act_dtype, PARO_DTYPE_F16,
                       (hipStream_t)stream, L->n_parts);
  if (rc != PARO_OK) return rc;
  return check_launch("paro_rotate_parts");
}

extern "C" int paro_w4a16_gemv_chain(const paro_linear_t* L, const paro_chain_t* C, int64_t rows, void* workspace,
                                     int64_t workspace_bytes, int ksplit, int waves, void* stream) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (!C) return fail(PARO_ERR_INVALID, "null chain descriptor");
  if (rows == 0) return PARO_OK;
  if (rows < 0 || rows > kChainMaxRows) return fail(PARO_ERR_UNSUPPORTED, "the chain GEMV is a decode path: 1..%d rows (got %lld)", kChainMaxRows, (long long)rows);
  if (!C->x_rot) return fail(PARO_ERR_INVALID, "x_rot is null");
  if (quant_group(L->group_size) != 128) return fail(PARO_ERR_UNSUPPORTED, "the chain GEMV is built for group_size 128 (got %d)", L->group_size);
  for (int i = 0; i < L->n_parts; ++i)
    if (L->part_cols[i] % 128 != 0)
      return fail(PARO_ERR_UNSUPPORTED, "the chain GEMV owns 128-column blocks: partition %d has %d columns", i, L->part_cols[i]);
  if (ksplit < 0 || ksplit > 16) return fail(PARO_ERR_INVALID, "ksplit must be in 0..16 (got %d)", ksplit);
  if (waves != 0 && waves != 4 && waves != 8) return fail(PARO_ERR_INVALID, "waves must be 0 (auto), 4 or 8 (got %d)", waves);
  if ((uint64_t)L->N * (uint64_t)rows >= (1ull << 31) || (uint64_t)L->K * (uint64_t)rows * L->n_parts >= (1ull << 31))
    return fail(PARO_ERR_UNSUPPORTED, "layer too large for the 32-bit element offsets of the chain GEMV");
  const int G = (int)(L->K / 128);
  const int nblocks_all = (int)(L->N / 128);
  const paro_linear_t* X = C->next;
  const bool pair = C->next_act == PARO_CHAIN_ACT_SILU_MUL || C->next_act == PARO_CHAIN_ACT_GELU_TANH_MUL;
  if (C->next_act != PARO_CHAIN_ACT_NONE && !pair) return fail(PARO_ERR_INVALID, "unknown next_act %d", C->next_act);
  ChainArgs a;
  a.nrot = nullptr; a.ncs = nullptr; a.nx = nullptr; a.np = 0; a.Gn = 0; a.nblk0 = 0; a.act = 0; a.blk0 = 0; a.up_off = 0;
  if (X) {
    rc = validate_linear(X);
    if (rc != PARO_OK) return rc;
    if (!X->rot || X->krot > 8) return fail(PARO_ERR_UNSUPPORTED, "the consumer's rotation needs the packed schedule (krot <= 8)");
    if (X->act_dtype != L->act_dtype) return fail(PARO_ERR_INVALID, "producer and consumer activation types differ");
    if (!C->next_x_rot) return fail(PARO_ERR_INVALID, "next_x_rot is null");
    if (C->next_col0 < 0 || C->next_col0 % 128 != 0 || C->next_col0 + (pair ? 2 : 1) * X->K > L->N)
      return fail(PARO_ERR_INVALID, "the consumer reads columns [%lld, %lld) of %lld outputs: must lie inside and start on a 128-column block",
                  (long long)C->next_col0, (long long)(C->next_col0 + (pair ? 2 : 1) * X->K), (long long)L->N);
    if ((uint64_t)X->K * (uint64_t)rows * X->n_parts >= (1ull << 31)) return fail(PARO_ERR_UNSUPPORTED, "consumer too large for 32-bit offsets");
    a.nrot = (const unsigned*)X->rot;
    a.ncs = (const unsigned short*)X->channel_scales;
    a.nx = (unsigned short*)C->next_x_rot;
    a.np = X->n_parts;
    a.Gn = (int)(X->K / 128);
    a.nblk0 = (int)(C->next_col0 / 128);
    a.act = pair ? C->next_act : 0;
    if (pair) {
      if (C->next_col0 != 0 || 2 * X->K != L->N || L->n_parts != 2 || L->part_cols[0] != X->K)
        return fail(PARO_ERR_UNSUPPORTED, "silu(gate) * up needs the merged gate|up projection: two partitions of the consumer's K columns");
      if (C->ssq_out) return fail(PARO_ERR_INVALID, "ssq_out is not defined for the gate|up pair");
      a.blk0 = 0;
      a.up_off = a.Gn;
    }
  } else if (pair) {
    return fail(PARO_ERR_INVALID, "next_act without next");
  }
  if (!C->y && !X) return fail(PARO_ERR_INVALID, "neither y nor a consumer: nothing to write");
  if (C->ssq_in && (C->ssq_in_blocks < 1 || C->ssq_in_blocks > 128 || C->norm_dim < 1))
    return fail(PARO_ERR_INVALID, "ssq_in: 1..128 blocks and a positive norm_dim (got %d, %lld)", C->ssq_in_blocks, (long long)C->norm_dim);
  const int nblocks = pair ? a.Gn : nblocks_all;
  if (nblocks_all >= (1 << kChainBlockBits) || nblocks_all >= kChainEpochWords)
    return fail(PARO_ERR_UNSUPPORTED, "too many column blocks for the K-split tags (%d)", nblocks_all);

  int ks = ksplit, wv = waves;
  chain_shape(nblocks, G, pair, (int)rows, ks, wv);
  a.wq = (const u32x4*)L->wq;
  a.sz = (const unsigned*)L->sz;
  a.x = (const unsigned short*)C->x_rot;
  a.G = G;
  a.T = (int)(L->N / 16);
  a.order = L->wq_order;
  a.rows = (int)rows;
  a.szrow = (a.T >> 2) * 64;   // partitions are multiples of 8 tiles: the padded tile space is the tile space
  {
    int acc = 0;
    for (int q = 0; q < PARO_MAX_PARTS - 1; ++q) {
      acc += q < L->n_parts ? L->part_cols[q] / 128 : 0;
      a.pb[q] = (q + 1 < L->n_parts) ? acc : INT_MAX;
    }
  }
  a.y = (unsigned short*)C->y;
  a.bias = (const unsigned short*)L->bias;
  a.residual = (const unsigned short*)C->residual;
  a.ssq_in = C->ssq_in;
  a.ssq_out = C->ssq_out;
  a.ssq_in_n = C->ssq_in ? C->ssq_in_blocks : 0;
  a.inv_norm_dim = C->ssq_in ? 1.0f / (float)C->norm_dim : 0.f;
  a.eps = C->eps;
  a.N = (int)L->N;
  a.dbg = (unsigned long long*)g_chain_dbg;
  typedef int (*launch_fn)(const ChainArgs&, int, bool, dim3, hipStream_t);
  // row classes 1 / <= 4 / <= 8 / <= 16: one, one, two, four accumulator registers per tile
  static const launch_fn table[2][4] = {{launch_chain_f16_m1, launch_chain_f16_m4, launch_chain_f16_m8, launch_chain_f16_m16},
                                        {launch_chain_bf16_m1, launch_chain_bf16_m4, launch_chain_bf16_m8, launch_chain_bf16_m16}};
  const launch_fn fn = table[L->act_dtype == PARO_DTYPE_F16 ? 0 : 1][rows <= 1 ? 0 : (rows <= 4 ? 1 : (rows <= 8 ? 2 : 3))];
  hipStream_t st = (hipStream_t)stream;
  for (;;) {
    const int gps = (G + ks - 1) / ks;
    ks = (G + gps - 1) / gps;
    a.ksplit = ks;
    a.gps = gps;
    a.slabs = nullptr;
    a.epochs = nullptr;
    if (ks > 1) {
      const int64_t need = PARO_WS_COUNTER_BYTES + (int64_t)(ks - 1) * rows * L->N * 8;
      if (!workspace || workspace_bytes < need)
        return fail(PARO_ERR_INVALID, "workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
      a.epochs = (unsigned*)workspace;
      a.slabs = (unsigned long long*)((char*)workspace + PARO_WS_COUNTER_BYTES);
    }
    rc = fn(a, wv, pair, dim3((unsigned)nblocks, (unsigned)ks), st);
    if (rc == PARO_ERR_NOT_RESIDENT_CHAIN && ksplit == 0 && ks > 1) {   // the automatic split does not fit: shrink it
      --ks;
      continue;
    }
    break;
  }
  if (rc == PARO_ERR_NOT_RESIDENT_CHAIN) rc = PARO_ERR_UNSUPPORTED;
  if (rc != PARO_OK) return rc;
  return check_launch("paro_w4a16_gemv_chain");
}

extern "C" int paro_chain_launch_shape(const paro_linear_t* L, const paro_chain_t* C, int64_t rows, int* ksplit, int* waves) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (!C || !ksplit || !waves) return fail(PARO_ERR_INVALID, "null pointer");
  const bool pair = (C->next_act == PARO_CHAIN_ACT_SILU_MUL || C->next_act == PARO_CHAIN_ACT_GELU_TANH_MUL) && C->next;
  const int nblocks = pair ? (int)(C->next->K / 128) : (int)(L->N / 128);
  chain_shape(nblocks, (int)(L->K / 128), pair, (int)rows, *ksplit, *waves);
  return PARO_OK;
}

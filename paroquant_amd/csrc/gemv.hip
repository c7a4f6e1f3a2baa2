// Host side of the fused GEMV: validation, launch-shape heuristic, ABI entry point.
#include <stdlib.h>

#include <algorithm>

#include "gemv_impl.hpp"

namespace paro {

int gemm_ksplit(const paro_linear_t* L, int64_t rows);   // gemm.hip
int gemm4_ksplit(const paro_linear_t* L, int64_t rows);  // gemm.hip
int launch_rotate(const void* x, void* out, const int16_t* idx, const void* theta, const void* scales,
                  int64_t rows, int64_t hidden, int krot, int gs, int x_dt, int p_dt, hipStream_t st, int nparts);
int launch_prerot_sched(const void* x, void* out, const void* rot, const void* cs, int64_t rows, int64_t K, int krot, int nparts,
                        int dt, int frag_row_tiles, hipStream_t st);   // rotate.hip

int validate_linear(const paro_linear_t* L) {
  if (!L) return fail(PARO_ERR_INVALID, "null layer descriptor");
  if (L->K <= 0 || L->K % 128 != 0)
    return fail(PARO_ERR_INVALID, "in_features must be a multiple of 128 (got %lld)", (long long)L->K);
  if (L->n_parts < 1 || L->n_parts > PARO_MAX_PARTS)
    return fail(PARO_ERR_INVALID, "n_parts must be in 1..%d", PARO_MAX_PARTS);
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
  if (!L->wq || !L->sz || !L->pairs || !L->theta || !L->channel_scales)
    return fail(PARO_ERR_INVALID, "null parameter pointer");
  if (L->krot <= 8 && !L->rot) return fail(PARO_ERR_INVALID, "packed rotation words missing (paro_pack_rotation)");
  if (quant_group(L->group_size) < 0)
    return fail(PARO_ERR_UNSUPPORTED, "Unsupported group_size: %d; expected 64 or 128", L->group_size);
  return PARO_OK;
}

constexpr int kMaxKsplit = 16;
constexpr int kSharedRotRows = kShrTaskRows;   // rows per producer task of mode 3 (gemv_impl.hpp)
inline int shared_rot_row_tasks(int64_t rows) { return rows <= kSharedRotRows ? 1 : (int)((rows + kSharedRotRows - 1) / kSharedRotRows); }
// Producer workgroups of a mode-3 launch: one task per wave when the chip has room beside the consumers, up to four tasks per wave (run
// one after the other: +0.3 .. 0.5 us on the hand-over) when it has not; -1 = the launch does not fit at all.
// resident at once: two 8-wave workgroups per CU up to 8 rows (113 VGPRs) and at 9..16 rows on <= 2-tile blocks (119); one on 4- / 8-tile blocks there (145 / 190)
inline int64_t shared_rot_cap(int64_t rows, int wv, int tpw) { return 256 * ((rows <= 8 || tpw <= 2) ? 2 : 1) * (wv <= 4 ? 2 : 1); }
inline int shared_rot_prod_wgs(int64_t units, int wv, int64_t consumers, int64_t cap) {
  const int64_t want = (units + wv - 1) / wv, room = cap - consumers;
  if (room >= want) return (int)want;
  if (room >= 1 && room * 4 >= want) return (int)room;
  return -1;
}
// Mode 3's HYBRID form (every wave rotates its own first group under the first tiles' latency, the producers the rest): where it beat both
// the replicated rotation and the pure form (profiles/r06_shared_rot_hybrid.jsonl) -- 3..4 rows on mid-width / wide outputs (Qwen3-4B qkv
// 7.1 -> 6.2 us, gate_up 9.4 -> 8.6; Llama-3-8B 7.8 -> 7.3, 14.7 -> 12.9; narrow K-split outputs stall on the hand-over there: o +1.1), and
// 5..8 rows on narrow deep-K outputs (Qwen3-4B down 10.4 -> 9.2).  PARO_SHR_SELF = 0 / 1 overrides (experiments).
inline int shared_rot_self(const paro_linear_t* L, int64_t rows, int ksp, int wv) {
  static const int env_self = getenv("PARO_SHR_SELF") ? atoi(getenv("PARO_SHR_SELF")) : -1;
  if (rows > 8 || rows <= 1) return 0;               // (one row: the replicated rotation, or -- explicit mode 3 -- the pure form)
  if (env_self >= 0) return env_self ? 1 : 0;
  const int G = (int)(L->K / 128), gps = (G + ksp - 1) / ksp;
  const int64_t tiles = L->N / 16;
  if (gps - wv <= 0) return 0;                       // nobody has a second group: nothing to share
  // (one row: no hand-over of any kind on the batch-1 path; Llama-3-8B gate_up alone would gain 2.5 % from the hybrid)
  if (rows == 2) return tiles >= 1024 ? 1 : 0;       // wide outputs: Qwen3-4B gate_up 8.66 -> 8.26 us, Llama-3-8B 13.29 -> 12.74
  if (rows <= 4) return tiles >= 320 ? 1 : 0;
  return ((tiles < 320 && G >= 64) || (tiles >= 1024 && gps >= 3 * wv)) ? 1 : 0;   // deep narrow outputs; wide ones with >= 3 groups per wave (Llama-3-8B gate_up 14.8 -> 14.0)
}
constexpr int kSharedRotMinRows = 5;    // automatic mode 3 from this many rows on (profiles/r06_shared_rot_ab.jsonl: ahead from the 8-row instantiation on; 17 = never)

// Launch-shape heuristic (calibrated on MI355X with tools/sweep_gemv.py):
// one workgroup covers all of K whenever that still yields >= ~1 workgroup per CU.
void gemv_autotune(const paro_linear_t* L, int64_t rows, int& tpw, int& ksplit, int& waves, bool deferred = false) {
  // Measured on MI355X (tools/sweep_gemv.py; Llama-3-8B, Qwen3-4B, Qwen3-0.6B shapes, M = 1):
  //   * every workgroup rotates all the groups it covers, so the total rotation work is
  //     (#column blocks) x K/128 group rotations (VALU issue + 3 KiB of schedule through the CU's L1 each):
  //     wide outputs want few, fat column blocks (tpw 8), narrow outputs want the rotation cut by a K-split;
  //   * the in-launch K-split (data-tagged granules) costs one round trip to the coherence point: it pays
  //     for narrow-N layers with K >= 4096 (o_proj, down_proj: tpw 4 x ksplit 4 -- o_proj 6.3 -> 6.0 us,
  //     Qwen3-4B o_proj 6.2 -> 5.6 over tpw 2 x ksplit 2).
  const int G = (int)(L->K / 128);
  const int64_t tiles = L->N / 16;
  if (rows > 16) {
    // 17..64 rows (pre-rotated activations, 2 / 4 MFMA row tiles per weight fragment): every workgroup reads
    // all of x_rot, so few, fat column blocks -- as many tiles per wave as the accumulators allow -- and a
    // K-split only to reach ~256 workgroups on narrow outputs
    // (round 6, behind the schedule pre-pass -- x in MFMA-fragment order, one 1-KiB wave load per row tile and k-step: re-reading x is
    // cheap now, and at 17..32 rows outputs below 1024 tiles want 2-tile blocks -- more workgroups in flight -- with the K-split that
    // brings them to <= 256; wide merged projections keep 4-tile blocks, on four waves.  profiles/r06_sweep_rows32_frag.jsonl, us rule
    // before / after: Qwen3-4B qkv 12.6 -> 9.7, o 12.3 -> 10.3, gate_up 19.5 -> 17.8, down 15.0 -> 13.5; Llama-3-8B qkv 14.5 -> 12.6,
    // o 12.9 -> 11.1, gate_up 27.0 -> 25.6, down 20.2 -> 19.9.  At 33..64 rows the rule below is within 2 % of the best of 20 shapes.)
    const int cap = rows <= 32 ? 4 : 2;
    const bool auto_all = tpw <= 0 && waves <= 0 && ksplit <= 0;
    if (auto_all && rows <= 32) {
      tpw = tiles < 1024 ? 2 : 4;
      waves = tiles < 1024 ? 8 : 4;
    }
    if (tpw <= 0 || tpw > cap) tpw = cap;
    if (waves <= 0 || waves > 8) waves = 8;
    if (ksplit <= 0) {
      const int64_t cbs = (tiles + tpw - 1) / tpw;
      ksplit = (int)(256 / (cbs > 0 ? cbs : 1));
      if (ksplit > 4) ksplit = 4;
      if (ksplit > G / 8) ksplit = G / 8;
      if (ksplit < 1) ksplit = 1;
      while (ksplit > 1 && cbs * ksplit > 256) --ksplit;   // whole grid resident (one 8-wave workgroup per CU at worst)
    }
    return;
  }
  const bool auto_tpw = tpw <= 0, auto_ks = ksplit <= 0, auto_wv = waves <= 0;
  const bool narrow = tiles <= 320;
  if (auto_tpw && auto_ks && auto_wv && tiles < 1024 && G >= 64) {
    // deep K (down_proj, every 70B-class projection but gate_up): a 4-way K-split cuts the replicated rotation
    // and still leaves >= 256 workgroups (Llama-3-70B: qkv 20.0 -> 14.2 us, down 31.5 -> 25.3, o 12.2 -> 11.6)
    tpw = (tiles >= 512 && G >= 128) ? 8 : 4;
    ksplit = 4;
    waves = 8;
    // per-rank shapes of tensor-parallel 70B-class models (tools/sweep_gemv.py --tp 4 / 8): wide enough for ~200 column
    // blocks of 4 tiles -> no K-split (TP = 4 gate_up 8192 -> 14336: 16.9 -> 15.0 us); very narrow (TP = 8 qkv
    // 8192 -> 1280: 80 tiles) -> 2-tile blocks, 16 waves (6.7 -> 6.0 us)
    if (tiles >= 768 && G < 128 && rows <= 4) { tpw = 4; ksplit = 1; waves = 16; }
    // (re-swept in round 4: 16 groups per K-slice on sixteen waves is one unit per wave -- eight waves run two: TP = 8 qkv 6.18 -> 5.75 us)
    else if (tiles < 128 && rows <= 4) { tpw = 2; waves = (G + 3) / 4 > 16 ? 16 : 8; }
    else if (rows == 1) {
      // one split fewer when that brings (column blocks x K-slices) down to one round of 8-wave workgroups over the 256 CUs
      // (Llama-3-70B TP = 2 qkv, 8192 -> 5120: 80 blocks x 4 = 320 -> x 3 = 240: 9.29 -> 8.40 us; profiles/r04_sweep_llama3-70b_tp2.jsonl)
      const int64_t cbs = (tiles + tpw - 1) / tpw;
      if (cbs * 4 > 256 && cbs * 3 <= 256) ksplit = 3;
    }
  } else if (auto_tpw && auto_ks && auto_wv && narrow && G >= 32) {   // o_proj class
    tpw = 4; ksplit = 4;
    waves = G / 4 <= 8 ? 4 : 8;   // 8 groups per split: 4 waves x 2 units beat 8 x 1 (o_proj 7.0 -> 6.6 us on the same box)
    // more than four rows (tools/sweep_gemv.py --rows 8, profiles/r03_sweep_rows.jsonl): 8 waves (o_proj Qwen3-4B 9.27 -> 7.88 us at 8 rows,
    // Llama-3-8B 9.55 -> 8.28).  At 2..4 rows the one-row shape stays ahead on the build without packed-FP32 ops (re-swept in round 4,
    // profiles/r04_sweep_rows*.jsonl: Qwen3-4B o_proj 6.73 -> 6.22 us at 2 rows, 6.83 -> 6.38 at 4; Llama-3-8B 7.02 -> 6.26, 7.18 -> 6.49;
    // round 3's 2-tile x 8-wave choice for them dated from the build with packed ops)
    if (rows > 4) waves = 8;
    // (very narrow outputs -- the un-merged k_proj / v_proj of an HF module tree, 4096 -> 1024: 64 one-tile blocks x 4 K-slices are 256
    // workgroups, one round; 4-tile blocks leave 64: 4.65 -> 4.43 us, profiles/r04_sweep_shape_4096x1024.jsonl)
    else if (rows == 1 && tiles <= 64) tpw = 1;
    // (one K-slice fewer when that brings (column blocks x slices) down to one round of workgroups: Qwen3.5-27B-class out_proj 6144 -> 5120
    // and the TP = 4 down shard 4352 -> 5120, 80 blocks x 4 = 320 -> x 3 = 240: 7.63 -> 6.73 us, 7.12 -> 6.11; profiles/r04_sweep_qwen3.5-27b-class_tp*.jsonl)
    if (rows == 1 && tpw == 4) {
      const int64_t cbs = (tiles + 3) / 4;
      if (cbs * 4 > 256 && cbs * 3 <= 256) ksplit = 3;
    }
  } else if (auto_tpw && auto_ks && auto_wv && narrow && G >= 16 && rows <= 4) {
    // small models' o / down (Qwen3-0.6B: 2048 -> 1024, 3072 -> 1024): 2 K-splits of 8-wave workgroups (down 4.81 -> 4.36 us).
    // Re-swept in round 4 on the build without packed-FP32 ops (profiles/r04_sweep_qwen3-0.6b.jsonl): below 24 groups the in-launch
    // hand-off no longer pays -- o_proj (16 groups) unsplit 3.89 us against 4.20 -- unless the splits leave partial sums (`deferred`)
    tpw = 1; ksplit = (G >= 24 || deferred) ? 2 : 1; waves = 8;
  } else if (auto_tpw && auto_ks && auto_wv && !deferred && rows == 1 && tiles > 512 && tiles < 1024 && G >= 16 && G < 64) {
    // almost-wide merged projections (the Qwen3.5 family's in_proj_qkvz 768 tiles, gated qkv 640): 4-tile blocks -- 160..192 of them --
    // unsplit below 32 groups (4B-class, K = 2560: 7.39 -> 6.10 us and 7.28 -> 5.99 against 2-tile blocks), four K-slices of four waves
    // from 32 groups on (9B, K = 4096: 8.91 -> 8.08 and 8.86 -> 7.93 against 2 slices of eight); profiles/r04_sweep_qwen3.5-*.jsonl.
    // The pattern behind the unsplit choices: the fewest tiles per wave that still give at most 256 column blocks -- ONE round of
    // workgroups over the 256 CUs (384 tiles: 2 -> 192 blocks; 512: 2 -> 256; 640 / 768: 4 -> 160 / 192; 1216 / 1792: 8 -> 152 / 224).
    // Exactly 512 tiles stay on 2-tile blocks (Llama-3-70B TP = 4 o_proj 2048 -> 8192: 4.67 us against 5.32 on 4-tile blocks)
    tpw = 4;
    if (G >= 32 && tiles >= 832) { tpw = 8; ksplit = 2; waves = 8; }     // 27B-class gated qkv (896 tiles, 40 groups): 112 blocks x 2 = 224, one round: 11.66 -> 9.93 us
    else if (G >= 32) { ksplit = 4; waves = 4; } else { ksplit = 1; waves = 8; }
  } else if (auto_tpw && auto_ks && auto_wv && tiles < 1024 && G >= ((deferred && rows == 1) || rows > 4 ? 16 : 32)) {
    // mid-width outputs with K >= 4096 (Llama-3-8B qkv: 384 tiles x 32 groups): 96 fat column blocks x 2 splits halve
    // the replicated rotation; pays since the reducer polls all splits at once (7.10 -> 6.71 us; at G = 20, Qwen3-4B
    // qkv, the unsplit 2-tile shape stays ahead: 5.65 vs 5.94).  `deferred` (the splits leave partial sums, nobody polls):
    // pays from G = 16 (Qwen3-4B qkv 5.50 -> 4.96 us, Llama-3-8B 6.78 -> 5.73, profiles/r03_sweep_noreducer_wide.jsonl); so it does at
    // 5..8 rows, where the replicated rotation has grown with the rows (Qwen3-4B qkv at 8 rows 9.58 -> 8.35, r03_sweep_rows.jsonl)
    tpw = 4; ksplit = 2; waves = 8;
  }
  if (tpw <= 0) {
    if (tiles >= 1024)
      tpw = rows > 8 ? 4 : 8;
    else if (tiles >= 768 && G >= 64 && rows <= 4)
      // a caller that fixes ksplit (the RMSNorm prologue: 1) on a deep-K, almost-wide output -- TP = 4 gate_up,
      // 8192 -> 14336: unsplit 4 x 16 waves 15.0 us, 2 x 16 waves 18.7 (profiles/r02_sweep_llama3-70b_tp4.jsonl;
      // the fused launch read 19.0 with the 2-tile shape, profiles/r02_fused_tp4.jsonl)
      tpw = 4;
    else if (tiles > 512 && rows == 1)
      tpw = 4;      // (a caller that fixes ksplit = 1 on an almost-wide output: the same 4-tile blocks as the automatic shape above)
    else if (tiles >= 320)
      tpw = 2;
    else
      tpw = 1;
  }
  if (auto_tpw && auto_ks && auto_wv && !deferred && rows == 1 && tpw == 8 && ksplit <= 0) {
    // wide outputs on 8-tile blocks (profiles/r04_sweep_qwen3.5-27b-class_tp1.jsonl), the one-round pattern again:
    const int64_t cbs = (tiles + 7) / 8;
    if (cbs * 2 <= 256 && G >= 32) ksplit = 2;                 // at most 128 blocks: two K-slices fill the round (in_proj_qkvz 5120 -> 16384: 12.82 -> 10.54 us)
    else if (cbs > 256 && cbs <= 320) tpw = 4;                 // a thin second round of fat blocks (gate_up 5120 -> 34816, 272 blocks: 25.3 -> 22.3 us)
  }
  if (ksplit <= 0) ksplit = 1;
  if (waves <= 0) {
    if (tpw == 4 && tiles >= 768 && tiles < 1024 && G >= 64 && rows <= 4)
      waves = 16;
    else if (rows > 4 || tpw > 2 || G < 16)
      // (eight groups or fewer per workgroup at up to four rows: four waves run two units each instead of eight running one -- Qwen3-0.6B
      // qkv 3.42 -> 3.29 us, gate_up 3.84 -> 3.43, profiles/r04_sweep_qwen3-0.6b.jsonl)
      waves = (G > 8 || (G == 8 && (rows > 4 || tpw > 2))) ? 8 : 4;     // (9..15 groups: eight waves -- 27B-class TP = 4 out_proj, 12 groups: 4.57 -> 4.04 us)
    else if (rows == 1 && tpw == 2 && ksplit == 1 && G <= 24)
      // 17..24 groups on 2-tile blocks (Qwen3-4B qkv, 2560 -> 6144): sixteen waves leave most of them ONE unit -- no tile request in flight
      // behind the rotation -- eight waves run two or three (5.41 -> 5.16 us on the build without packed-FP32 ops,
      // profiles/r04_sweep_qwen3-4b.jsonl; with them the two were level)
      waves = 8;
    else
      waves = 16;
  }
  if (ksplit > G) ksplit = G;
  if (ksplit > kMaxKsplit) ksplit = kMaxKsplit;
  // An in-launch K-split needs its whole grid resident at once (reducers spin on partials of the same launch):
  // shrink an AUTOMATIC split until column blocks x splits fits 256 CUs x the workgroups a CU holds at <= 128
  // VGPRs (16 waves); the launcher re-checks against the real occupancy of the instantiation.
  if (auto_ks && ksplit > 1) {
    const int64_t cbs = (tiles + tpw - 1) / tpw;
    const int64_t cap = 256 * (16 / (waves > 0 ? waves : 8));
    while (ksplit > 1 && cbs * ksplit > cap) --ksplit;
  }
}

}  // namespace paro

namespace paro {
// The launch shape a GEMV call ends up with: caller's knobs (0 = auto, mode -1 = auto) -> final values.
int resolve_launch_shape(const paro_linear_t* L, int64_t rows, int& tpw, int& ksp, int& wv, int& mode, bool deferred = false) {
  // a shape measured for this layer at load time (paro_linear_t.launch_hint, PackedParoWeights.autotune): one-row launches whose
  // knobs are all auto take it instead of the rules below; everything after this point treats it like explicit knobs
  if (rows == 1 && !deferred && tpw == 0 && ksp == 0 && wv == 0 && L->launch_hint != 0) {
    tpw = L->launch_hint & 0xff;
    ksp = (L->launch_hint >> 8) & 0xff;
    wv = (L->launch_hint >> 16) & 0xff;
    if (tpw != 1 && tpw != 2 && tpw != 4 && tpw != 8) tpw = 0;
    if (ksp > kMaxKsplit) ksp = 0;
    if (wv != 4 && wv != 8 && wv != 16) wv = 0;
  }
  const int waves_in = wv;
  if (tpw != 0 && tpw != 1 && tpw != 2 && tpw != 4 && tpw != 8)
    return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave must be 0 (auto), 1, 2, 4 or 8 (got %d)", tpw);
  if (ksp < 0 || ksp > kMaxKsplit) return fail(PARO_ERR_INVALID, "ksplit must be in 0..%d (got %d)", kMaxKsplit, ksp);
  const bool mode_auto = mode < 0;
  if (mode_auto) mode = 0;
  if (mode != 0 && mode != 1 && mode != 2 && mode != 3)
    return fail(PARO_ERR_INVALID, "mode must be -1 (auto), 0 (rotation inside every workgroup), 1 (rotate pre-pass), 2 (x is already rotated) or 3 (rotation shared inside the launch)");
  if (mode != 2 && (L->krot > 8 || rows > 16)) mode = 1;  // the packed schedule holds 8 stages; 17..64 rows exist pre-rotated only
  // The fused rotation is replicated in every workgroup and its cost grows with the rows: beyond 8 rows,
  // and from 5 rows on for merged projections (one replicated rotation PER partition), rotating once up
  // front with the stage kernel is cheaper (measured, Llama-3-8B shapes, us fused / pre-pass:
  // M=4 o 8.1/12.4 qkv 10.0/13.1; M=6 o 10.5/12.5 down 17.3/17.5 but qkv 15.2/13.5 gate_up 22.0/20.9;
  // M=16 gate_up 53/24).
  // (re-measured with the round-2 prologue, Llama-3-8B, us fused / pre-pass: M=6 qkv 11.3 / 13.2, gate_up 19.9 / 20.0;
  // M=8 qkv 11.4 / 13.4, o 10.1 / 13.4, gate_up 21.5 / 20.8, down 17.5 / 16.8; M=12 qkv 16.7 / 14.9, gate_up 47 / 22:
  // below 9 rows only the wide merged projections still prefer the pre-pass)
  // (round 4, profiles/r04_sweep_rows16_*.jsonl: a NARROW single-partition output with K < 8192 -- o_proj: 40 column blocks x 4 K-slices --
  // replicates so little rotation that the fused form stays ahead up to 16 rows: 11.4 us against 13.5 / 13.7 with the pre-pass)
  // (round 6, profiles/r06_rot_modes_sweep.jsonl -- Qwen3-4B, Llama-3-8B, Qwen3-0.6B shapes at 2..16 rows, us fused / pre-pass: at 5..8 rows
  // the fused form now wins EVERYWHERE, the wide merged projections included -- Qwen3-4B gate_up 11.7 / 14.7, Llama-3-8B gate_up 17.7 /
  // 19.5: round 2's rule for them dated from the kernels with packed-FP32 ops -- and at 9..16 rows it still wins where a workgroup rotates
  // few groups for few columns: up to 24 groups below 1024 tiles -- Qwen3-4B qkv 12.6 / 13.6, Qwen3-0.6B qkv 7.7 / 9.9, gate_up 8.2 / 10.2;
  // deep K and wide outputs keep the pre-pass there: Qwen3-4B gate_up 29.6 / 16.6, down 17.8 / 15.3, Llama-3-8B qkv 15.2 / 14.2)
  const bool narrow_single = L->n_parts == 1 && L->N / 16 <= 320 && L->K / 128 < 64;
  const bool few_groups = L->K / 128 <= 24 && L->N / 16 < 1024;
  if (mode_auto && rows > 8 && !narrow_single && !few_groups) mode = 1;
  // Round 6: the rotation shared inside the launch (mode 3: every (partition, group, row quad) rotated ONCE per launch by one wave of the
  // grid, handed over as {launch tag, two channels} granules) removes the replicated rotation without a second launch -- the GEMV on
  // rotated activations is flat in the rows (Qwen3-4B layer at 8 rows 26.3 us against 25.5 at one row, profiles/r06_prerot_rows.jsonl),
  // ALL of the batched-decode overhead is rotation.  PARO_SHARED_ROT_MIN_ROWS = first row count that takes it (plain calls; 17 = never).
  static const int shr_min = getenv("PARO_SHARED_ROT_MIN_ROWS") ? atoi(getenv("PARO_SHARED_ROT_MIN_ROWS")) : kSharedRotMinRows;
  const bool want_shared = mode_auto && !deferred && L->krot <= 8 && rows <= 16 && (rows >= shr_min || (rows >= 2 && shr_min <= 16));   // (2..4 rows: only as the hybrid, below)
  const int tpw_in = tpw, ksp_in = ksp;
  gemv_autotune(L, rows, tpw, ksp, wv, deferred);
  // 9..16 rows: 4 tiles per wave (the accumulators of 16 rows x 8 tiles do not fit beside a replicated rotation); mode 3 runs 8 tiles
  // there (no rotation state in a consumer) when that is what brings the grid onto the chip at once -- wide outputs: gate_up
  if (rows > 8 && rows <= 16 && tpw > 4) tpw = 4;
  (void)waves_in;
  if (tpw == 8 && wv == 16) wv = 8;   // 16 waves x 8 tiles does not fit the 128-VGPR budget
  if (quant_group(L->group_size) == 64) {   // group_size 64 instantiations: 1 / 2 / 4 / 8 tiles, 4 or 8 waves
    if (wv == 16) wv = 8;
  }
  const int G = (int)(L->K / 128);
  const int gps = (G + ksp - 1) / ksp;
  ksp = (G + gps - 1) / gps;           // empty splits are dropped
  if (want_shared) {
    // mode 3 only where producers + column blocks x K-slices fit the chip at once (the launcher checks the real occupancy and falls
    // back to the replicated rotation): wide outputs at 9..16 rows run 8-tile blocks for it, or keep the pre-pass
    const int64_t units = (int64_t)L->n_parts * G * shared_rot_row_tasks(rows);
    auto fits_ks = [&](int t, int ks) {
      int64_t cbs = 0;
      for (int i = 0; i < L->n_parts; ++i) cbs += (L->part_cols[i] / 16 + t - 1) / t;
      return shared_rot_prod_wgs(units, wv, cbs * ks, shared_rot_cap(rows, wv, t)) > 0;
    };
    auto fits = [&](int t) { return fits_ks(t, ksp); };
    // 9..16 rows, launch shapes re-swept UNDER mode 3 (profiles/r06_sweep_mode3.jsonl): with the rotation no longer replicated per workgroup,
    // thin column blocks win on outputs below 1024 tiles that are not deep-K -- more workgroups in flight, and the 2-tile build fits two per
    // CU -- and a mid-width merged projection of < 32 groups needs no K-split at all (Qwen3-4B qkv 9.3 -> 8.0 us at 16 rows, o 8.2 -> 7.7;
    // Llama-3-8B o 9.2 -> 8.8, qkv 11.1 -> 10.7; deep K keeps 4-tile blocks: Llama-3-8B down 16.0 against 18.2)
    bool thin = false;
    if (rows > 8 && rows >= shr_min && tpw_in == 0 && ksp_in == 0 && tpw == 4 && G < 64 && L->N / 16 < 1024) {
      int ks2 = (L->n_parts > 1 && G < 32) ? 1 : ksp;
      if (!fits_ks(2, ks2) && ks2 >= 3) ks2 -= 1;                      // (Llama-3-8B o_proj: 128 blocks x 4 slices do not fit, x 3 do)
      if (fits_ks(2, ks2)) {
        const int gps2 = (G + ks2 - 1) / ks2;
        tpw = 2; ksp = (G + gps2 - 1) / gps2; mode = 3; thin = true;
      }
    }
    if (thin) {
    } else if (rows < shr_min) {
      if (shared_rot_self(L, rows, ksp, wv) == 1 && fits(tpw)) mode = 3;
    } else if (fits(tpw)) mode = 3;
    else if (tpw_in == 0 && rows > 8 && tpw == 4 && L->N / 16 >= 1024 && wv <= 8 && fits(8)) { tpw = 8; mode = 3; }
    else if (ksp_in == 0 && rows > 8 && ksp >= 3 && fits_ks(tpw, ksp - 1)) {
      // 9..16 rows, one workgroup per CU: a K-split whose slices fill the chip leaves no room for the producers -- one slice fewer does
      // (Llama-3-8B o_proj 64 x 4 -> 64 x 3 + 32, down_proj + 64)
      ksp -= 1;
      const int gps2 = (G + ksp - 1) / ksp;
      ksp = (G + gps2 - 1) / gps2;
      mode = 3;
    }
  }
  return PARO_OK;
}
}  // namespace paro

extern "C" int paro_gemv_launch_shape(const paro_linear_t* L, int64_t rows, int* tiles_per_wave, int* ksplit, int* waves,
                                      int* mode) {
  using namespace paro;
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows < 1 || rows > 64) return fail(PARO_ERR_INVALID, "paro_w4a16_gemv handles 1..64 rows (got %lld)", (long long)rows);
  if (!tiles_per_wave || !ksplit || !waves || !mode) return fail(PARO_ERR_INVALID, "null pointer");
  return resolve_launch_shape(L, rows, *tiles_per_wave, *ksplit, *waves, *mode);
}

extern "C" int64_t paro_linear_workspace_bytes(const paro_linear_t* L, int64_t rows) {
  using namespace paro;
  if (validate_linear(L) != PARO_OK) return -1;
  if (rows < 0) return -1;
  const int64_t r = rows < 1 ? 1 : rows;
  // <= 16 rows: as {two channels, tag} granules, 16 bytes per lane and row PAIR (mode 3), or one 16-row tile in fragment order (mode 1);
  // 17..64 rows: fragment order pads to 32 / 64 rows (mode 1); above: plain rotated rows (GEMM path)
  const int64_t xrot = r <= 16 ? (int64_t)L->n_parts * L->K * std::max<int64_t>(((r + 1) / 2) * 8, 32)
                               : (int64_t)L->n_parts * (r <= 32 ? 32 : (r <= 64 ? 64 : r)) * L->K * 2;
  // 8-byte {tag, partial} granules of the GEMV K-split: any split up to kMaxKsplit for <= 16 rows (the
  // launch-shape knobs are the caller's), the automatic one for 17..64 rows
  int64_t slabs = 0;
  if (r <= 16) {
    slabs = (int64_t)kMaxKsplit * r * L->N * 8;
  } else if (r <= 64) {
    int tpw = 0, ks = 0, wv = 0;
    gemv_autotune(L, r, tpw, ks, wv);
    slabs = (int64_t)(ks > 1 ? ks : 1) * r * L->N * 8;
    if (slabs < (int64_t)kMaxKsplit * 16 * L->N * 8) slabs = (int64_t)kMaxKsplit * 16 * L->N * 8;   // two 16-row passes
  }
  int gks = gemm_ksplit(L, r);                                            // fp32 partial tiles of the small-M GEMM
  if (gemm4_ksplit(L, r) > gks) gks = gemm4_ksplit(L, r);                  // (either kernel may be the one that runs)
  const int64_t partial = gks > 1 ? 256 + (int64_t)gks * r * L->N * 4 : 0;
  return PARO_WS_COUNTER_BYTES + slabs + xrot + partial;
}

namespace paro {
static int gemv_impl(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                     int64_t workspace_bytes, int tiles_per_wave, int ksplit, int waves, int mode,
                     const paro_fusion_t* F, const paro_experts_t* E, void* stream) {
  int rc = validate_linear(L);
  if (rc != PARO_OK) return rc;
  if (rows == 0) return PARO_OK;
  if (rows < 0 || rows > 64) return fail(PARO_ERR_INVALID, "paro_w4a16_gemv handles 1..64 rows (got %lld)", (long long)rows);
  const bool pout = F && F->parts_out;   // deferred K-split reduction (v12): this launch leaves / receives fp32 partial sums
  const bool pin = F && F->parts_in;
  const bool ain = F && F->attn_in;      // x = the slots of a split attention launch (v14)
  if ((!x && !ain) || (!y && !pout)) return fail(PARO_ERR_INVALID, "null pointer");
  const bool ar = F && F->ar_peers && F->ar_world >= 1;
  const bool fused = (F && (F->prologue != PARO_PROLOGUE_NONE || F->residual)) || E || ar || pin || ain;
  static const paro_fusion_t no_fusion = {PARO_PROLOGUE_NONE, 0.f, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
  int attn_shift = 0;
  if (ain) {
    if (rows != 1) return fail(PARO_ERR_UNSUPPORTED, "attn_in is a batch-1 decode path (got %lld rows)", (long long)rows);
    if (E || ar || pin) return fail(PARO_ERR_UNSUPPORTED, "attn_in cannot be combined with expert slots, the all-reduce epilogue or parts_in");
    if (F->prologue != PARO_PROLOGUE_NONE || F->residual) return fail(PARO_ERR_UNSUPPORTED, "attn_in feeds the plain linear (no prologue, no residual: leave partial sums or add it downstream)");
    if (L->krot > 8 || mode == 1 || mode == 2) return fail(PARO_ERR_UNSUPPORTED, "attn_in needs the in-kernel rotation (krot <= 8, mode 0)");
    attn_shift = F->attn_head_dim == 64 ? 6 : (F->attn_head_dim == 128 ? 7 : (F->attn_head_dim == 256 ? 8 : 0));
    if (!attn_shift || L->K % F->attn_head_dim != 0) return fail(PARO_ERR_INVALID, "attn_head_dim must be 64, 128 or 256 and divide K (got %d, K = %lld)", F->attn_head_dim, (long long)L->K);
    if (((uintptr_t)F->attn_in & 15) != 0) return fail(PARO_ERR_INVALID, "attn_in must be 16-byte aligned");
  }
  const paro_attn_tail_t* T = F ? F->attn_tail : nullptr;   // v18: the consuming decode attention runs in this launch
  if (T) {
    if (rows != 1 || !pout || F->prologue != PARO_PROLOGUE_RMSNORM || E || ar || ain)
      return fail(PARO_ERR_UNSUPPORTED, "attn_tail rides the one-row RMSNorm-prologue projection that leaves partial sums (no expert slots, all-reduce, attn_in)");
    if (!T->kcache || !T->vcache || !T->attn_parts || !T->pos || !T->rope || !T->workspace) return fail(PARO_ERR_INVALID, "attn_tail: null pointer");
    if (T->head_dim != 128) return fail(PARO_ERR_UNSUPPORTED, "attn_tail: head_dim must be 128 (got %d)", T->head_dim);
    if (T->n_heads < 1 || T->n_kv_heads < 1 || T->n_kv_heads > 64 || T->n_heads % T->n_kv_heads != 0 || T->n_heads / T->n_kv_heads > 4)
      return fail(PARO_ERR_UNSUPPORTED, "attn_tail: n_heads must be 1..4 x n_kv_heads (got %d / %d)", T->n_heads, T->n_kv_heads);
    if ((int64_t)(T->n_heads + 2 * T->n_kv_heads) * T->head_dim != L->N) return fail(PARO_ERR_INVALID, "attn_tail: N = %lld is not (n_heads + 2 n_kv_heads) * head_dim", (long long)L->N);
    if ((T->q_norm_w == nullptr) != (T->k_norm_w == nullptr)) return fail(PARO_ERR_INVALID, "attn_tail: q / k norm weights come together");
    if (T->max_positions < 8 || T->max_positions % 8 != 0) return fail(PARO_ERR_INVALID, "attn_tail: max_positions must be a multiple of 8 (got %d)", T->max_positions);
    if (T->workspace_bytes < paro_attn_decode_workspace_bytes(T->n_heads, T->n_kv_heads, T->head_dim, T->max_positions))
      return fail(PARO_ERR_INVALID, "attn_tail: attention workspace too small");
    if (quant_group(L->group_size) != 128) return fail(PARO_ERR_UNSUPPORTED, "attn_tail: group_size 128 only");
    if (((uintptr_t)F->parts_out & 15) != 0) return fail(PARO_ERR_INVALID, "attn_tail: parts_out must be 16-byte aligned");
  }
  if (pout || pin) {
    if (rows != 1) return fail(PARO_ERR_UNSUPPORTED, "partial sums (parts_out / parts_in) are a batch-1 decode path (got %lld rows)", (long long)rows);
    if (E || ar) return fail(PARO_ERR_UNSUPPORTED, "partial sums are not defined for expert slots or the all-reduce epilogue");
    if (L->krot > 8 || mode == 1 || mode == 2) return fail(PARO_ERR_UNSUPPORTED, "partial sums need the in-kernel rotation (krot <= 8, mode 0)");
  }
  if (pout) {
    if (F->parts_out_n < 2 || F->parts_out_n > PARO_MAX_PARTIALS) return fail(PARO_ERR_INVALID, "parts_out_n must be in 2..%d (got %d): a launch that does not split K writes y itself", PARO_MAX_PARTIALS, F->parts_out_n);
    if (F->residual || L->bias) return fail(PARO_ERR_UNSUPPORTED, "parts_out: residual and bias are added where the partial sums are completed; this launch takes neither");
    if (ksplit != 0 && ksplit != F->parts_out_n) return fail(PARO_ERR_INVALID, "ksplit %d contradicts parts_out_n %d", ksplit, F->parts_out_n);
  }
  if (pin) {
    if (F->prologue != PARO_PROLOGUE_NONE && F->prologue != PARO_PROLOGUE_RMSNORM) return fail(PARO_ERR_UNSUPPORTED, "parts_in feeds the plain or the RMSNorm prologue");
    if (F->x_out && (F->x_out == x || F->x_out == (const void*)F->parts_in || F->x_out == (void*)F->parts_out || F->x_out == y))
      return fail(PARO_ERR_INVALID, "x_out must not alias x, parts_in, parts_out or y (other workgroups still read / write them)");
  }
  if (E) {
    if (!F) F = &no_fusion;
    if (!E->expert_idx || E->n_slots < 1 || E->n_slots > 65535) return fail(PARO_ERR_INVALID, "bad expert slot table");
    if (E->x_slot_div < 1) return fail(PARO_ERR_INVALID, "x_slot_div must be >= 1");
    if (E->n_experts < 1) return fail(PARO_ERR_INVALID, "n_experts must be >= 1 (the device checks every slot's id against it)");
    if (F->residual) return fail(PARO_ERR_UNSUPPORTED, "the residual epilogue is not defined for expert slots");
    if (ksplit > 1) return fail(PARO_ERR_INVALID, "expert launches do not K-split (the slots already fill the grid)");
    ksplit = 1;
  }
  if (fused) {
    if (F->prologue < PARO_PROLOGUE_NONE || F->prologue > PARO_PROLOGUE_GELU_TANH_MUL) return fail(PARO_ERR_INVALID, "unknown prologue %d", F->prologue);
    if (rows > 4) return fail(PARO_ERR_UNSUPPORTED, "fused prologue / epilogue is a decode path: at most 4 rows (got %lld)", (long long)rows);
    if (L->krot > 8) return fail(PARO_ERR_UNSUPPORTED, "fused prologue / epilogue needs the in-kernel rotation (krot <= 8)");
    if (mode == 1 || mode == 2) return fail(PARO_ERR_INVALID, "fused prologue / epilogue needs mode 0 (in-kernel rotation)");
    mode = 0;
    // sum(x^2) of the RMSNorm prologue is collected per workgroup: every workgroup must cover all of K
    // ... unless the launch leaves partial sums (parts_out): the partial sums of squares travel with them (row N) and the consumer scales
    if (F->prologue == PARO_PROLOGUE_RMSNORM && !pout) {
      if (ksplit > 1) return fail(PARO_ERR_INVALID, "the RMSNorm prologue cannot be combined with a K-split");
      ksplit = 1;
    }
    if (ar) {
      if (E) return fail(PARO_ERR_UNSUPPORTED, "the all-reduce epilogue is not defined for expert slots");
      if (rows != 1) return fail(PARO_ERR_UNSUPPORTED, "the all-reduce epilogue is a batch-1 decode path (got %lld rows)", (long long)rows);
      if (F->prologue == PARO_PROLOGUE_RMSNORM) return fail(PARO_ERR_INVALID, "the RMSNorm prologue cannot feed a row-parallel shard (a norm over a K slice is not the layer's norm)");
      if (F->ar_world > kArMaxWorld || F->ar_rank < 0 || F->ar_rank >= F->ar_world) return fail(PARO_ERR_INVALID, "bad all-reduce world / rank (%d / %d)", F->ar_world, F->ar_rank);
      if (!F->ar_own || !F->ar_state) return fail(PARO_ERR_INVALID, "all-reduce epilogue: ar_own / ar_state is null");
      if (F->ar_peers[F->ar_rank] != F->ar_own) return fail(PARO_ERR_INVALID, "all-reduce epilogue: ar_peers[ar_rank] must be ar_own");
      if (L->N > F->ar_max_elems) return fail(PARO_ERR_INVALID, "all-reduce buffers sized for %lld elements, the layer has %lld outputs", (long long)F->ar_max_elems, (long long)L->N);
    }
    const int64_t min_stride = (F->prologue >= PARO_PROLOGUE_SILU_MUL ? 2 : 1) * L->K;
    if (F->x_stride != 0 && F->x_stride < min_stride) return fail(PARO_ERR_INVALID, "x_stride %lld < %lld", (long long)F->x_stride, (long long)min_stride);
  }
  if (mode == 3 && (fused || pout)) return fail(PARO_ERR_UNSUPPORTED, "mode 3 (shared rotation) runs the plain linear: no prologue / epilogue fusion, no partial sums");
  int tpw = tiles_per_wave, ksp = ksplit, wv = waves;
  if (E && tpw == 0 && wv == 0) {
    // expert slots: the grid is (column blocks) x (slots), so the slots fill the chip and fat column blocks cut the rotation every
    // workgroup repeats: 4 tiles per wave, 8 waves from 16 groups of K on, else 4 (64 experts, top-8, 2048 / 768: one token 18.7 ->
    // ~11.6 us per MoE block, four tokens 59.4 -> ~23; profiles/r03_moe_decode_shapes.txt)
    const int64_t tiles_min = L->part_cols[0] / 16;
    tpw = tiles_min >= 4 ? 4 : (tiles_min >= 2 ? 2 : 1);
    wv = L->K / 128 >= 16 ? 8 : 4;
  }
  rc = resolve_launch_shape(L, rows, tpw, ksp, wv, mode, pout);
  if (rc != PARO_OK) return rc;
  if (pout && ksp != F->parts_out_n) {
    // the automatic shape splits K differently from what the caller sized its buffers for: the caller's count wins
    tpw = tiles_per_wave; ksp = F->parts_out_n; wv = waves;
    rc = resolve_launch_shape(L, rows, tpw, ksp, wv, mode);
    if (rc != PARO_OK) return rc;
    if (ksp != F->parts_out_n) return fail(PARO_ERR_INVALID, "parts_out_n = %d: K = %lld splits into %d non-empty slices", F->parts_out_n, (long long)L->K, ksp);
  }
  if (fused) {
    if (tpw == 8 && wv == 16) wv = 8;
  }
  const int G = (int)(L->K / 128);
  hipStream_t st = (hipStream_t)stream;

  GemvArgs a;
  PartTable pt;
  if (!fill_part_table(pt, L->n_parts, L->part_cols, tpw)) return fail(PARO_ERR_INVALID, "bad partition table");
  static const int env_pd = getenv("PARO_GEMV_PD") ? atoi(getenv("PARO_GEMV_PD")) : 0;
  static const int env_skew = getenv("PARO_GEMV_SKEW") ? atoi(getenv("PARO_GEMV_SKEW")) : 1;
  static const int env_prio = getenv("PARO_GEMV_PRIO") ? atoi(getenv("PARO_GEMV_PRIO")) : 1;
  a.hot.wq = (const u32x4*)L->wq;
  a.hot.sz = (const unsigned*)L->sz;
  a.hot.rot = (const unsigned*)L->rot;
  a.hot.cs = (const unsigned short*)L->channel_scales;
  a.hot.x = (const unsigned short*)x;
  {
    const unsigned long long rp = fused ? (unsigned long long)(uintptr_t)F->residual : 0ull;
    a.hot.residual_lo = (unsigned)rp;
    a.hot.residual_hi = (unsigned)(rp >> 32);
  }
  a.eps = fused ? F->eps : 0.f;
  a.bias = (const unsigned short*)L->bias;
  a.y = (unsigned short*)y;
  a.rows = (int)rows;
  a.qs = 128 / quant_group(L->group_size);
  int gps = (G + ksp - 1) / ksp;
  a.ksplit = (G + gps - 1) / gps;  // drop empty splits
  a.slabs = nullptr;
  a.counters = nullptr;
  a.prologue = fused ? F->prologue : PARO_PROLOGUE_NONE;
  a.parts_out = pout ? 1 : 0;
  a.parts_in = pin ? F->parts_in : nullptr;
  a.x_out = pin ? (unsigned short*)F->x_out : nullptr;
  a.attn_in = ain ? 1 : 0;
  if (ain) {   // the same 16-byte argument pair: the slots' outputs [K][4], the slots' (max, sum) [K / head_dim][8] | log2(head_dim)
    a.parts_in = F->attn_in;
    a.x_out = (unsigned short*)((uintptr_t)(F->attn_in + (int64_t)L->K * 4) | (uintptr_t)attn_shift);
  }
  const long long xstride = (fused && F->x_stride != 0) ? F->x_stride : (int64_t)L->K * ((fused && F->prologue >= PARO_PROLOGUE_SILU_MUL) ? 2 : 1);
  a.expert_idx = E ? E->expert_idx : nullptr;
  a.wq_estride = E ? E->wq_stride_bytes : 0;
  a.sz_estride = E ? E->sz_stride_bytes : 0;
  a.x_sstride = E ? E->x_slot_stride : 0;
  a.y_sstride = E ? E->y_slot_stride : 0;
  a.x_div = E ? E->x_slot_div : 1;
  a.n_experts = E ? E->n_experts : 0;
  for (int r = 0; r < kArMaxWorld; ++r) a.ar_peer[r] = (ar && r < F->ar_world) ? (unsigned char*)F->ar_peers[r] : nullptr;
  a.ar_mine = ar ? (unsigned char*)F->ar_own : nullptr;
  a.ar_state = ar ? (unsigned*)F->ar_state : nullptr;
  a.ar_world = ar ? F->ar_world : 0;
  a.ar_rank = ar ? F->ar_rank : 0;
  a.ar_slot = ar ? ar_slot_b(F->ar_max_elems) : 0;
  a.ar_off = ar ? ar_region_b_off(F->ar_world, F->ar_max_elems) : 0;
  a.attn_wgs = 0;
  a.attn_cbs = 1;
  if (T) {
    // the attention row: (KV head, 64-position chunk) workgroups -- the split launch's grid (attn.hip), flattened
    AttnArgs& t = a.attn;
    t.qkv = nullptr;
    t.qkv_parts = (const f32x4*)F->parts_out;       // (as granules: attn_impl.hpp, TAGGED)
    t.norm_dim = (float)L->K;
    t.norm_eps = F->eps;
    t.kcache = (unsigned short*)T->kcache;
    t.vcache = (unsigned short*)T->vcache;
    t.out = nullptr;
    t.pos = T->pos;
    t.rope = T->rope;
    t.qnw = (const unsigned short*)T->q_norm_w;
    t.knw = (const unsigned short*)T->k_norm_w;
    t.part = (float*)((char*)T->workspace + kAttnWsHeader);
    t.ticket = (unsigned*)T->workspace;
    t.split_o = T->attn_parts;
    t.split_ml = T->attn_parts + (int64_t)T->n_heads * T->head_dim * 4;
    t.eps = T->eps;
    t.scale = T->scale;
    t.Hq = T->n_heads;
    t.Hkv = T->n_kv_heads;
    t.hd = T->head_dim;
    t.T_max = T->max_positions;
    t.chunks = (T->max_positions + 63) / 64;
    t.dbg = 0;
    a.attn_wgs = t.Hkv * t.chunks;
    a.attn_cbs = (int)pt.cbs;
    if ((a.attn_wgs + a.attn_cbs - 1) / a.attn_cbs + a.ksplit > 65535) return fail(PARO_ERR_UNSUPPORTED, "attn_tail: too many attention workgroups for the grid");
    if (wv != 4 && wv != 8) return fail(PARO_ERR_UNSUPPORTED, "attn_tail: built for launch shapes of 4 or 8 waves (this one: %d)", wv);
  }
  a.pd = (env_pd == 11 || env_pd == 21 || env_pd == 31 || env_pd == 41 || env_pd == 51 || env_pd == 61 || env_pd == 71 || env_pd == 81 || env_pd == 99) ? env_pd : 1;
  // K-split reducer: its first poll goes out a few hundred cycles after its own partial sums are staged -- a poll that lands before the
  // other slices' granules costs a whole extra round trip (profiles/r06_poll_delay_sweep.jsonl, us at 0 / 256 / 512 cycles: Qwen3-4B o
  // 5.00 / 4.88 / 4.96, down 7.42 / 7.24 / 7.14; Llama-3-8B o 5.45 / 5.31 / 5.25, down 9.55 / 9.39 / 9.32).  PARO_POLL_DELAY overrides.
  static const int env_poll = getenv("PARO_POLL_DELAY") ? atoi(getenv("PARO_POLL_DELAY")) : -1;
  a.poll_delay = env_poll >= 0 ? env_poll : (gps >= 16 ? 8 : 4);
  // mode 1 on the packed schedule (krot <= 8): the schedule pre-pass (rotate.hip) -- one wave per (partition, group, 4 rows), the in-kernel
  // rotation's arithmetic (bit-identical to modes 0 / 3), x handed over in MFMA-fragment order.  PARO_PREROT_SCHED=0: the stage kernel and
  // plain rows, as up to round 5 (A/B; krot > 8 always).
  static const int env_sched = getenv("PARO_PREROT_SCHED") ? atoi(getenv("PARO_PREROT_SCHED")) : 1;
  const bool sched_prepass = mode == 1 && L->krot <= 8 && L->rot && env_sched != 0;
  auto repack_hot = [&]() {
    if (!pack_hot(a.hot, pt, G, L->wq_order, a.rows, L->krot, a.ksplit, gps, env_skew, env_prio, a.prologue, E != nullptr, xstride, pout)) return false;
    if (sched_prepass) a.hot.meta |= 1u << 30;   // x arrives in fragment order
    return true;
  };
  if (!repack_hot()) return fail(PARO_ERR_UNSUPPORTED, "layer too large for the 16-bit partition tables of the GEMV (N / 16 must stay below 65535)");

  const int64_t slab_bytes = (a.ksplit > 1 && !pout) ? (int64_t)(a.ksplit - 1) * rows * L->N * 8 : 0;
  const bool shared = mode == 3 && !fused;
  if (mode == 3 && !shared) mode = 0;
  const int64_t shr_bytes = (int64_t)L->n_parts * G * ((rows + 1) / 2) * 1024;   // granules: [partition][group][row pair][64 lanes] x 16 bytes
  // (fragment order pads the rows to whole 16-row tiles: [n_parts][G][row tiles][4][64] x 16 bytes)
  const int64_t xrot_bytes = mode == 1 ? (int64_t)L->n_parts * (sched_prepass ? (rows <= 16 ? 16 : (rows <= 32 ? 32 : 64)) : rows) * L->K * 2 : (shared ? shr_bytes : 0);
  a.shared_rot = shared ? 1 : 0;
  a.xg = nullptr;
  a.shr_self = shared ? shared_rot_self(L, rows, a.ksplit, wv) : 0;
  {
    // producer tasks: (partition, group that is no wave's own first group, chunk of rows); the pure form: every group
    const int gps_ = (G + a.ksplit - 1) / a.ksplit;
    const int l_full = std::max(0, gps_ - a.shr_self * wv), l_last = std::max(0, (G - (a.ksplit - 1) * gps_) - a.shr_self * wv);
    a.shr_units = L->n_parts * ((a.ksplit - 1) * l_full + l_last) * shared_rot_row_tasks(rows);
  }
  a.shr_prod_wgs = 0;                                                         // producer workgroups in front of every grid row
  if (shared) {
    const int pw = shared_rot_prod_wgs(a.shr_units, wv, (int64_t)pt.cbs * a.ksplit, shared_rot_cap(rows, wv, tpw));
    a.shr_prod_wgs = pw > 0 ? pw : (a.shr_units + wv - 1) / wv;               // (explicit mode 3 on a grid that may not fit: the launcher decides)
  }
  a.shr_bytes = (int)shr_bytes;
  if (shared && shr_bytes >= (1ll << 31)) return fail(PARO_ERR_UNSUPPORTED, "shared rotation: granule buffer too large");
  const int64_t need = PARO_WS_COUNTER_BYTES + slab_bytes + xrot_bytes;
  if (slab_bytes + xrot_bytes > 0) {
    if (!workspace || workspace_bytes < need)
      return fail(PARO_ERR_INVALID, "workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
    if ((int64_t)pt.cbs * 4 > PARO_WS_STATUS_OFFSET || pt.cbs > 4096) return fail(PARO_ERR_INVALID, "too many column blocks for the K-split epoch words");
    a.counters = (unsigned*)workspace;
    a.slabs = (unsigned long long*)((char*)workspace + PARO_WS_COUNTER_BYTES);
  }
  if (shared) a.xg = (unsigned long long*)((char*)workspace + PARO_WS_COUNTER_BYTES + slab_bytes);   // [n_parts][G][rows][64] granules
  if (pout) a.slabs = (unsigned long long*)F->parts_out;        // float [N][4]
  if (a.pd == 31 && a.ksplit == 1 && workspace && workspace_bytes >= PARO_WS_COUNTER_BYTES + (int64_t)pt.cbs * 640)
    a.slabs = (unsigned long long*)((char*)workspace + PARO_WS_COUNTER_BYTES);   // per-workgroup phase timestamps (diagnostic build)
  // mode 2: the caller hands over rotated activations, [n_parts][rows][K] in the activation type (whoever produced x
  // rotated it -- rotation::rotate per partition, or a producer kernel's epilogue): the pre-rotated kernels, no pre-pass
  if (mode == 1) {
    unsigned short* xrot = (unsigned short*)((char*)workspace + PARO_WS_COUNTER_BYTES + slab_bytes);
    // ONE launch rotates x with every merged partition's parameters (blockIdx.z = partition)
    if (sched_prepass)
      rc = launch_prerot_sched(x, xrot, L->rot, L->channel_scales, rows, L->K, L->krot, L->n_parts, L->act_dtype, rows <= 16 ? 1 : (rows <= 32 ? 2 : 4), st);
    else
      rc = launch_rotate(x, xrot, L->pairs, L->theta, L->channel_scales, rows, L->K, L->krot, 128, L->act_dtype,
                         PARO_DTYPE_F16, st, L->n_parts);
    if (rc != PARO_OK) return rc;
    a.hot.x = xrot;
  }
  dim3 grid((unsigned)(pt.cbs + a.shr_prod_wgs), (unsigned)(a.ksplit + (a.attn_wgs ? (a.attn_wgs + a.attn_cbs - 1) / a.attn_cbs : 0)), E ? (unsigned)E->n_slots : 1u);   // (+ the attention rows)
  typedef int (*launch_fn)(const GemvArgs&, int, dim3, hipStream_t);
  // [type][pre-rotated][tiles per wave - 1]: 1, 2, 4, 8 tiles per wave (the 3 / 5 / 6 / 7-tile builds of round 2 measured
  // within +-3 % of their neighbours on every shape and were dropped)
  static const launch_fn table[2][2][8] = {
      {{launch_gemv_f16_0_t1, launch_gemv_f16_0_t2, nullptr, launch_gemv_f16_0_t4, nullptr, nullptr, nullptr, launch_gemv_f16_0_t8},
       {launch_gemv_f16_1_t1, launch_gemv_f16_1_t2, nullptr, launch_gemv_f16_1_t4, nullptr, nullptr, nullptr, launch_gemv_f16_1_t8}},
      {{launch_gemv_bf16_0_t1, launch_gemv_bf16_0_t2, nullptr, launch_gemv_bf16_0_t4, nullptr, nullptr, nullptr, launch_gemv_bf16_0_t8},
       {launch_gemv_bf16_1_t1, launch_gemv_bf16_1_t2, nullptr, launch_gemv_bf16_1_t4, nullptr, nullptr, nullptr, launch_gemv_bf16_1_t8}}};
  const launch_fn fn = (tpw >= 1 && tpw <= 8) ? table[L->act_dtype == PARO_DTYPE_F16 ? 0 : 1][(mode == 1 || mode == 2) ? 1 : 0][tpw - 1] : nullptr;
  if (!fn) return fail(PARO_ERR_UNSUPPORTED, "tiles_per_wave = %d is not built for this mode", tpw);
  rc = fn(a, wv, grid, st);
  if (rc == PARO_ERR_NOT_RESIDENT && shared && tpw == 8 && rows > 8)
    // (8 tiles x 9..16 rows exist for the shared rotation only: the pre-pass route is what such a wide output takes otherwise)
    return gemv_impl(L, x, y, rows, workspace, workspace_bytes, tiles_per_wave, ksplit, waves, 1, F, E, stream);
  if (rc == PARO_ERR_NOT_RESIDENT && shared) {
    // the grid does not fit the chip at once (nobody may wait for a producer that is not running): the replicated rotation is always legal
    a.shared_rot = 0;
    a.shr_prod_wgs = 0;
    grid = dim3((unsigned)pt.cbs, (unsigned)a.ksplit, 1u);
    rc = fn(a, wv, grid, st);
  }
  if (rc == PARO_ERR_NOT_RESIDENT && ksplit == 0 && a.ksplit > 1 && !pout) {
    // the automatic K-split does not fit this instantiation's real occupancy: run unsplit (always legal)
    a.ksplit = 1;
    gps = G;
    repack_hot();
    rc = fn(a, wv, dim3((unsigned)(pt.cbs + a.shr_prod_wgs), 1), st);
  }
  if (rc == PARO_ERR_NOT_RESIDENT) rc = PARO_ERR_UNSUPPORTED;
  if (rc != PARO_OK) return rc;
  return check_launch("paro_w4a16_gemv");
}
}  // namespace paro

extern "C" int paro_w4a16_gemv(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                               int64_t workspace_bytes, int tiles_per_wave, int ksplit, int waves, int mode,
                               void* stream) {
  return paro::gemv_impl(L, x, y, rows, workspace, workspace_bytes, tiles_per_wave, ksplit, waves, mode, nullptr, nullptr, stream);
}

extern "C" int paro_w4a16_gemv_fused(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                                     int64_t workspace_bytes, const paro_fusion_t* fusion, void* stream) {
  return paro::gemv_impl(L, x, y, rows, workspace, workspace_bytes, 0, 0, 0, 0, fusion, nullptr, stream);
}

extern "C" int paro_w4a16_gemv_experts(const paro_linear_t* L, const void* x, void* y, int64_t rows, void* workspace,
                                       int64_t workspace_bytes, const paro_fusion_t* fusion, const paro_experts_t* experts,
                                       void* stream) {
  if (!experts) return paro::fail(PARO_ERR_INVALID, "null expert descriptor");
  return paro::gemv_impl(L, x, y, rows, workspace, workspace_bytes, 0, 0, 0, 0, fusion, experts, stream);
}

// ---- deferred K-split reduction: helpers (v12)
namespace paro {
template <typename AT>
__global__ void parts_finish_kernel(const unsigned short* __restrict__ x, const float* __restrict__ parts, int64_t K,
                                    unsigned short* __restrict__ out) {
  typedef Act<AT> A;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const f32x4 p = ((const f32x4*)parts)[k];             // slots in the in-launch reducer's order (last split first), unused slots zero
  float v = ((p[0] + p[1]) + p[2]) + p[3];
  if (x) v += A::to_f32(x[k]);
  out[k] = A::from_f32(v);
}
}  // namespace paro

extern "C" int paro_gemv_parts_count(const paro_linear_t* L) {
  using namespace paro;
  if (validate_linear(L) != PARO_OK) return -1;
  if (L->krot > 8 || L->bias) return 0;
  int tpw = 0, ks = 0, wv = 0, mode = 0;
  if (resolve_launch_shape(L, 1, tpw, ks, wv, mode, true) != PARO_OK) return -1;
  return (ks >= 2 && ks <= PARO_MAX_PARTIALS) ? ks : 0;
}

extern "C" int paro_attn_tail_supported(const paro_linear_t* L, int n_heads, int n_kv_heads, int head_dim, int max_positions) {
  using namespace paro;
  if (validate_linear(L) != PARO_OK) return -1;
  if (L->krot > 8 || L->bias || quant_group(L->group_size) != 128) return 0;
  if (head_dim != 128 || n_heads < 1 || n_kv_heads < 1 || n_kv_heads > 64 || n_heads % n_kv_heads != 0 || n_heads / n_kv_heads > 4) return 0;
  if ((int64_t)(n_heads + 2 * n_kv_heads) * head_dim != L->N || max_positions < 8 || max_positions % 8 != 0) return 0;
  int tpw = 0, ks = 0, wv = 0, mode = 0;
  if (resolve_launch_shape(L, 1, tpw, ks, wv, mode, true) != PARO_OK) return -1;
  if (ks < 2 || ks > PARO_MAX_PARTIALS || (wv != 4 && wv != 8) || !(tpw == 1 || tpw == 2 || tpw == 4 || tpw == 8)) return 0;
  if (tpw == 8 && wv == 16) return 0;
  return 1;
}

extern "C" int paro_parts_finish(const void* x, const float* parts, int64_t K, void* out, int act_dtype, void* stream) {
  using namespace paro;
  if (!parts || !out || K < 1) return fail(PARO_ERR_INVALID, "paro_parts_finish: bad arguments");
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  const dim3 grid((unsigned)((K + 255) / 256));
  if (act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(parts_finish_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, parts, K, (unsigned short*)out);
  else
    hipLaunchKernelGGL(parts_finish_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, parts, K, (unsigned short*)out);
  return check_launch("paro_parts_finish");
}

/* Experimental entry points of libparo_mi355x.so: built only by `make -C paroquant_amd/csrc EXPERIMENTAL=1`
 * (csrc/experimental/engine.hip, engine2.hip).  NOT part of the default library or of the default ABI (include/paro_abi.h, v17):
 * both persistent decode engines are parity-green and measured SLOWER than one launch per linear on every shape this repo times
 * (Qwen3-4B step: per-call 0.929 ms, engine 0.930, engine 2 1.204; profiles/NOTES.md 4.2 / 5.1 hold the per-edge timelines and the
 * post-mortem).  Nothing in the product reaches them; they stay in the tree for whoever continues the design.
 * Bindings: paroquant_amd/_native.py binds these symbols only when the loaded library exports them (_native.has_experimental()). */
#ifndef PARO_ABI_EXPERIMENTAL_H
#define PARO_ABI_EXPERIMENTAL_H
#include "paro_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 * Persistent decode engine (v13): a CHAIN of linears at batch 1 in ONE launch.  The reference issues `rotate -> INT4 GEMM` per
 * linear (transformers/modules.py:57-71, vllm/plugin.py:281-311; RotateQuantizedLinear x 3 + activation inside an HF MLP block,
 * transformers' LlamaMLP.forward); at one row that is a chain of dependent launches of a few microseconds each.  Here one resident
 * grid (one 16-wave workgroup per compute unit) runs the whole chain: every CU requests its INT4 tiles of linear i + 1 while linear
 * i's outputs are still being handed over, each 128-channel group is rotated ONCE per partition (by one wave, from the producers'
 * fp32 partial sums) and handed to the CUs that multiply by it -- csrc/engine.hip describes the protocol.
 *   phase i :  y_i = rotate_i(x_i * cs_i) @ dequant(W_i) + bias_i,  rounded once to the activation type (what the linear would have
 *              stored);  x_0 = x,  x_{i+1} = y_i[in_col0_{i+1} : in_col0_{i+1} + K_{i+1}]
 * One row; krot <= 8; quantisation group_size 128; in_col0 even; at most 448 phases of at most 8 distinct linear shapes per chain
 * (the plan is cached on chip).  All layers of a chain share the activation type.
 *   paro_engine_plan   host only: chooses the work split for `n_cus` compute units (0 = the current device's) and fills `out`
 *                      (sizes of the plan blob and of the workspace).
 *   paro_engine_build  host only: writes the plan blob (plan_bytes) into HOST memory; the caller copies it to the device.  The
 *                      blob holds the layers' device pointers (wq / sz / rot / channel_scales / bias): they must stay alive and in
 *                      place while the plan is in use.
 *   paro_engine_run    one launch on `stream`; HIP-graph capturable, no host work on replay.  workspace: workspace_bytes,
 *                      zero-filled ONCE by the caller, private to this engine instance (launches of one instance must not overlap).
 *                      Workspace word 1 is the sticky status of the hand-offs (PARO_WS_STATUS_GIVEUP: a wait was abandoned after
 *                      its bound and the outputs are NaN -- impossible while the whole grid is resident, which the call checks).
 *   paro_engine_describe  the split the planner chose for one phase (K-chunks, most / fewest tiles per CU): tooling and tests. */
typedef struct paro_engine_phase {
  const struct paro_linear* L;
  int64_t in_col0;       /* first column of the previous phase's output that this linear reads (0 for phase 0) */
  int32_t flags;         /* reserved, 0 */
  int32_t reserved0;
} paro_engine_phase_t;
typedef struct paro_engine {
  int32_t n_phases, n_cus, act_dtype, last_split;
  int64_t plan_bytes, workspace_bytes, in_features, out_features, last_out_offset;
  const void* last_bias;
  int32_t n_shapes, shape_off[8];   /* distinct linear shapes of the chain and where their work tables start in the plan */
  int32_t reserved0;
} paro_engine_t;
int paro_engine_plan(const paro_engine_phase_t* phases, int n_phases, int n_cus, paro_engine_t* out);
int paro_engine_build(const paro_engine_phase_t* phases, const paro_engine_t* e, void* plan_host);
int paro_engine_describe(const paro_engine_phase_t* phases, const paro_engine_t* e, int phase, int32_t* out_split,
                         int32_t* out_max_tiles, int32_t* out_min_tiles);
int paro_engine_run(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* Diagnostic twin of paro_engine_run (its own kernel instantiation; never on the hot path): the same launch, and every compute unit
 * stamps its events per phase with the chip-wide 100 MHz counter into trace: uint64 [n_phases][n_cus][32] (16 events, then the shader-clock counter at the same 16 events) --
 * service wave: 0 phase entered, 1 partial sums arrived, 2 rotated group published; wave 0: 3 gather entered, 4 gathered, 5 past the
 * first barrier, 8 its units consumed, 6 partial sums staged, 9 past the second barrier, 7 outputs published (tools/engine_timeline.py
 * turns them into the per-edge timeline). */
int paro_engine_trace(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                      int64_t workspace_bytes, void* trace, void* stream);

/* v15: the engine's second build (csrc/engine2.hip) -- the same five entry points, the same descriptors, another geometry: per CU one LOADER
 * wave streams the INT4 tiles HBM -> LDS by LDS-DMA into a ring of 7 x 16 tiles and runs ahead across the linears; seven CONSUMER waves
 * (a build parameter) each own every 7th group of the CU's K-chunk: they complete the previous linear's K-chunk partial sums for those
 * groups (one hop per edge), rotate them in registers and multiply their tiles out of LDS.  Parity-tested; measured SLOWER than the engine
 * above and than the per-call launches (profiles/NOTES.md 5.1): kept for whoever continues the design.
 * paro_engine_phase_t.flags & 0xf: 0 = the planner's K-split, 1..4 = that many K-chunks (tuning, tests).  trace: uint64
 * [n_phases][n_cus][8 waves][8 events] stamps of the 100 MHz counter -- consumer waves (rows 1..): 0 phase entered, 1 hand-off loads
 * issued, 2 its first groups' partial sums there, 3 rotated and in LDS, 4 its tiles accumulated, 5 every wave of the CU has arrived, 6 its
 * share of the outputs published; the loader (row 0): 0 first slot of the phase issued, 1 last slot issued, 2 ticks waiting for a free
 * slot, 3 ticks yielding to hand-offs.  Environment (experiments, read per call): PARO_E2_THIN = 0 / 1 / 2 (how the loader yields while
 * a wave of its CU polls: not / one 4 KiB burst in flight / stands still; default 0), PARO_E2_COST, PARO_E2_COST_GATHER
 * (the planner's constants).  Replaces rotate -> GEMM per linear of transformers/modules.py:57-71 / vllm/plugin.py:281-311 for a caller
 * that owns the chain. */
int paro_engine2_plan(const paro_engine_phase_t* phases, int n_phases, int n_cus, paro_engine_t* out);
int paro_engine2_build(const paro_engine_phase_t* phases, const paro_engine_t* e, void* plan_host);
int paro_engine2_describe(const paro_engine_phase_t* phases, const paro_engine_t* e, int phase, int32_t* out_split,
                          int32_t* out_max_tiles, int32_t* out_min_tiles);
int paro_engine2_run(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                     int64_t workspace_bytes, void* stream);
int paro_engine2_trace(const paro_engine_t* e, const void* plan_dev, const void* x, void* y, void* workspace,
                       int64_t workspace_bytes, void* trace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARO_ABI_EXPERIMENTAL_H */

#!/bin/bash
# One GPU session that produces what is kept under profiles/ for a round:
#   bash tools/profile_round.sh r03       (run from the repo root on the GPU box; scratch under gpurun_out/<round>/)
# decode     : bench lines + per-shape tables (qwen3-4b = the driver's line, llama3-8b, qwen3-0.6b, qwen3.5-9b, llama3-70b x 8 layers),
#              rocprofv3 kernel traces of the qwen3-4b / llama3-8b / llama3-70b bench commands,
#              batched decode (--rows 2..16) through both routes, the C++ chain harness (fused vs chain per shape and per layer)
# end to end : fused decode harness tokens/s (deferred K-split reduction, and the in-launch reducer for comparison); tools/bench_parts.py
# prefill    : TFLOP/s per linear at M = 65536 (variant 4), MoE grouped prefill vs the per-expert loop
# PMC        : HBM traffic of the bench command (separate FETCH_SIZE / WRITE_SIZE passes) -- LAST, so that the summary's
#              kernel_sources_sha is the tree's (bench.py refuses an older summary as roofline.traffic)
set -u
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R/prof
P=$ROOT/profiles
mkdir -p $OUT $P
export TMPDIR=/tmp
# ---- decode bench lines
for wl in qwen3-4b llama3-8b qwen3-0.6b qwen3.5-9b; do
  extra=""; [ $wl != qwen3-4b ] && extra="--no-cpu-baseline --no-north-star"
  timeout 400 python bench.py --workload $wl --per-shape $extra > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json > $P/${R}_bench_$wl.jsonl
  grep us_per_launch $OUT/bench_$wl.err >> $P/${R}_bench_$wl.jsonl
done
timeout 400 python bench.py --workload llama3-70b --layers 8 --per-shape --no-cpu-baseline --no-e2e --no-north-star > $OUT/bench_l70.json 2> $OUT/bench_l70.err
tail -1 $OUT/bench_l70.json > $P/${R}_bench_llama3-70b_8layers.jsonl; grep us_per_launch $OUT/bench_l70.err >> $P/${R}_bench_llama3-70b_8layers.jsonl
# ---- batched decode: both routes
rm -f $P/${R}_rows_bench.jsonl
for spec in "qwen3-4b:" "llama3-8b:--layers 8"; do
  wl=${spec%%:*}; la=${spec#*:}
  for rows in 2 4 8 16; do
    for route in chain fused; do
      timeout 300 python bench.py --workload $wl $la --rows $rows --route $route --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-north-star 2>> $OUT/rows.err | tail -1 >> $P/${R}_rows_bench.jsonl
    done
  done
done
# ---- the chain harness (no Python): per-layer and per-shape, fused vs chain, rows 1 / 8
if [ -x tools/chain_harness ]; then
  rm -f $P/${R}_chain_harness.jsonl
  for m in qwen3-4b llama3-8b; do for rows in 1 8; do timeout 120 tools/chain_harness paroquant_amd/_lib/libparo_mi355x.so $m 0 $rows >> $P/${R}_chain_harness.jsonl 2>> $OUT/harness.err; done; done
fi
# ---- kernel traces (rocprofv3 --kernel-trace --stats) of the bench command per model
cd /tmp
for spec in "qwen3-4b:" "llama3-8b:--workload llama3-8b" "llama3-70b_8layers:--workload llama3-70b --layers 8"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o b -- python $ROOT/bench.py $args --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-route-ab --no-north-star > $OUT/stats_$name.log 2>&1
  S=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && (cd $ROOT && python tools/pmc_summary.py stats $S $P/${R}_bench_${name}_kernel_stats.csv)
done
cd $ROOT
# ---- end to end: kernel trace of the decode harness (per-launch durations of one model), then tokens/s
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_stats -o b -- python $ROOT/tools/bench_e2e.py --model qwen3-4b --runs 2 --warmup 1 > $OUT/e2e_stats.log 2>&1)
S=$(find $OUT/e2e_stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S $OUT/e2e_stats_short.csv && head -12 $OUT/e2e_stats_short.csv > $P/${R}_e2e_qwen3-4b_kernel_stats.csv
rm -f $P/${R}_e2e.jsonl
for m in qwen3-4b llama3-8b qwen3-0.6b qwen3.5-9b; do timeout 400 python tools/bench_e2e.py --model $m >> $P/${R}_e2e.jsonl 2>> $OUT/e2e.err; done
# ... and with the in-launch K-split reducer instead of the deferred reduction (paroquant_amd/decoder.py), same session
rm -f $P/${R}_e2e_reducer.jsonl
for m in qwen3-4b llama3-8b; do PARO_DEFERRED_KSPLIT=0 timeout 300 python tools/bench_e2e.py --model $m >> $P/${R}_e2e_reducer.jsonl 2>> $OUT/e2e.err; done
# ---- decode attention alone: the split launch (merge left to o_proj) and the in-launch merge, short and long caches
rm -f $P/${R}_attn_split.jsonl
for t in 264 2048 8192; do
  timeout 120 python tools/bench_attn.py --tmax $t --positions 0,63,128,255,256,511,700,2047,8191 --split >> $P/${R}_attn_split.jsonl 2>> $OUT/attn.err
  timeout 120 python tools/bench_attn.py --tmax $t --positions 0,63,128,255,256,511,700,2047,8191 >> $P/${R}_attn_split.jsonl 2>> $OUT/attn.err
done
# ... and end to end without it (PARO_SPLIT_ATTN=0), same session
rm -f $P/${R}_e2e_nosplit.jsonl
for m in qwen3-4b llama3-8b; do PARO_SPLIT_ATTN=0 timeout 300 python tools/bench_e2e.py --model $m >> $P/${R}_e2e_nosplit.jsonl 2>> $OUT/e2e.err; done
# ---- deferred K-split reduction per launch and per producer -> consumer pair
rm -f $P/${R}_parts_micro.jsonl
for m in qwen3-4b llama3-8b; do timeout 300 python tools/bench_parts.py --model $m >> $P/${R}_parts_micro.jsonl 2>> $OUT/parts.err; done
# (the persistent engine's per-edge timeline: EXPERIMENTAL builds only -- tools/engine_timeline.py; profiles/r04_engine_timeline*)
# ---- determinism stress of the fused GEMV family (10 000 iterations x 8 cases; tools/stress_fused.py)
timeout 900 python tools/stress_fused.py 10000 2>&1 | grep -v amdgpu.ids | tail -3 > $P/${R}_stress_fused.txt
# ---- prefill
timeout 400 python tools/bench_gemm.py --model llama3-8b --rows 65536 --variants 0 > $P/${R}_prefill_llama3-8b.jsonl 2> $OUT/gemm.err
timeout 300 python tools/bench_moe.py > $P/${R}_moe_prefill.jsonl 2>> $OUT/gemm.err
# ---- BASELINE config 3 (Qwen3.5-4B: batch-1 decode + batch 32 x 2048 prefill): the 4B-class hybrid linear set, decode line and prefill TFLOP/s
timeout 400 python bench.py --workload qwen3.5-4b-class --per-shape --no-cpu-baseline --no-e2e --no-north-star > $OUT/bench_q35_4b.json 2> $OUT/bench_q35_4b.err
tail -1 $OUT/bench_q35_4b.json > $P/${R}_bench_qwen3.5-4b-class.jsonl; grep us_per_launch $OUT/bench_q35_4b.err >> $P/${R}_bench_qwen3.5-4b-class.jsonl
timeout 400 python tools/bench_gemm.py --model qwen3.5-4b-class --rows 65536 --variants 0 > $P/${R}_prefill_qwen3.5-4b-class.jsonl 2>> $OUT/gemm.err
# ---- PMC passes LAST
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star > $OUT/write.log 2>&1
cd $ROOT
# L2-side traffic (TCP -> TCC read requests, hit rate): its own pass
cd /tmp
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star > $OUT/l2.log 2>&1
cd $ROOT
python tools/pmc_l2.py $OUT/l2 qwen3-4b $P/${R}_pmc_l2_qwen3-4b.json
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1))
W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1))
python tools/pmc_summary.py pmc $F $W qwen3-4b $P/${R}_pmc_bench_qwen3-4b.json
mkdir -p $ROOT/gpurun_out/$R/profiles_copy && cp $P/${R}_* $ROOT/gpurun_out/$R/profiles_copy/
tail -1 $P/${R}_bench_qwen3-4b.jsonl | cut -c1-300; head -1 $P/${R}_bench_qwen3-4b.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline'], d.get('end_to_end'), d.get('cpu_baseline'))"

#!/bin/bash
# The reduced profile session of a round (a few GPU-minutes; tools/profile_round.sh is the full one):
#   bash tools/profile_round_min.sh r05     (from the repo root on the GPU box; scratch under gpurun_out/<round>/)
# bench lines (qwen3-4b = the driver's line with extras; llama3-8b per shape), the kernel trace of the bench command, the prefill table
# and its matrix-core counters, the engine timelines, and LAST the HBM / L2 counter passes of the bench command (bench.py refuses a
# traffic figure collected on other kernel sources).
set -u
R=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R/prof
P=$ROOT/profiles
mkdir -p $OUT $P
export TMPDIR=/tmp
timeout 400 python bench.py --per-shape > $OUT/bench_q4b.json 2> $OUT/bench_q4b.err
tail -1 $OUT/bench_q4b.json > $P/${R}_bench_qwen3-4b.jsonl; grep us_per_launch $OUT/bench_q4b.err >> $P/${R}_bench_qwen3-4b.jsonl
timeout 400 python bench.py --workload llama3-8b --per-shape --no-cpu-baseline --no-north-star --no-extra > $OUT/bench_l8b.json 2> $OUT/bench_l8b.err
tail -1 $OUT/bench_l8b.json > $P/${R}_bench_llama3-8b.jsonl; grep us_per_launch $OUT/bench_l8b.err >> $P/${R}_bench_llama3-8b.jsonl
# kernel trace of the bench command
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-route-ab --no-north-star --no-extra > $OUT/stats.log 2>&1)
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S $P/${R}_bench_qwen3-4b_kernel_stats.csv
# prefill: TFLOP/s per linear at M = 65536, then the matrix-core counters of the same run (two passes: instruction counts, busy cycles)
timeout 300 python tools/bench_gemm.py --model llama3-8b --rows 65536 --variants 0 > $P/${R}_prefill_llama3-8b.jsonl 2> $OUT/gemm.err
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/gemm_pmc1 -o p -- python $ROOT/tools/bench_gemm.py --model llama3-8b --rows 65536 --variants 0 --rounds 1 --reps 1 > $OUT/gemm_pmc1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/gemm_pmc2 -o p -- python $ROOT/tools/bench_gemm.py --model llama3-8b --rows 65536 --variants 0 --rounds 1 --reps 1 > $OUT/gemm_pmc2.log 2>&1)
python tools/pmc_gemm_summary.py $OUT/gemm_pmc1 $OUT/gemm_pmc2 > $P/${R}_gemm_pmc.json 2>> $OUT/gemm.err
# (the persistent engines' timelines: EXPERIMENTAL builds only -- tools/engine_timeline.py, tools/engine2_timeline.py; profiles/r05_engine*)
# ---- PMC passes LAST
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star --no-extra > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star --no-extra > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-route-ab --no-north-star --no-extra > $OUT/l2.log 2>&1
cd $ROOT
python tools/pmc_l2.py $OUT/l2 qwen3-4b $P/${R}_pmc_l2_qwen3-4b.json
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1))
W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1))
python tools/pmc_summary.py pmc $F $W qwen3-4b $P/${R}_pmc_bench_qwen3-4b.json
mkdir -p $ROOT/gpurun_out/$R/profiles_copy && cp $P/${R}_* $ROOT/gpurun_out/$R/profiles_copy/
# the bench line once more: now it carries roofline.traffic from this tree's counters
timeout 400 python bench.py --no-cpu-baseline --no-north-star --no-extra --no-e2e > $OUT/bench_q4b_traffic.json 2>> $OUT/bench_q4b.err
tail -1 $OUT/bench_q4b_traffic.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline'])"

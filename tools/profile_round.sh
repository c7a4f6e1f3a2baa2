#!/bin/bash
# One GPU session that produces everything kept under profiles/ for a round:
#   bash tools/profile_round.sh r01       (run from the repo root on the GPU box; scratch under gpurun_out/)
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT $ROOT/profiles
export TMPDIR=/tmp
for wl in qwen3-4b llama3-8b; do
  [ "${SKIP_BENCH:-0}" = 1 ] && continue
  timeout 400 python bench.py --workload $wl --per-shape > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json >> $ROOT/profiles/${R}_bench_$wl.jsonl
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/write.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S profiles/${R}_bench_qwen3-4b_kernel_stats.csv
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1))
W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1))
python tools/pmc_summary.py pmc $F $W qwen3-4b profiles/${R}_pmc_bench_qwen3-4b.json
tail -2 $OUT/bench_qwen3-4b.err

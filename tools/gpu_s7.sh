#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
for m in qwen3-4b llama3-8b qwen3-0.6b; do
  timeout 300 python tools/bench_e2e.py --model $m >> $O/s7_e2e.jsonl 2>> $O/s7.err
done
cat $O/s7_e2e.jsonl; tail -5 $O/s7.err
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s7_stats -o e2e -- python $R/tools/bench_e2e.py --model qwen3-4b --runs 1 --warmup 1 > $R/$O/s7_stats.log 2>&1
cd $R
S=$(find $O/s7_stats -name "*kernel_stats.csv" | head -1); head -14 $S | cut -c1-160

#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "attn_decode or decoder_harness" > $O/s8_pytest.log 2>&1
tail -30 $O/s8_pytest.log
rm -f $O/s8_e2e.jsonl
for m in qwen3-4b llama3-8b qwen3-0.6b; do
  timeout 300 python tools/bench_e2e.py --model $m >> $O/s8_e2e.jsonl 2>> $O/s8.err
done
timeout 300 python tools/bench_e2e.py --model qwen3-4b --prompt 1900 --new 128 >> $O/s8_e2e.jsonl 2>> $O/s8.err
cat $O/s8_e2e.jsonl | cut -c100-420
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s8_stats -o e2e -- python $R/tools/bench_e2e.py --model qwen3-4b --runs 1 --warmup 1 > $R/$O/s8_stats.log 2>&1
cd $R
S=$(find $O/s8_stats -name "*kernel_stats.csv" | head -1); head -9 $S | cut -c1-150

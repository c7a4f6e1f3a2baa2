#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "attn_decode or decoder_harness" > $O/s9_pytest.log 2>&1
tail -5 $O/s9_pytest.log
rm -f $O/s9_e2e.jsonl
timeout 300 python tools/bench_e2e.py --model qwen3-4b >> $O/s9_e2e.jsonl 2>> $O/s9.err
timeout 300 python tools/bench_e2e.py --model qwen3-4b --prompt 1900 --new 128 >> $O/s9_e2e.jsonl 2>> $O/s9.err
cut -c100-330 $O/s9_e2e.jsonl
timeout 300 python tools/bench_fused.py --model qwen3-4b > $O/s9_fused.jsonl 2>> $O/s9.err
cat $O/s9_fused.jsonl
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s9_stats -o e2e -- python $R/tools/bench_e2e.py --model qwen3-4b --runs 1 --warmup 1 > $R/$O/s9_stats.log 2>&1
cd $R
S=$(find $O/s9_stats -name "*kernel_stats.csv" | head -1); head -7 $S | cut -c1-130

#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "lm_head or decoder_harness" > $O/s13_pytest.log 2>&1
tail -25 $O/s13_pytest.log
rm -f $O/s13_e2e.jsonl
for m in qwen3-4b llama3-8b qwen3-0.6b; do timeout 300 python tools/bench_e2e.py --model $m >> $O/s13_e2e.jsonl 2>> $O/s13.err; done
cut -c100-330 $O/s13_e2e.jsonl

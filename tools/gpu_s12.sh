#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fused_ or decoder_harness" > $O/s12_pytest.log 2>&1
tail -5 $O/s12_pytest.log
timeout 300 python tools/bench_fused.py --model qwen3-4b > $O/s12_fused.jsonl 2>> $O/s12.err
cat $O/s12_fused.jsonl
rm -f $O/s12_e2e.jsonl
timeout 300 python tools/bench_e2e.py --model qwen3-4b >> $O/s12_e2e.jsonl 2>> $O/s12.err
cut -c100-330 $O/s12_e2e.jsonl

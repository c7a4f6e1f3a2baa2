#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 300 python tools/mall_probe.py --model qwen3-4b > $O/s4_mall_q4b.jsonl 2> $O/s4.err
cat $O/s4_mall_q4b.jsonl; tail -3 $O/s4.err
timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "tp_sharded" 2>&1 | tail -30

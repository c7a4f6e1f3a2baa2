#!/bin/bash
# round-2 GPU session 1: parity suite + prefill variant A/B + decode sanity
set -u
O=gpurun_out/r02
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=25 -p no:cacheprovider > $O/s1_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s1_pytest.log
tail -5 $O/s1_pytest.log
timeout 300 python tools/bench_gemm.py --model llama3-8b --rows 8192,65536 --variants 3,4 > $O/s1_gemm_l8b.jsonl 2> $O/s1_gemm_l8b.err
timeout 300 python tools/bench_gemm.py --model qwen3-4b --rows 8192,65536 --variants 3,4 > $O/s1_gemm_q4b.jsonl 2> $O/s1_gemm_q4b.err
timeout 200 python tools/bench_gemm.py --model llama3-8b --rows 8192 --variants 1,4 --dtype bf16 > $O/s1_gemm_l8b_bf16.jsonl 2> $O/s1_gemm_bf16.err
timeout 200 python bench.py --no-cpu-baseline > $O/s1_bench_q4b.json 2> $O/s1_bench_q4b.err
cat $O/s1_gemm_l8b.jsonl $O/s1_gemm_q4b.jsonl $O/s1_gemm_l8b_bf16.jsonl | cut -c1-250
tail -2 $O/s1_bench_q4b.json | cut -c1-600

#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 300 python tools/bench_gemm.py --model llama3-8b --only gate_up_proj,down_proj --rows 8192 --variants 3,4,41,42,43 --rounds 3 > $O/s3_gemm_ablate.jsonl 2> $O/s3.err
cut -c1-230 $O/s3_gemm_ablate.jsonl; tail -3 $O/s3.err

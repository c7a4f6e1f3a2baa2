#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
for d in 0 1 2 3; do PARO_ATTN_DBG=$d timeout 120 python tools/bench_attn.py --positions 0,100,255,700 2>/dev/null; done | tee $O/s10_attn.jsonl

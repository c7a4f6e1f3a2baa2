#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "attn_decode or decoder_harness" > $O/s11_pytest.log 2>&1
tail -5 $O/s11_pytest.log
for d in 0 1 2 3; do PARO_ATTN_DBG=$d timeout 120 python tools/bench_attn.py --positions 0,100,255,700,2047 2>/dev/null; done | tee $O/s11_attn.jsonl
rm -f $O/s11_e2e.jsonl
timeout 300 python tools/bench_e2e.py --model qwen3-4b >> $O/s11_e2e.jsonl 2>> $O/s11.err
cut -c100-330 $O/s11_e2e.jsonl

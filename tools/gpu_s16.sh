#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
for p in 0 1; do
  PARO_GEMV_PRIO=$p PARO_GEMV_PD=31 python tools/timeline_gemv.py --model llama3-8b --linear o_proj --tpw 1 --waves 16 2>/dev/null | sed -n 1,3p\;11,28p | sed "s/^/prio=$p /"
  PARO_GEMV_PRIO=$p PARO_GEMV_PD=31 python tools/timeline_gemv.py --model qwen3-4b --linear gate_up_proj --tpw 8 --waves 8 2>/dev/null | sed -n 1,3p\;11,20p | sed "s/^/prio=$p /"
done > $O/s16_timeline.txt
cat $O/s16_timeline.txt

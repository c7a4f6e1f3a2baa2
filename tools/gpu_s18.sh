#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
for p in 0 1 0 1; do
  for wl in qwen3-4b llama3-8b; do
    PARO_GEMV_PRIO=$p timeout 300 python bench.py --workload $wl --no-cpu-baseline --per-shape 2> $O/s18_tmp.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=$p', '$wl', d['value'], 'tok/s', d['roofline']['frac'])"
    grep us_per_launch $O/s18_tmp.jsonl | python -c "import sys,json; print('   ', [ (json.loads(l)['linear'], json.loads(l)['us_per_launch']) for l in sys.stdin])"
  done
done | tee $O/s18_prio.txt

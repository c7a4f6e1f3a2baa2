#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "gemv or fused_ or moe_ or vllm or full_size" > $O/s21_pytest.log 2>&1
tail -4 $O/s21_pytest.log
for i in 1 2; do
  for wl in qwen3-4b llama3-8b; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-e2e --per-shape 2> $O/s21_tmp.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], 'tok/s', d['roofline']['frac'])"
    grep us_per_launch $O/s21_tmp.jsonl | python -c "import sys,json; print('   ', [ (json.loads(l)['linear'], json.loads(l)['us_per_launch']) for l in sys.stdin])"
  done
done | tee $O/s21_bench.txt

#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
for m in qwen3-4b llama3-8b; do
  timeout 600 python tools/sweep_gemv.py --model $m --reps 100 --rounds 3 --tpw 1,2,4,8 --ksplit 1,2,4 --waves 4,8,16 > $O/s19_sweep_$m.jsonl 2> $O/s19.err
  python - <<PY
import json,collections
rows=[json.loads(l) for l in open("$O/s19_sweep_$m.jsonl")]
by=collections.defaultdict(list)
for r in rows: by[r["linear"]].append(r)
for k,v in by.items():
    v.sort(key=lambda r:r["us"])
    print("$m",k,[(r["tpw"],r["ksplit"],r["waves"],r["us"]) for r in v[:6]])
PY
done

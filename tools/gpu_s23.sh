#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 600 python bench.py --gpus 2 --workload llama3-70b-tp --layers 2 --tp-backend gloo --same-device --steps 3 --warmup 1 > $O/s23_tp2_gloo.json 2> $O/s23.err
tail -1 $O/s23_tp2_gloo.json | cut -c1-900; tail -3 $O/s23.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --workload qwen3-32b-tp --layers 2 --tp-backend gloo --same-device --steps 3 --warmup 1 > $O/s23_tp4_gloo.json 2>> $O/s23.err
tail -1 $O/s23_tp4_gloo.json | cut -c1-700; tail -3 $O/s23.err

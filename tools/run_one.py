#!/usr/bin/env python3
"""Run ONE fused-GEMV configuration a few times (eager launches, distinct weight copies) -- the
target of rocprofv3 --pmc / --kernel-trace runs.
    python tools/run_one.py --model llama3-8b --linear gate_up_proj --tpw 8 --ksplit 1 --waves 8 --n 20"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import alg_bytes, layer_shapes, synth_packed
from paroquant_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b"); ap.add_argument("--linear", default="gate_up_proj")
ap.add_argument("--tpw", type=int, default=0); ap.add_argument("--ksplit", type=int, default=0)
ap.add_argument("--waves", type=int, default=0); ap.add_argument("--rows", type=int, default=1)
ap.add_argument("--n", type=int, default=20); ap.add_argument("--copies", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(0)
name, K, sizes, _ = [s for s in layer_shapes(a.model) if s[0] == a.linear][0]
packs = [synth_packed(K, sizes, dev, gen) for _ in range(a.copies)]
x = torch.randn(a.rows, K, device=dev, dtype=torch.float16, generator=gen)
torch.cuda.synchronize()
for i in range(a.n):
    ops.w4a16_gemv_tuned(x, packs[i % a.copies], a.tpw, a.ksplit, a.waves, 0)
torch.cuda.synchronize()
print("bytes", alg_bytes(K, sum(sizes), len(sizes)))

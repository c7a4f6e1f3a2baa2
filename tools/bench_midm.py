#!/usr/bin/env python3
"""Fused linear at the batched-decode sizes M = 1 .. 512 (auto dispatch: fused GEMV / pre-pass + GEMV /
pre-pass + MFMA GEMM), per Llama-3-8B / Qwen3-4B linear: us per call inside a HIP graph and the effective
weight-stream rate (algorithmic bytes / time).
    python tools/bench_midm.py [--model llama3-8b] [--rows 1,4,8,16,32,64,128,256,512]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import alg_bytes, layer_shapes, synth_packed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--rows", default="1,4,8,16,32,64,128,256,512")
    ap.add_argument("--reps", type=int, default=16)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    for name, K, sizes, _ in layer_shapes(args.model):
        if args.only and name not in args.only.split(","):
            continue
        nb = alg_bytes(K, sum(sizes), len(sizes))
        copies = max(2, min(8, int((1 << 29) // nb) + 1))
        packs = [synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        for rows in [int(r) for r in args.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            for pk in packs:          # every copy once outside the capture (lazy prefill state is built here)
                pk.apply(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(args.reps):
                    packs[i % copies].apply(x)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / args.reps)
            us = float(np.median(ts))
            print(json.dumps({"model": args.model, "linear": name, "M": rows, "us": round(us, 2),
                              "weight_GBps": round(nb / us / 1e3, 1),
                              "TFLOPs": round(2.0 * rows * K * sum(sizes) / us / 1e6, 1)}), flush=True)
        del packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

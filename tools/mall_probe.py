#!/usr/bin/env python3
"""Does touching a layer's packed weights (ops.prefetch) leave them in the Infinity Cache for the GEMV that follows?
For each Qwen3-4B / Llama-3-8B linear: time the GEMV (HIP events around the single launch, median of N)
  cold : after streaming 1 GiB of unrelated data (flush),
  warm : flush, then ops.prefetch(wq, sz, rot), then the GEMV,
  hot  : the GEMV repeated back to back on the same weights (L2 / MALL resident upper bound).
    python tools/mall_probe.py [--model qwen3-4b]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--n", type=int, default=15)
    ap.add_argument("--wgs", default="64,256")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    junk = torch.empty(1 << 28, dtype=torch.int32, device=dev)   # 1 GiB
    junk.random_()
    sink = torch.zeros(1, dtype=torch.int32, device=dev)

    def flush():
        sink.add_(junk.sum().to(torch.int32))

    def timed(pre, fn):
        ts = []
        for _ in range(args.n):
            flush()
            pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts)), float(np.min(ts))

    for name, K, sizes, _ in bench.layer_shapes(args.model):
        pk = bench.synth_packed(K, sizes, dev, gen)
        x = torch.randn(1, K, device=dev, dtype=torch.float16, generator=gen)
        nb = bench.alg_bytes(K, sum(sizes), len(sizes))
        pk.apply(x); torch.cuda.synchronize()
        row = {"model": args.model, "linear": name, "bytes": nb}
        row["cold_us"], _ = timed(lambda: None, lambda: pk.apply(x))
        for w in [int(v) for v in args.wgs.split(",")]:
            row[f"warm{w}_us"], _ = timed(lambda: ops.prefetch(pk.stream_buffers(), w), lambda: pk.apply(x))
            row[f"prefetch{w}_us"], _ = timed(lambda: None, lambda: ops.prefetch(pk.stream_buffers(), w))
        row["hot_us"], _ = timed(lambda: pk.apply(x), lambda: pk.apply(x))
        row["note"] = "single eager launches: each figure includes ~2-3 us of launch/event overhead"
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Deferred K-split reduction (include/paro_abi.h v12) against the in-launch reducer, per launch and per producer -> consumer pair.

    python tools/bench_parts.py [--model qwen3-4b] [--reps 200] > gpurun_out/parts.jsonl

For (o_proj -> gate_up) and (down_proj -> qkv) of the model, each variant captured in a HIP graph of `reps` launches (pairs: reps / 2
pairs) that cycle >= 1 GiB of distinct weight copies, variants interleaved over 5 rounds, median reported:
  producer   : plain | residual epilogue (in-launch reducer)  |  parts_out
  consumer   : plain | residual epilogue only | RMSNorm prologue | RMSNorm + parts_in | RMSNorm + parts_in + x_out      (partial sums not rewritten: cache-warm)
  pair       : [producer(residual) -> consumer(RMSNorm)]  |  [producer(parts_out) -> consumer(RMSNorm + parts_in + x_out)]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bench import alg_bytes, layer_shapes, synth_packed
from paroquant_amd import ops, _native as nat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    sh = {name: (K, sizes) for name, K, sizes, _ in layer_shapes(args.model, 1)}
    R = nat.PROLOGUE_RMSNORM
    for pname, cname in (("o_proj", "gate_up_proj"), ("down_proj", "qkv_proj")):
        (Kp, sp), (Kc, sc) = sh[pname], sh[cname]
        H = Kc
        nbp, nbc = alg_bytes(Kp, sum(sp), 1), alg_bytes(Kc, sum(sc), len(sc))
        cp = max(2, min(48, int((1 << 30) // nbp) + 1))
        cc = max(2, min(48, int((1 << 30) // nbc) + 1))
        prod = [synth_packed(Kp, sp, dev, gen) for _ in range(cp)]
        cons = [synth_packed(Kc, sc, dev, gen) for _ in range(cc)]
        n = ops.gemv_parts_count(prod[0])
        xa = torch.randn(1, Kp, device=dev, dtype=torch.float16, generator=gen)
        h0 = torch.randn(1, H, device=dev, dtype=torch.float16, generator=gen)
        h1 = torch.zeros(1, H, device=dev, dtype=torch.float16)
        parts = torch.zeros(H, 4, device=dev, dtype=torch.float32)
        yc = torch.zeros(1, sum(sc), device=dev, dtype=torch.float16)
        resc = torch.zeros(1, sum(sc), device=dev, dtype=torch.float16)
        variants = {
            "producer plain": lambda i: ops.w4a16_gemv_tuned(xa, prod[i % cp], 0, 0, 0, 0),
            "consumer residual only": lambda i: ops.w4a16_gemv_fused(h0, cons[i % cc], 0, residual=resc, out=yc),
            "producer residual": lambda i: ops.w4a16_gemv_fused(xa, prod[i % cp], 0, residual=h0, out=h1),
            "producer parts_out": lambda i: ops.w4a16_gemv_fused(xa, prod[i % cp], 0, parts_out=parts),
            "consumer plain": lambda i: ops.w4a16_gemv_tuned(h0, cons[i % cc], 0, 1, 0, 0),
            "consumer rmsnorm": lambda i: ops.w4a16_gemv_fused(h0, cons[i % cc], R, 1e-6, out=yc),
            "consumer rmsnorm+parts_in": lambda i: ops.w4a16_gemv_fused(h0, cons[i % cc], R, 1e-6, out=yc, parts_in=parts),
            "consumer rmsnorm+parts_in+x_out": lambda i: ops.w4a16_gemv_fused(h0, cons[i % cc], R, 1e-6, out=yc, parts_in=parts, x_out=h1.view(-1)),
            "pair reducer": lambda i: (ops.w4a16_gemv_fused(xa, prod[(i // 2) % cp], 0, residual=h0, out=h1)
                                       if i % 2 == 0 else ops.w4a16_gemv_fused(h1, cons[(i // 2) % cc], R, 1e-6, out=yc)),
            "pair deferred": lambda i: (ops.w4a16_gemv_fused(xa, prod[(i // 2) % cp], 0, parts_out=parts)
                                        if i % 2 == 0 else ops.w4a16_gemv_fused(h0, cons[(i // 2) % cc], R, 1e-6, out=yc, parts_in=parts, x_out=h1.view(-1))),
        }
        graphs = {}
        for name, fn in variants.items():
            for i in range(4):
                fn(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(args.reps):
                    fn(i)
            graphs[name] = g
        times = {k: [] for k in graphs}
        for _ in range(args.rounds):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / args.reps)
        for k, ts in times.items():
            per = "us_per_pair" if k.startswith("pair") else "us_per_launch"
            print(json.dumps({"model": args.model, "producer": pname, "consumer": cname, "parts": n, "variant": k,
                              per: round(float(np.median(ts)) * (2 if k.startswith("pair") else 1), 3)}), flush=True)
        del graphs, prod, cons
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

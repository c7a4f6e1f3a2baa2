#!/usr/bin/env python3
"""A/B of the plain fused-rotation GEMV against its FUSED instantiation (prologue / epilogue fusions), per Qwen3-4B /
Llama-3-8B linear, M = 1: us per launch inside a HIP graph of `reps` launches cycling >= 1 GiB of distinct weights,
variants interleaved over `rounds` rounds.
    python tools/bench_fused.py [--model qwen3-4b]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd import ops, _native as nat


def graph_of(fn, reps):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tp", type=int, default=1, help="time one rank's shard of a TP model; adds the all-reduce-epilogue variant of o / down "
                                                       "(world of one: the epilogue's own cost, no peers)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ar = None
    if args.tp > 1:
        import torch.distributed as dist
        from paroquant_amd import tp as ptp
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29672")
        dist.init_process_group("gloo", rank=0, world_size=1)
        ar = ptp.OneShotAllReduce(dev, bench.MODELS[args.model][0])
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    for name, K, sizes, _ in bench.layer_shapes(args.model, args.tp):
        N = sum(sizes)
        nb = bench.alg_bytes(K, N, len(sizes))
        copies = max(2, min(64, int((1 << 30) // nb) + 1))
        packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        x = torch.randn(1, K, device=dev, dtype=torch.float16, generator=gen)
        x2 = torch.randn(1, 2 * K, device=dev, dtype=torch.float16, generator=gen)
        res = torch.randn(1, N, device=dev, dtype=torch.float16, generator=gen)
        out = torch.empty(1, N, device=dev, dtype=torch.float16)
        variants = {"plain": lambda i: packs[i % copies].apply(x),
                    "residual": lambda i: ops.w4a16_gemv_fused(x, packs[i % copies], 0, residual=res, out=out)}
        if name in ("qkv_proj", "gate_up_proj"):
            variants["rmsnorm"] = lambda i: ops.w4a16_gemv_fused(x, packs[i % copies], nat.PROLOGUE_RMSNORM, 1e-6, out=out)
        if name == "down_proj":
            variants["silu_mul+res"] = lambda i: ops.w4a16_gemv_fused(x2, packs[i % copies], nat.PROLOGUE_SILU_MUL, residual=res, out=out)
        if ar is not None and name in ("o_proj", "down_proj"):
            variants["residual+allreduce"] = lambda i: ops.w4a16_gemv_fused(x, packs[i % copies], 0, residual=res, out=out, allreduce=ar)
            part = torch.empty(1, N, device=dev, dtype=torch.float16)
            variants["plain+allreduce launch"] = lambda i: ar(ops.w4a16_gemv_fused(x, packs[i % copies], 0, out=part), residual=res, out=out)
        graphs = {k: graph_of(fn, args.reps) for k, fn in variants.items()}
        times = {k: [] for k in graphs}
        for _ in range(args.rounds):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / args.reps)
        print(json.dumps({"model": args.model, "tp": args.tp, "linear": name, "K": K, "N": N,
                          **{k: round(float(np.median(v)), 2) for k, v in times.items()}}), flush=True)
        del packs, graphs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

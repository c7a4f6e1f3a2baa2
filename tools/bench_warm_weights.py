#!/usr/bin/env python3
"""Is the one-row GEMV sensitive to where its weights come from?  us per launch inside a HIP graph of `reps` launches (a) cycling >= 1 GiB of
distinct weights (every launch streams from HBM: bench.py's protocol) and (b) on the SAME weights every time (L2 / Infinity-Cache resident
for the shapes that fit).  The difference bounds what any prefetch of the next linear's weights into the caches could buy (round 2 measured
none on that kernel: NOTES 3.1; this is the re-measurement on the round-6 kernel).
    python tools/bench_warm_weights.py [--model qwen3-4b] [--rows 1]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--rows", type=int, default=1)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    for name, K, sizes, _ in bench.layer_shapes(a.model):
        N = sum(sizes)
        nb = bench.alg_bytes(K, N, len(sizes))
        copies = max(2, min(64, int((1 << 30) // nb) + 1))
        packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        x = torch.randn(a.rows, K, device=dev, dtype=torch.float16, generator=gen)
        graphs = {}
        for kind in ("cold", "warm"):
            fn = (lambda i: packs[i % copies].apply(x)) if kind == "cold" else (lambda i: packs[0].apply(x))
            fn(0); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(a.reps):
                    fn(i)
            graphs[kind] = g
        ts = {k: [] for k in graphs}
        for _ in range(a.rounds):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g.replay(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                ts[k].append(e0.elapsed_time(e1) * 1e3 / a.reps)
        out = {"model": a.model, "linear": name, "K": K, "N": N, "rows": a.rows, "bytes": nb,
               "cold_us": round(float(np.median(ts["cold"])), 2), "warm_us": round(float(np.median(ts["warm"])), 2)}
        print(json.dumps(out), flush=True)
        del packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU tuning sweep for the fused GEMV: (tiles_per_wave x ksplit) per linear shape.

    python tools/sweep_gemv.py [--model llama3-8b] [--rows 1] [--reps 300] > gpurun_out/sweep.jsonl

Each variant is captured in a HIP graph of `reps` back-to-back launches that cycle through
>= 1 GiB of distinct weight copies (so the 256 MB Infinity Cache cannot serve them) and timed with
events around one replay; variants of one shape are interleaved over `rounds` rounds and the
median is reported (within-process A/B, cdna guide section 5.4 rule 24)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bench import MODELS, alg_bytes, layer_shapes, synth_packed
from paroquant_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--rows", type=int, default=1)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tpw", default="1,2,4,8")
    ap.add_argument("--ksplit", default="1,2,4")
    ap.add_argument("--waves", default="4,8,16")
    ap.add_argument("--mode", default="0")
    ap.add_argument("--order", type=int, default=-1, help="wq tile order: 0 [tile][group], 1 [group][tile], -1 auto")
    ap.add_argument("--only", default="", help="comma-separated linear names to run")
    ap.add_argument("--shape", default="", help="K:N1+N2+... -- one custom linear instead of a model's (e.g. 4096:1024, the north star's k_proj / v_proj)")
    ap.add_argument("--tp", type=int, default=1, help="per-rank shapes of a tensor-parallel split")
    ap.add_argument("--share_rot", type=int, default=0, help="1: every weight copy uses copy 0's rotation schedule and "
                    "channel scales (they stay cache resident): measures what the cold first touch of those small streams costs")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    tpws = [int(t) for t in args.tpw.split(",")]
    ksps = [int(k) for k in args.ksplit.split(",")]
    wvs = [int(k) for k in args.waves.split(",")]
    modes = [int(k) for k in args.mode.split(",")]
    import bench
    if args.shape:
        K_, cols = args.shape.split(":")
        shapes = [("custom", int(K_), [int(c) for c in cols.split("+")], "col")]
    elif args.model in bench.HYBRID:       # Qwen3.5 family: the distinct linears of both layer kinds
        shapes, seen = [], set()
        for full in (False, True):
            for nm, K_, sz_, kind in bench.hybrid_layer_shapes(args.model, full, args.tp):
                if (K_, tuple(sz_)) not in seen:
                    seen.add((K_, tuple(sz_)))
                    shapes.append((nm, K_, sz_, kind))
    else:
        shapes = layer_shapes(args.model, args.tp)
    for name, K, sizes, _ in shapes:
        nb = alg_bytes(K, sum(sizes), len(sizes))
        copies = max(2, min(48, int((1 << 30) // nb) + 1))
        if args.only and name not in args.only.split(","):
            continue
        packs = [synth_packed(K, sizes, dev, gen, None if args.order < 0 else args.order) for _ in range(copies)]
        if args.share_rot:
            for pk in packs[1:]:
                pk.rot, pk.channel_scales = packs[0].rot, packs[0].channel_scales
                if args.share_rot >= 2:
                    pk.sz = packs[0].sz
        x = torch.randn(args.rows, K, device=dev, dtype=torch.float16, generator=gen)
        graphs = {}
        G = K // 128
        import itertools
        for tpw, ksp, wv, mode in itertools.product(tpws, ksps, wvs, modes):
            auto = tpw == 0 and ksp == 0 and wv == 0          # (0, 0, 0) = the library's automatic launch shape, next to the forced ones
            if not auto and (0 in (tpw, ksp, wv) or ksp > max(1, G // 2) or (tpw == 8 and wv == 16) or (args.rows > 4 and wv == 16)
                             or tpw not in (1, 2, 4, 8)):
                continue

            def run(i, tpw=tpw, ksp=ksp, wv=wv, mode=mode):
                return ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, mode)
            try:
                for i in range(3):
                    run(i)
            except RuntimeError:        # launch shape not built / K-split grid not resident: not a candidate
                continue
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(args.reps):
                    run(i)
            graphs[(tpw, ksp, wv, mode)] = g
        times = {k: [] for k in graphs}
        for _ in range(args.rounds):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / args.reps)
        for (tpw, ksp, wv, mode), ts in sorted(times.items(), key=lambda kv: np.median(kv[1])):
            us = float(np.median(ts))
            print(json.dumps({"model": args.model, "tp": args.tp, "linear": name, "K": K, "N": sum(sizes), "rows": args.rows, "tpw": tpw,
                              "ksplit": ksp, "waves": wv, "mode": mode, "order": packs[0].wq_order, "us": round(us, 3), "min_us": round(min(ts), 3), "GBps": round(nb / us / 1e3, 1),
                              "frac": round(nb / us / 1e3 / 8000, 4)}), flush=True)
        del graphs, packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""MoE expert prefill: the grouped route (rotate once, device-side sort, one grouped W4A16 GEMM per projection) against the
round-2 per-expert host loop, same synthetic experts (Qwen3-30B-A3B-like block: hidden 2048, expert intermediate 768).
    python tools/bench_moe.py [--experts 64] [--topk 8] [--tokens 512,4096] > profiles/rNN_moe_prefill.jsonl"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import paro_oracle as po  # noqa: E402  (synthetic checkpoint-format experts only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--experts", type=int, default=64)
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--inter", type=int, default=768)
    ap.add_argument("--tokens", default="1,4,512,4096")
    args = ap.parse_args()
    import paroquant_amd  # noqa: F401
    from paroquant_amd.moe import ParoMoEExperts
    dev = torch.device("cuda:0")
    E, H, I, k = args.experts, args.hidden, args.inter, args.topk
    experts, rot = po.make_moe(1, E, H, I)
    tensors = {f"{e}.{proj}.{name}": torch.from_numpy(stack[e]) for proj, d in experts.items() for name, stack in d.items() for e in range(E)}
    tensors.update({n: torch.from_numpy(v) for n, v in rot.items()})
    moe = ParoMoEExperts(tensors, E, dev)
    rng = np.random.default_rng(0)
    for T in [int(t) for t in args.tokens.split(",")]:
        x = torch.from_numpy(rng.standard_normal((T, H)).astype(np.float16)).to(dev)
        idx = torch.from_numpy(np.stack([rng.choice(E, size=k, replace=False) for _ in range(T)]).astype(np.int64)).to(dev)

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(reps):
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        if T * k <= 64:      # decode: the (token, expert) slots of paro_w4a16_gemv_experts, two launches, in a HIP graph
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                moe(x, idx)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                for _ in range(50):
                    moe(x, idx)
            t = timed(g.replay) / 50
            wbytes = T * k * (H * 2 * I + I * H) / 2           # INT4 weights of the selected experts
            print(json.dumps({"experts": E, "topk": k, "hidden": H, "inter": I, "tokens": T, "route": "decode slots (2 launches)",
                              "us_per_moe_block": round(t * 1e3, 2), "GBps_int4_weights": round(wbytes / t / 1e6, 1)}), flush=True)
            continue
        t_grouped = timed(lambda: moe(x, idx))
        t_loop = timed(lambda: moe.per_expert_prefill(x, idx))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            moe(x, idx)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            moe(x, idx)
        t_graph = timed(g.replay)
        flops = 2.0 * T * k * (H * 2 * I + I * H)
        print(json.dumps({"experts": E, "topk": k, "hidden": H, "inter": I, "tokens": T, "ms_grouped": round(t_grouped, 3), "ms_grouped_graph": round(t_graph, 3),
                          "ms_per_expert_loop": round(t_loop, 3), "speedup": round(t_loop / t_graph, 2), "TFLOPs_grouped_graph": round(flops / t_graph / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Does a concurrent weight-prefetch branch in the captured decode step pay?  

Builds the bench.py decode stack (all quantised linears of a model, distinct weights per layer, M = 1), captures the
step in a HIP graph in several variants and times them INTERLEAVED:
   plain            the bench.py step
   pf(L, W)         at the start of layer k a side stream touches the packed weights of layer k + L with W workgroups
                    (ops.prefetch -> paro_prefetch), joined back at the end of the step
    python tools/prefetch_probe.py [--workload qwen3-4b] [--layers 0] [--rounds 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd import ops


def capture(stack, dev, look, wgs, per_linear):
    side = torch.cuda.Stream(dev)

    def step(x):
        h = x
        main = torch.cuda.current_stream(dev)
        n = len(stack.layers)
        for k, (qkv, o, gu, down) in enumerate(stack.layers):
            if look > 0 and k + look < n:
                tgt = stack.layers[k + look]
                if per_linear:
                    pass
                else:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        ops.prefetch([t for pk in tgt for t in pk.stream_buffers()], wgs)
            a = qkv.apply(h)[:, : stack.q_local]
            if look > 0 and per_linear and k + look < n:
                # one small prefetch per linear, forked after each GEMV launch: the branch can only start once the
                # previous kernel has been issued, which spreads the prefetch traffic over the layer
                for j, pk in enumerate(stack.layers[k + look]):
                    pass
            h = o.apply(a)
            d = gu.apply(h)[:, : stack.inter_local]
            h = down.apply(d)
        if look > 0:
            main.wait_stream(side)
        return h

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(stack.x)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step(stack.x)
    return g, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="qwen3-4b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variants", default="0:0,1:32,1:64,1:256,2:64,3:64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stack = bench.DecodeStack(args.workload, dev, n_layers=args.layers or None)
    ref = stack.step(stack.x).clone()
    graphs = {}
    for spec in args.variants.split(","):
        look, wgs = (int(v) for v in spec.split(":"))
        g, out = capture(stack, dev, look, wgs, False)
        g.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(out, ref), f"prefetch changed the result ({spec})"
        graphs[spec] = g
    times = {k: [] for k in graphs}
    for _ in range(args.rounds):
        for k, g in graphs.items():
            g.replay()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                g.replay()
            e1.record()
            torch.cuda.synchronize(dev)
            times[k].append(e0.elapsed_time(e1) / args.steps)
    for k, ts in times.items():
        med = float(np.median(ts))
        print(json.dumps({"workload": args.workload, "layers": stack.n_layers, "lookahead:wgs": k, "ms_per_step": round(med, 4),
                          "ms_min": round(float(np.min(ts)), 4), "tokens_per_s": round(1e3 / med, 1),
                          "GBps": round(stack.bytes_per_step / med / 1e6, 1),
                          "frac_8TBps": round(stack.bytes_per_step / med / 1e6 / 8000.0, 4)}), flush=True)


if __name__ == "__main__":
    main()

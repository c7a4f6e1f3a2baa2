#!/usr/bin/env python3
"""Mid-M sweep of the MFMA GEMM (round 6, session 3): per linear and row count, us per call of `apply` (the automatic route) against GEMM
variant 4 under every (row tiles per block, K-split) -- PARO_GEMM4_TUNE=1 makes the library read PARO_GEMM4_RT / PARO_GEMM4_KS at each
call, so one process covers the grid.  HIP graph of `reps` calls cycling distinct weight copies; pre-pass included.
    PARO_GEMM4_TUNE=1 python tools/sweep_gemm4.py [--model llama3-8b] [--rows 128,256,512,1024,2048]"""
import argparse, json, os, sys
os.environ.setdefault("PARO_GEMM4_TUNE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import alg_bytes, layer_shapes, synth_packed
from paroquant_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--rows", default="128,256,512,1024,2048")
    ap.add_argument("--rts", default="2,4,8")
    ap.add_argument("--kss", default="1,2,3,4,6,8")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.get_workspace(dev, 2 << 30)     # one buffer for every configuration (captured graphs keep its address)
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    for name, K, sizes, _ in layer_shapes(a.model):
        nb = alg_bytes(K, sum(sizes), len(sizes))
        copies = max(2, min(8, int((1 << 29) // nb) + 1))
        packs = [synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        for pk in packs:
            pk.prepare_prefill(torch.float16)
        for rows in [int(r) for r in a.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            cfgs = [("auto", 0, 0)] + [("v4", rt, ks) for rt in [int(v) for v in a.rts.split(",")] for ks in [int(v) for v in a.kss.split(",")]
                                       if rt * 32 <= max(64, ((rows + 31) // 32) * 32) and ks <= K // 256] + [("auto", 0, 1)]   # (auto again, LAST: the first graph of a round runs on colder caches)
            graphs = {}
            for kind, rt, ks in cfgs:
                os.environ["PARO_GEMM4_RT"], os.environ["PARO_GEMM4_KS"] = (str(rt), str(ks)) if kind == "v4" else ("0", "0")
                fn = (lambda i: packs[i % copies].apply(x)) if kind == "auto" else (lambda i: ops.w4a16_gemm_forced(x, packs[i % copies], None, True, 4))
                try:
                    fn(0); torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for i in range(a.reps):
                            fn(i)
                    graphs[(kind, rt, ks)] = g
                except Exception as e:
                    print(json.dumps({"linear": name, "rows": rows, "cfg": [kind, rt, ks], "error": str(e)[:100]}), flush=True)
            os.environ["PARO_GEMM4_RT"], os.environ["PARO_GEMM4_KS"] = "0", "0"
            ts = {k: [] for k in graphs}
            for _ in range(a.rounds):
                for k, g in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    ts[k].append(e0.elapsed_time(e1) * 1e3 / a.reps)
            res = sorted(((float(np.median(v)), k) for k, v in ts.items()))
            flops = 2.0 * rows * K * sum(sizes)
            print(json.dumps({"model": a.model, "linear": name, "K": K, "N": sum(sizes), "rows": rows,
                              "auto_us": round(min(t for t, k in res if k[0] == "auto"), 2), "auto_first_us": round([t for t, k in res if k == ("auto", 0, 0)][0], 2),
                              "best": [[k[1], k[2], round(t, 2)] for t, k in res if k[0] == "v4"][:4],
                              "best_TFLOPs": round(flops / res[0][0] / 1e6, 1)}), flush=True)
        del packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

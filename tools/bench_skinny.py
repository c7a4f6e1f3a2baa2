#!/usr/bin/env python3
"""17..64 rows per linear (round 6, session 3): us per call inside a HIP graph of `reps` calls cycling >= 1 GiB of distinct weights, for
  gemv  = schedule pre-pass + the GEMV on fragment-order x (mode 1; 2 row tiles up to 32 rows, 4 up to 64 -- tiles per wave <= 2 there),
  gemm  = schedule pre-pass (plain rows) + MFMA GEMM variant auto (64- / 128-row blocks, K-split + reduce launch),
  apply = what paro_w4a16_linear picks.
PARO_PREROT_SCHED=0 in the environment restores the stage-kernel pre-pass (rounds 1..5) for both.  Results: profiles/r06_skinny_routes.jsonl.
    python tools/bench_skinny.py [--model qwen3-4b] [--rows 17,24,32,33,48,64]"""
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
    ap.add_argument("--rows", default="17,24,32,33,48,64")
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--knobs", default="0,0,0", help="GEMV tiles_per_wave,ksplit,waves (0 = auto)")
    a = ap.parse_args()
    tpw, ksp, wv = [int(v) for v in a.knobs.split(",")]
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    for name, K, sizes, _ in bench.layer_shapes(a.model):
        N = sum(sizes)
        nb = bench.alg_bytes(K, N, len(sizes))
        copies = max(2, min(48, int((1 << 30) // nb) + 1))
        packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        for rows in [int(r) for r in a.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            out = {"model": a.model, "linear": name, "K": K, "N": N, "rows": rows}
            fns = {"gemv": lambda i: ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, 1),
                   "gemm": lambda i: ops.w4a16_gemm_forced(x, packs[i % copies], None, False),
                   "apply": lambda i: packs[i % copies].apply(x)}
            graphs, ys = {}, {}
            for k, fn in fns.items():
                try:
                    ys[k] = fn(0).clone()
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for i in range(a.reps):
                            fn(i)
                    graphs[k] = g
                except Exception as e:
                    out[k + "_error"] = str(e)[:100]
            ts = {k: [] for k in graphs}
            for _ in range(a.rounds):
                for k, g in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    ts[k].append(e0.elapsed_time(e1) * 1e3 / a.reps)
            for k in graphs:
                out[k + "_us"] = round(float(np.median(ts[k])), 2)
            if "gemv" in ys and "gemm" in ys:
                out["max_rel_diff"] = round(float((ys["gemv"].float() - ys["gemm"].float()).abs().max() / ys["gemm"].float().abs().max()), 5)
            ops.check_workspace(packs[0].workspace)
            print(json.dumps(out), flush=True)
        del packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Prefill-path microbench: fused linear at large M (rotate pre-pass + W4A16 MFMA GEMM), reporting
TFLOP/s (2*M*K*N).  `ms_rotate_prepass` times the STAGE kernel (torch.ops.rotation.rotate per partition)
as a stand-alone reference; the fused linear itself uses the dense MFMA pre-pass, which is faster, so
`TFLOPs_gemm_only` is an upper estimate -- use rocprofv3 --kernel-trace for the real split.
    python tools/bench_gemm.py [--model llama3-8b] [--rows 8192] [--reps 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import layer_shapes, synth_packed
from paroquant_amd import ops

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 (MI355X_MICROARCH.md)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--rows", default="2048,8192")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="f16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    for name, K, sizes, _ in layer_shapes(args.model):
        pk = synth_packed(K, sizes, dev, gen)
        for rows in [int(r) for r in args.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float32, generator=gen).to(dt)
            ms = timed(lambda: pk.apply(x), args.reps)
            ms_rot = timed(lambda: [torch.ops.rotation.rotate(x, pk.pairs[p], pk.theta[p], pk.channel_scales[p])
                                    for p in range(len(sizes))], args.reps)
            flops = 2.0 * rows * K * sum(sizes)
            print(json.dumps({"model": args.model, "linear": name, "M": rows, "K": K, "N": sum(sizes), "P": len(sizes),
                              "ms_total": round(ms, 4), "ms_rotate_prepass": round(ms_rot, 4),
                              "TFLOPs_total": round(flops / ms / 1e9, 1),
                              "TFLOPs_gemm_only": round(flops / max(ms - ms_rot, 1e-6) / 1e9, 1),
                              "mfma_util_total": round(flops / ms / 1e9 / MFMA_PEAK_TFLOPS, 4)}), flush=True)
            del x
        del pk; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

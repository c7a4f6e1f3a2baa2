#!/usr/bin/env python3
"""Prefill-path microbench: fused linear at large M (rotate pre-pass + W4A16 MFMA GEMM), TFLOP/s = 2*M*K*N / time.

    python tools/bench_gemm.py [--model llama3-8b] [--rows 2048,8192,65536] [--variants 0,1,2,4] [--dtype f16] [--rounds 3]

Variants (include/paro_abi.h) are timed INTERLEAVED in one process, `rounds` rounds of `reps` calls each, and the
median + min are reported (cdna guide rule 24: perf deltas come from within-probe interleaved rounds).
`ms_total` includes the rotate pre-pass (dense MFMA pre-pass for >= 256 rows); `ms_prepass` times that pre-pass
alone (rotation matrices through the same kernel, via a GEMM call on a 16-column dummy layer is not possible, so
it is measured as variant time minus the GEMM kernel time from rocprofv3 when profiling -- here it is simply the
stand-alone stage-kernel time as an upper bound)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import layer_shapes, synth_packed
from paroquant_amd import ops

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 (MI355X_MICROARCH.md)


def time_once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--rows", default="2048,8192")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--variants", default="0", help="comma list of GEMM variants (0 = the dispatcher's choice)")
    ap.add_argument("--only", default="")
    ap.add_argument("--fill", default="randn", choices=["randn", "zero"], help="activation fill (DVFS check: zeros clock higher)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    variants = [int(v) for v in args.variants.split(",")]
    bad = [v for v in variants if v not in (0, 1, 2, 4, 41, 42, 43, 44)]       # (variant 3 was removed in ABI v11)
    if bad:
        raise SystemExit(f"--variants: unknown GEMM variant(s) {bad}; the library builds 0 (auto), 1, 2 and 4")
    import bench
    if args.model in bench.HYBRID:       # Qwen3.5 family: the distinct linears of both layer kinds (delta-net block first, then the full-attention block's)
        shapes, seen = [], set()
        for full in (False, True):
            for name, K, sizes, kind in bench.hybrid_layer_shapes(args.model, full):
                if (K, tuple(sizes)) not in seen:
                    seen.add((K, tuple(sizes)))
                    shapes.append((name, K, sizes, kind))
    else:
        shapes = layer_shapes(args.model)
    for name, K, sizes, _ in shapes:
        if args.only and name not in args.only.split(","):
            continue
        pk = synth_packed(K, sizes, dev, gen)
        pk.prepare_prefill(dt)
        for rows in [int(r) for r in args.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float32, generator=gen).to(dt)
            if args.fill == "zero":
                x.zero_()
            fns = {}
            for v in variants:
                if v == 2 and dt != torch.float16:       # variant 2 keeps exact fp16 weights in registers: f16 only
                    continue
                fns[v] = (lambda v=v: pk.apply(x)) if v == 0 else (lambda v=v: ops.w4a16_gemm_forced(x, pk, variant=v))
            for fn in fns.values():           # warm-up (also builds lazy state)
                fn()
            torch.cuda.synchronize()
            times = {v: [] for v in fns}
            for _ in range(args.rounds):
                for v, fn in fns.items():
                    times[v].append(time_once(fn, args.reps))
            flops = 2.0 * rows * K * sum(sizes)
            for v, ts in times.items():
                med, best = float(np.median(ts)), float(np.min(ts))
                print(json.dumps({"model": args.model, "linear": name, "dtype": args.dtype, "fill": args.fill, "variant": v, "M": rows, "K": K,
                                  "N": sum(sizes), "P": len(sizes), "ms_total": round(med, 4), "ms_best": round(best, 4),
                                  "TFLOPs_total": round(flops / med / 1e9, 1), "TFLOPs_best": round(flops / best / 1e9, 1),
                                  "mfma_util_total": round(flops / med / 1e9 / MFMA_PEAK_TFLOPS, 4)}), flush=True)
            del x
        del pk; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

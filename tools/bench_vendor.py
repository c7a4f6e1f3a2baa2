#!/usr/bin/env python3
"""Context for the roofline fractions: what the VENDOR's dense kernels reach on the same box, same shapes, same random data.

    python tools/bench_vendor.py [--model llama3-8b] [--rows 1,8192,65536] [--dtype f16]

For every linear shape of the model: torch.matmul(x[M,K], W[K,N]) on DENSE fp16/bf16 weights (hipBLASLt / rocBLAS behind
torch), interleaved with this repo's fused rotate + INT4 linear on the same activations.  M = 1: GB/s over the bytes each
kernel has to read (dense: 2 K N, INT4: K N / 2 + scales/zeros + rotation); M >= 256: TFLOP/s = 2 M K N / time.  The dense
kernels move 4x the bytes at M = 1 and do no rotation / dequant at large M, so this is not a like-for-like race: it shows how
much of the gap to the paper peaks (8 TB/s, 2.5 PFLOP/s) is the machine (launch boundary, power limit on random operands)
and how much is this kernel."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import layer_shapes, synth_packed, alg_bytes


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
    ap.add_argument("--rows", default="1,8192,65536")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    for name, K, sizes, _ in layer_shapes(args.model):
        N = sum(sizes)
        pk = synth_packed(K, sizes, dev, gen)
        pk.prepare_prefill(dt)
        w = (torch.randn(K, N, device=dev, dtype=torch.float32, generator=gen) * 0.02).to(dt)
        for rows in [int(r) for r in args.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float32, generator=gen).to(dt)
            y = torch.empty(rows, N, device=dev, dtype=dt)
            reps = 200 if rows <= 16 else 3
            fns = {"dense_vendor": lambda: torch.matmul(x, w, out=y), "paro_int4": lambda: pk.apply(x)}
            if rows <= 16:   # decode: time inside a HIP graph like bench.py does (launch overhead of the eager call is not the kernel's)
                graphs = {}
                for k, fn in fns.items():
                    fn(); torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        fn()
                        with torch.cuda.graph(g, stream=s):
                            for _ in range(20):
                                fn()
                    graphs[k] = g
                torch.cuda.synchronize()
                fns = {k: (lambda g=g: g.replay()) for k, g in graphs.items()}
                per = 20
                reps = 10
            else:
                per = 1
                for fn in fns.values():
                    fn()
            torch.cuda.synchronize()
            times = {k: [] for k in fns}
            for _ in range(args.rounds):
                for k, fn in fns.items():
                    times[k].append(time_once(fn, reps) / per)
            for k, ts in times.items():
                med = float(np.median(ts))
                rec = {"model": args.model, "linear": name, "dtype": args.dtype, "kernel": k, "M": rows, "K": K, "N": N, "us": round(med * 1e3, 2)}
                if rows <= 16:
                    nbytes = 2.0 * K * N if k == "dense_vendor" else float(alg_bytes(K, N, len(sizes)))
                    rec.update({"bytes": int(nbytes), "GBps": round(nbytes / med / 1e6, 1), "frac_of_8TBps": round(nbytes / med / 1e6 / 8000.0, 3)})
                else:
                    fl = 2.0 * rows * K * N
                    rec.update({"TFLOPs": round(fl / med / 1e9, 1), "frac_of_2500TF": round(fl / med / 1e9 / 2500.0, 3)})
                print(json.dumps(rec), flush=True)
            del x, y
        del pk, w; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

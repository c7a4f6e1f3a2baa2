#!/usr/bin/env python3
"""Determinism stress of the 17..64-row route (round 6, session 3): schedule pre-pass (rotate.hip prerot_sched_kernel, x handed over in
MFMA-fragment order) + the GEMV on 2 / 4 MFMA row tiles with its in-launch K-split, and of the mid-M GEMM (64- / 128-row blocks + reduce
launch): `iters` eager calls and `replays` replays of a HIP graph of 20 calls through `apply` must return the bits of the first call;
mode 1 must equal mode 0 (rotation inside every workgroup) on the same launch shape where mode 0 exists (<= 16 rows).
    python tools/stress_skinny.py [--iters 300] [--replays 30]        ->  profiles/r06_stress_skinny.txt"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paroquant_amd import ops

CASES = [(2560, [4096, 1024, 1024], 17), (2560, [4096, 1024, 1024], 32), (4096, [2560], 24), (9728, [2560], 32), (2560, [9728, 9728], 32),
         (4096, [4096], 48), (4096, [4096, 1024, 1024], 64), (14336, [4096], 33), (4096, [14336, 14336], 40), (4096, [4096], 100), (9728, [2560], 384)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--replays", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(23)
    ops.get_workspace(dev, 1 << 30)
    out = {"cases": [], "mismatches": 0, "calls": 0}
    for K, sizes, rows in CASES:
        pk = bench.synth_packed(K, sizes, dev, gen)
        x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
        y0 = pk.apply(x).clone()
        bad = 0
        for _ in range(a.iters):
            bad += int(not torch.equal(pk.apply(x), y0))
        g = torch.cuda.CUDAGraph()
        ys = []
        with torch.cuda.graph(g):
            for _ in range(20):
                ys.append(pk.apply(x))
        for _ in range(a.replays):
            g.replay(); torch.cuda.synchronize()
            bad += sum(int(not torch.equal(y, y0)) for y in ys)
        case = {"case": f"{K}:{'+'.join(map(str, sizes))}:{rows}", "mismatches": bad, "finite": bool(torch.isfinite(y0.float()).all())}
        x16 = x[:16].contiguous()
        ks = 1 if sum(sizes) // 16 >= 1024 else 2      # (a K-split grid must be resident at once)
        case["mode1_equals_mode0_at_16_rows"] = bool(torch.equal(ops.w4a16_gemv_tuned(x16, pk, 2, ks, 8, 1), ops.w4a16_gemv_tuned(x16, pk, 2, ks, 8, 0)))
        ops.check_workspace(pk.workspace)
        out["cases"].append(case)
        out["mismatches"] += bad
        out["calls"] += a.iters + 20 * a.replays
    print(json.dumps(out))


if __name__ == "__main__":
    main()

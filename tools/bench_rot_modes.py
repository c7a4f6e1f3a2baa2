#!/usr/bin/env python3
"""A/B of the GEMV's rotation modes per linear and row count (round 6): us per launch inside a HIP graph of `reps` launches cycling >= 1 GiB
of distinct weights.  mode -1 = what `apply` picks (PARO_SHARED_ROT_MIN_ROWS=17 in the environment: the rules before mode 3), 0 = rotation
replicated in every workgroup, 1 = stage-kernel pre-pass + the GEMV on rotated x, 3 = the rotation shared inside the launch (producer
workgroups in front of the grid, {two channels, launch tag} granules; automatic from 5 rows; an EXPLICIT mode 3 keeps the caller's /
the rule tree's tiles per wave and falls back to mode 0 when that grid is not resident at once).  Also checks that mode 3 returns mode 0's
bits.  Results of the final build of round 6: profiles/r06_shared_rot_ab.jsonl (first cut: r06_shared_rot_cut1.jsonl), NOTES 6.2.
    python tools/bench_rot_modes.py [--model qwen3-4b] [--rows 1,2,4,8,16] [--modes=-1,0,1,3]"""
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
    ap.add_argument("--rows", default="1,2,4,8,16")
    ap.add_argument("--modes", default="-1,0,1,3")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--knobs", default="0,0,0", help="tiles_per_wave,ksplit,waves (0 = auto)")
    a = ap.parse_args()
    tpw, ksp, wv = [int(v) for v in a.knobs.split(",")]
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    modes = [int(m) for m in a.modes.split(",")]
    for name, K, sizes, _ in bench.layer_shapes(a.model):
        N = sum(sizes)
        nb = bench.alg_bytes(K, N, len(sizes))
        copies = max(2, min(48, int((1 << 30) // nb) + 1))
        packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        for rows in [int(r) for r in a.rows.split(",")]:
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            out = {"model": a.model, "linear": name, "K": K, "N": N, "rows": rows}
            ys = {}
            graphs = {}
            for m in modes:
                try:
                    fn = (lambda i, m=m: packs[i % copies].apply(x)) if m == -1 else (lambda i, m=m: ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, m))
                    ys[m] = fn(0).clone()
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for i in range(a.reps):
                            fn(i)
                    graphs[m] = g
                except Exception as e:
                    out[f"mode{m}_error"] = str(e)[:120]
            ts = {m: [] for m in graphs}
            for _ in range(a.rounds):
                for m, g in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    ts[m].append(e0.elapsed_time(e1) * 1e3 / a.reps)
            for m in graphs:
                out[f"mode{m}_us"] = round(float(np.median(ts[m])), 2)
            if 0 in ys and 3 in ys:
                out["mode3_equals_mode0"] = bool(torch.equal(ys[0], ys[3]))
                out["finite"] = bool(torch.isfinite(ys[3].float()).all())
            ops.check_workspace(packs[0].workspace)
            print(json.dumps(out), flush=True)
        del packs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

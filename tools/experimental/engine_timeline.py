#!/usr/bin/env python3
"""Per-edge timeline of the persistent decode engine (paro_engine_trace): where a phase's time goes.

    python tools/engine_timeline.py [--model qwen3-4b] [--layers 4] [--reps 5]

For every phase kind of the model's decoder layer (qkv, o, gate_up, down) prints, in microseconds after the LAST compute unit has
published the previous phase's outputs (the edge's time zero), the median / latest over the compute units of:
  got     the rotating wave has all partial sums of its group          (hop 1: store flight + poll)
  rotpub  it has published the rotated group                           (8 Givens stages, LDS transpose, store)
  gath    a CU has gathered its groups' rotated x into LDS             (hop 2)
  b1      it is past the first barrier (every wave of the CU has gathered)
  w0units wave 0 has consumed its units;   units  ... and staged its partial sums;   b2  the CU is past the second barrier
  pub     it has published its outputs = the next edge's time zero     (phase duration)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd.engine import DecodeEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stack = bench.DecodeStack(args.model, dev, n_layers=args.layers, route="fused")
    flat = [pk for lay in stack.layers for pk in lay]
    eng = DecodeEngine(flat)
    names = [n for n, _, _, _ in stack.shapes]
    for _ in range(3):
        eng(stack.x)
    torch.cuda.synchronize()
    acc = {}
    for rep in range(args.reps):
        tr = eng.trace(stack.x).cpu().numpy().astype(np.int64)        # [phases, cus, 8], 10 ns ticks
        n_ph = tr.shape[0]
        for p in range(1, n_ph):
            t0 = tr[p - 1, :, 7].max()
            kind = names[p % len(names)]
            nt = eng._descs[p].K // 128 * eng._descs[p].n_parts
            rot = slice(0, min(nt, tr.shape[1]))
            row = {"got_med": np.median(tr[p, rot, 1]) - t0, "got_max": tr[p, rot, 1].max() - t0,
                   "rotpub_med": np.median(tr[p, rot, 2]) - t0, "rotpub_max": tr[p, rot, 2].max() - t0,
                   "gath_med": np.median(tr[p, :, 4]) - t0, "gath_max": tr[p, :, 4].max() - t0,
                   "b1_med": np.median(tr[p, :, 5]) - t0, "b1_max": tr[p, :, 5].max() - t0,
                   "w0units_med": np.median(tr[p, :, 8]) - t0,
                   "units_med": np.median(tr[p, :, 6]) - t0, "units_max": tr[p, :, 6].max() - t0,
                   "b2_med": np.median(tr[p, :, 9]) - t0, "b2_max": tr[p, :, 9].max() - t0,
                   "pub_med": np.median(tr[p, :, 7]) - t0, "pub_max": tr[p, :, 7].max() - t0,
                   "entered_med": np.median(tr[p, :, 3]) - t0,
                   "u0_med": np.median(tr[p, :, 10]) - t0, "u1_med": np.median(tr[p, :, 11]) - t0,
                   # shader clocks per 10 ns tick between two far-apart events of one CU (2.4 GHz = 24): is the chip clocked down while it polls?
                   "clk_per_tick": float(np.median((tr[p, :, 16 + 7] - tr[p, :, 16 + 3]) / np.maximum(tr[p, :, 7] - tr[p, :, 3], 1))) * 100.0,
                   "unit_cycles": float(np.median(tr[p, :, 16 + 11] - tr[p, :, 16 + 10])) * 100.0,
                   "b1_to_u0_cycles": float(np.median(tr[p, :, 16 + 10] - tr[p, :, 16 + 5])) * 100.0,
                   "u1_to_staged_cycles": float(np.median(tr[p, :, 16 + 6] - tr[p, :, 16 + 11])) * 100.0,
                   "b2_to_pub_cycles": float(np.median(tr[p, :, 16 + 7] - tr[p, :, 16 + 9])) * 100.0}
            for k, v in row.items():
                acc.setdefault(kind, {}).setdefault(k, []).append(float(v) * 0.01)      # -> microseconds
    for kind in names:
        if kind in acc:
            print(json.dumps({"model": args.model, "phase": kind, "split": eng.describe()[names.index(kind)][0],
                              **{k: round(float(np.median(v)), 2) for k, v in acc[kind].items()}}), flush=True)
    # whole-chain time per phase from events (the production kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            eng(stack.x)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(json.dumps({"model": args.model, "layers": args.layers, "us_per_phase": round(e0.elapsed_time(e1) * 1e3 / 20 / len(flat), 3),
                      "us_per_layer": round(e0.elapsed_time(e1) * 1e3 / 20 / args.layers, 2)}), flush=True)


if __name__ == "__main__":
    main()

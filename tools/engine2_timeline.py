#!/usr/bin/env python3
"""Per-edge timeline of the loader / consumer engine (paro_engine2_trace, csrc/engine2.hip): where a phase's time goes.

    python tools/engine2_timeline.py [--model qwen3-4b] [--layers 4] [--reps 5] [--split 0,0,0,0]

For every phase kind of the model's decoder layer (qkv, o, gate_up, down) prints, in microseconds after the LAST compute unit has
published the previous phase's outputs (the edge's time zero), the median / latest over the compute units of consumer wave 0's stamps:
  enter   the wave is in the phase (its previous publish is behind it)
  got     all partial sums of its first batch of groups have arrived            (the hop: store flight + poll)
  rot     its groups are rotated and flagged in LDS                             (eight Givens stages)
  tile0   its first tile is accumulated                                         (waits for the loader's slot and the other waves' groups)
  tiles   its share of every slot is accumulated
  bar     every consumer of the CU has accumulated
  pub     the CU's outputs are published = the next edge's time zero            (phase duration = pub_max)
and of the loader: ld0 = first slot of the phase issued, ld1 = last slot issued (negative = ahead of the edge);
durations: w_poll = wave 0 in the hand-off poll, w_stream = wave 0 waiting for the loader's slots, w_ring = the loader waiting for a free slot.
The last line is the production kernel's whole-chain time per phase from a graph replay."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd.engine import DecodeEngine

EV = {"enter": 0, "got": 1, "rot": 2, "tile0": 3, "tiles": 4, "bar": 5, "pub": 6, "ld0": 8, "ld1": 9}
DUR = {"w_stream": 7, "w_ring": 10, "w_poll": 11}      # ticks spent waiting: consumer wave 0 for the loader's slots, the loader for a free slot, wave 0 in the hand-off poll


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--split", default="", help="K-chunks per phase kind, e.g. 3,3,4,3 (0 = the planner's)")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stack = bench.DecodeStack(args.model, dev, n_layers=args.layers, route="fused")
    flat = [pk for lay in stack.layers for pk in lay]
    names = [n for n, _, _, _ in stack.shapes]
    split = None
    if args.split:
        per = [int(v) for v in args.split.split(",")]
        split = [per[i % len(per)] for i in range(len(flat))]
    eng = DecodeEngine(flat, version=2, split=split)
    for _ in range(3):
        eng(stack.x)
    torch.cuda.synchronize()
    if not eng.status_ok():
        print(json.dumps({"model": args.model, "error": "a hand-off gave up"}), flush=True)
    acc = {}
    for rep in range(args.reps):
        tr = eng.trace(stack.x).cpu().numpy().astype(np.int64)        # [phases, cus, 16], 10 ns ticks
        n_ph = tr.shape[0]
        for p in range(1, n_ph):
            t0 = tr[p - 1, :, 6].max()
            kind = names[p % len(names)]
            busy = tr[p, :, 6] > 0                                      # CUs with work in this phase stamp
            row = {}
            for name, ix in EV.items():
                v = tr[p, busy, ix]
                v = v[v > 0]
                if v.size:
                    row[name + "_med"] = float(np.median(v) - t0)
                    row[name + "_max"] = float(v.max() - t0)
            for name, ix in DUR.items():
                v = tr[p, busy, ix]
                row[name + "_med"] = float(np.median(v))
                row[name + "_max"] = float(v.max())
            for k, v in row.items():
                acc.setdefault(kind, {}).setdefault(k, []).append(v * 0.01)      # -> microseconds
    desc = eng.describe()
    for i, kind in enumerate(names):
        if kind in acc:
            print(json.dumps({"model": args.model, "tag": args.tag, "phase": kind, "split": desc[i][0], "tiles_max": desc[i][1], "tiles_min": desc[i][2],
                              **{k: round(float(np.median(v)), 2) for k, v in acc[kind].items()}}), flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            eng(stack.x)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(json.dumps({"model": args.model, "tag": args.tag, "layers": args.layers, "split": args.split or "planner",
                      "us_per_phase": round(e0.elapsed_time(e1) * 1e3 / 20 / len(flat), 3),
                      "us_per_layer": round(e0.elapsed_time(e1) * 1e3 / 20 / args.layers, 2), "status_ok": eng.status_ok()}), flush=True)


if __name__ == "__main__":
    main()

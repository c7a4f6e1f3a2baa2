#!/usr/bin/env python3
"""Per-edge timeline of the loader / consumer engine (paro_engine2_trace, csrc/engine2.hip): where a phase's time goes.

    python tools/engine2_timeline.py [--model qwen3-4b] [--layers 4] [--reps 5] [--split 0,0,0,0] [--dump qkv_proj]

The trace holds, per (phase, compute unit, wave), eight stamps of the 100 MHz counter.  Consumer waves (rows 1..):
  enter   the wave is in the phase                         poll    it starts to poll the hand-off (its own CU has published the previous phase)
  got     the partial sums of its first groups are there    rot     ... rotated and in LDS as A fragments
  tiles   its tiles are accumulated                          red / pub / ack   (the wave that arrived last) starts to add the waves' rows /
                                                                               has issued the granule stores / the stores have left the CU
and the loader (row 0): ld0 / ld1 first / last slot of the phase issued, ring = ticks waiting for a free slot, gate = ticks yielding to hand-offs.
Per phase kind, in microseconds after the LAST compute unit has issued the previous phase's stores (the edge's time zero): the median
over the compute units of each event of the FIRST and of the LAST wave to reach it, and the latest compute unit.
The last line is the production kernel's whole-chain time per phase from a graph replay."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from paroquant_amd.engine import DecodeEngine

EV = ["enter", "poll", "got", "rot", "tiles", "red", "pub", "ack"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--split", default="", help="K-chunks per phase kind, e.g. 3,3,4,3 (0 = the planner's)")
    ap.add_argument("--tag", default="")
    ap.add_argument("--dump", default="", help="phase kind: print the slowest compute units of that phase, wave by wave")
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
    tr = None
    for rep in range(args.reps):
        tr = eng.trace(stack.x).cpu().numpy().astype(np.int64)        # [phases, cus, 64] = [phases, cus, 8 waves, 8 events], 10 ns ticks
        tr = tr.reshape(tr.shape[0], tr.shape[1], 8, 8)
        n_ph = tr.shape[0]
        for p in range(1, n_ph):
            t0 = tr[p - 1, :, 1:, 6].max()
            kind = names[p % len(names)]
            cons = tr[p, :, 1:, :].astype(np.float64)                  # [cus, waves, events]
            cons[cons == 0] = np.nan
            busy = np.isfinite(cons[:, :, 6]).any(axis=1)
            row = {}
            with np.errstate(all="ignore"):
                for e, name in enumerate(EV):
                    v = cons[busy][:, :, e] - t0
                    if not np.isfinite(v).any():
                        continue
                    first, last = np.nanmin(v, axis=1), np.nanmax(v, axis=1)
                    row[name + "_first_med"] = float(np.nanmedian(first))
                    row[name + "_last_med"] = float(np.nanmedian(last))
                    row[name + "_last_max"] = float(np.nanmax(last))
            ld = tr[p, busy, 0, :].astype(np.float64)
            row["ld0_med"] = float(np.median(ld[:, 0]) - t0)
            row["ld1_med"] = float(np.median(ld[:, 1]) - t0)
            row["ring_med"] = float(np.median(ld[:, 2]))
            row["gate_med"] = float(np.median(ld[:, 3]))
            for k, v in row.items():
                acc.setdefault(kind, {}).setdefault(k, []).append(v * 0.01)      # -> microseconds
    if args.dump:
        k = names.index(args.dump)
        p = k + len(names) * (args.layers - 1)                # the last layer's phase of that kind
        t0 = tr[p - 1, :, 1:, 6].max()
        pubs = tr[p, :, 1:, 6].max(axis=1)
        order = np.argsort(-pubs)[:6]
        for c in list(order) + [int(np.argsort(pubs)[len(pubs) // 2])]:
            ldr = tr[p, c, 0, :]
            print(json.dumps({"cu": int(c), "pub": round(float(pubs[c] - t0) * 0.01, 2), "loader": {"ld0": round(float(ldr[0] - t0) * 0.01, 2), "ld1": round(float(ldr[1] - t0) * 0.01, 2),
                                                                                                   "ring_wait": round(float(ldr[2]) * 0.01, 2), "gate_wait": round(float(ldr[3]) * 0.01, 2)},
                              "loader_next_phase": {"ld0": round(float(tr[min(p + 1, tr.shape[0] - 1), c, 0, 0] - t0) * 0.01, 2), "ld1": round(float(tr[min(p + 1, tr.shape[0] - 1), c, 0, 1] - t0) * 0.01, 2)}}), flush=True)
            for w in range(1, 8):
                st = tr[p, c, w, :]
                if st[0] == 0:
                    continue
                print("   wave", w, " ".join(f"{n}={(st[e] - t0) * 0.01:6.2f}" if st[e] else f"{n}=   -  " for e, n in enumerate(EV)), flush=True)
    desc = eng.describe()
    for i, kind in enumerate(names):
        if kind in acc:
            print(json.dumps({"model": args.model, "tag": args.tag, "phase": kind, "split": desc[i][0], "ntile_max": desc[i][1], "ntile_min": desc[i][2],
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

#!/usr/bin/env python3
"""Decode-attention kernel alone: us per launch in a HIP graph of 50 launches (distinct KV caches per launch so
that nothing is cache-resident), for a few positions.   python tools/bench_attn.py [--heads 32 --kv 8 --hd 128]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from paroquant_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--heads", type=int, default=32); ap.add_argument("--kv", type=int, default=8); ap.add_argument("--hd", type=int, default=128)
ap.add_argument("--tmax", type=int, default=2048); ap.add_argument("--positions", default="0,100,255,256,700,2047")
ap.add_argument("--layers", type=int, default=36)
ap.add_argument("--split", action="store_true", help="paro_attn_decode_split: slots out, no in-launch merge")
a = ap.parse_args()
dev = torch.device("cuda:0")
H, KV, hd, T = a.heads, a.kv, a.hd, a.tmax
caches = [(torch.randn(KV, T, hd, device=dev).half(), torch.randn(KV, hd, T, device=dev).half()) for _ in range(a.layers)]
qkv = torch.randn((H + 2 * KV) * hd, device=dev).half()
half = hd // 2
inv = 1.0 / (1e6 ** (torch.arange(half, device=dev).float() * 2 / hd))
ang = torch.arange(T, device=dev).float()[:, None] * inv[None]
rope = torch.cat([ang.cos(), ang.sin()], -1).contiguous()
w = torch.ones(hd, device=dev).half()
out = torch.empty(H * hd, device=dev).half()
pos = torch.zeros(1, dtype=torch.int32, device=dev)
sp = torch.zeros(ops.attn_parts_floats(H, hd), dtype=torch.float32, device=dev) if a.split else None
def run():
    for k, v in caches:
        ops.attn_decode(qkv, k, v, pos, rope, H, KV, hd, w, w, 1e-6, out=out, split_out=sp)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
for p in [int(v) for v in a.positions.split(",") if int(v) < T]:
    pos.fill_(p)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / a.layers)
    print(json.dumps({"max_positions": T, "chunk": (64 if p < 256 else 128) if a.split else (256 if T <= 512 else 128), "split": bool(a.split), "pos": p, "us_per_launch": round(float(np.median(ts)), 2),
                      "dbg": os.environ.get("PARO_ATTN_DBG", "0")}), flush=True)

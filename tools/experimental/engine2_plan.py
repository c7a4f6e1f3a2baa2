#!/usr/bin/env python3
"""What the engine-2 planner chooses (host only, no GPU): python tools/engine2_plan.py [--model qwen3-4b] [--ncu 256] [--split 3,3,4,3]

Per linear of the model's decoder layer: K-chunks, groups per chunk, tiles per CU (min .. max), groups of the busiest wave."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from paroquant_amd import _native as nat

MODELS = {"qwen3-4b": [(2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560])],
          "llama3-8b": [(4096, [4096, 1024, 1024]), (4096, [4096]), (4096, [14336, 14336]), (14336, [4096])],
          "qwen3-0.6b": [(1024, [2048, 1024, 1024]), (2048, [1024]), (1024, [3072, 3072]), (3072, [1024])],
          "llama3-70b": [(8192, [8192, 1024, 1024]), (8192, [8192]), (8192, [28672, 28672]), (28672, [8192])]}
WORK = np.dtype([("s", "<i2"), ("p", "<i2"), ("g0", "<i2"), ("ng", "<i2"), ("t0", "<i4"), ("tz0", "<i4"), ("nt", "<i2"), ("pad0", "<i2"),
                 ("inv_nt", "<i4"), ("ntile", "<i4"), ("pad1", "<i4")])


def desc(K, sizes):
    d = nat.ParoLinearDesc()
    d.K, d.N, d.n_parts, d.krot, d.act_dtype, d.group_size = K, sum(sizes), len(sizes), 8, 1, 128
    for i, n in enumerate(sizes):
        d.part_cols[i] = n
    d.wq_order = 1 if d.N // 16 >= 1024 else 0
    for f in ("wq", "sz", "rot", "pairs", "theta", "channel_scales"):
        setattr(d, f, 0x1000)
    return d


def plan(lib, shapes, ncu, split=None):
    descs = [desc(*s) for s in shapes]
    ph = (nat.ParoEnginePhase * len(descs))()
    for i, d in enumerate(descs):
        ph[i].L, ph[i].in_col0 = ctypes.pointer(d), 0
        ph[i].flags = split[i % len(split)] if split else 0
    e = nat.ParoEngine()
    if lib.paro_engine2_plan(ph, len(descs), ncu, ctypes.byref(e)) != 0:
        raise RuntimeError(lib.paro_last_error())
    blob = np.zeros(e.plan_bytes, dtype=np.uint8)
    if lib.paro_engine2_build(ph, ctypes.byref(e), blob.ctypes.data_as(ctypes.c_void_p)) != 0:
        raise RuntimeError(lib.paro_last_error())
    return e, blob, descs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--ncu", type=int, default=256)
    ap.add_argument("--split", default="")
    ap.add_argument("--cons", type=int, default=3)
    a = ap.parse_args()
    lib = ctypes.CDLL(nat.lib_path())
    lib.paro_last_error.restype = ctypes.c_char_p
    shapes = MODELS[a.model]
    split = [int(v) for v in a.split.split(",")] if a.split else None
    e, blob, _ = plan(lib, shapes, a.ncu, split)
    n = len(shapes)
    phases = blob[: n * 128].reshape(n, 128)
    work = blob[n * 128:].view(WORK)
    for i, (K, sizes) in enumerate(shapes):
        S = int(phases[i, 88:92].view("<i4")[0])
        woff = int(phases[i, 104:108].view("<i4")[0])
        w = work[woff: woff + a.ncu]
        busy = w[w["ntile"] > 0]
        gpw = (busy["ng"] + a.cons - 1) // a.cons
        print(f"{a.model} K={K} N={sum(sizes)} P={len(sizes)}: S={S} CUs={len(busy)} ng={busy['ng'].min()}..{busy['ng'].max()} nt={busy['nt'].min()}..{busy['nt'].max()} "
              f"tiles/CU={busy['ntile'].min()}..{busy['ntile'].max()} (mean {busy['ntile'].mean():.1f}) busiest wave: {int((gpw * ((busy['nt'] + 1) // 2)).max())} tile pairs, "
              f"{int(((gpw + 2) // 3).max())} rotation batches")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Which phase of an engine-2 chain first disagrees with the per-call kernels (prefix chains of the Qwen3-4B layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import paro_oracle as po
from tests.test_gpu_parity import _np, _packed, _t
from paroquant_amd.engine import DecodeEngine

dev = torch.device("cuda:0")
shapes = [(2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560]), (2560, [4096, 1024, 1024])]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    shapes = [(1024, [2048, 1024, 1024]), (2048, [1024]), (1024, [3072, 3072]), (3072, [1024])] * 2
layers = []
for i, (K, sizes) in enumerate(shapes):
    L = po.make_layer(i, K, sizes)
    gain = 1.0 / (6.52 * np.sqrt(K) * np.sqrt(1.75) * np.sqrt(13.0 / 12.0)) / 0.011
    L["scales"] = (L["scales"].astype(np.float32) * gain).astype(np.float16)
    layers.append(L)
pks = [_packed(L, dev) for L in layers]
x = _t(np.random.default_rng(0).standard_normal((1, shapes[0][0])).astype(np.float32), dev, torch.float16)
cur = x
refs = []
for pk in pks:
    cur = pk.apply(cur[:, :pk.K].contiguous())
    refs.append(cur)
for n in range(1, len(pks) + 1):
    for split in (None, [1] * n, [2] * n):
        try:
            eng = DecodeEngine(pks[:n], dtype=torch.float16, version=2, split=split)
        except RuntimeError as e:
            print(n, split, "plan failed", str(e)[:60]); continue
        y = eng(x).clone(); torch.cuda.synchronize()
        y2 = eng(x).clone(); torch.cuda.synchronize()
        err = po.rel_err(_np(y), _np(refs[n - 1]))
        bad = (np.abs(_np(y) - _np(refs[n - 1])) > 0.05 * np.abs(_np(refs[n - 1])).max()).nonzero()[1]
        print(f"phases={n} split={split} planner={[d[0] for d in eng.describe()]} rel_err={err:.4g} ok={eng.status_ok()} same_twice={torch.equal(y, y2)} "
              f"bad_cols={len(bad)} first_bad={bad[:6].tolist()} last_bad={bad[-3:].tolist()}", flush=True)

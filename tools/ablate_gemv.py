#!/usr/bin/env python3
"""Phase ablation of the fused GEMV (PARO_GEMV_FLAGS bits) for fixed launch shapes; one process per
flag value because the flags are read once at library load.  Usage: python tools/ablate_gemv.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"o_proj": "1,1,16", "qkv_proj": "2,1,16", "gate_up_proj": "4,1,8", "down_proj": "4,4,16"}
FLAGS = [0, 1, 2, 3, 7, 11, 15, 16]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from bench import alg_bytes, layer_shapes, synth_packed
    from paroquant_amd import ops
    dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(3)
    for name, K, sizes, _ in layer_shapes("llama3-8b"):
        tpw, ksp, wv = [int(v) for v in CASES[name].split(",")]
        nb = alg_bytes(K, sum(sizes), len(sizes)); copies = max(2, min(48, int((1 << 30) // nb) + 1))
        packs = [synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        x = torch.randn(1, K, device=dev, dtype=torch.float16, generator=gen)
        for i in range(3): ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, 0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(100): ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, 0)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        print(json.dumps({"flags": int(os.environ.get("PARO_GEMV_FLAGS", "0")), "linear": name, "cfg": CASES[name],
                          "us": round(float(np.median(ts)), 2)}), flush=True)
        del packs, g; torch.cuda.empty_cache()
else:
    for f in FLAGS:
        env = dict(os.environ, PARO_GEMV_FLAGS=str(f))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)

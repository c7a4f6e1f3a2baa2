#!/usr/bin/env python3
"""Phase ablation of the fused GEMV with the diagnostic kernel builds (one process per variant: the
variant is read once at library load).  Needs `make -C paroquant_amd/csrc clean all DIAG=1`.

    python tools/ablate_gemv.py [--model llama3-8b]

PARO_GEMV_PD: 1 shipping kernel | 51 stages without the cross-lane fetch | 61 exchange through LDS memory
| 71 / 81 only 2 KiB / 1 KiB of the 3 KiB schedule fetched (stages run on garbage: timing only) | 41 schedule fetched, stages not run | 11 no schedule, no stages | 21 also no unpack / MFMA (pure stream)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"o_proj": "0,0,0", "qkv_proj": "0,0,0", "gate_up_proj": "0,0,0", "down_proj": "0,0,0"}   # the automatic launch shapes (what ships)
VARIANTS = [1, 51, 71, 81, 41, 11, 21]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from bench import alg_bytes, layer_shapes, synth_packed
    from paroquant_amd import ops
    dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(3)
    for name, K, sizes, _ in layer_shapes(sys.argv[2]):
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
        print(json.dumps({"pd": int(os.environ.get("PARO_GEMV_PD", "1")), "model": sys.argv[2], "linear": name,
                          "cfg": CASES[name], "us": round(float(np.median(ts)), 2)}), flush=True)
        del packs
        torch.cuda.empty_cache()
else:
    ap = argparse.ArgumentParser(); ap.add_argument("--model", default="llama3-8b"); a = ap.parse_args()
    for pd in VARIANTS:
        env = dict(os.environ, PARO_GEMV_PD=str(pd))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", a.model], env=env, check=False)

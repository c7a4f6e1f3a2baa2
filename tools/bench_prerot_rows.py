"""GEMV on ALREADY-ROTATED activations (mode 2) against the rotation replicated in every workgroup (mode 0), per linear and row count:\nus per launch in a HIP graph of 100 launches over >= 1 GiB of weights.  Round 6: the pre-rotated GEMV is flat in the rows -- all of the\nbatched-decode overhead is rotation (profiles/r06_prerot_rows.jsonl; what mode 3 was built on).   python tools/bench_prerot_rows.py"""
import sys, json, torch
sys.path.insert(0, '.')
import bench
from paroquant_amd import ops
ROWS = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 2, 4, 8, 16]     # (17..32 rows: python tools/bench_prerot_rows.py 16,17,24,32)
MODES = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 0]
dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(3)
for name, K, sizes, _ in bench.layer_shapes("qwen3-4b"):
    P = len(sizes)
    nb = bench.alg_bytes(K, sum(sizes), P); copies = max(2, min(48, int((1 << 30) // nb) + 1))
    packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
    for rows in ROWS:
        xr = torch.randn(P, rows, K, device=dev, dtype=torch.float16, generator=gen)
        x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
        out = {"linear": name, "rows": rows}
        for mode, inp in [(m, xr if m == 2 else x) for m in MODES]:
            if mode == 0 and rows > 16: continue
            try:
                for i in range(3): ops.w4a16_gemv_tuned(inp, packs[i % copies], 0, 0, 0, mode)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(100): ops.w4a16_gemv_tuned(inp, packs[i % copies], 0, 0, 0, mode)
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 100)
                out[f"mode{mode}_us"] = round(min(ts), 2)
            except Exception as e:
                out[f"mode{mode}_err"] = str(e)[:80]
        print(json.dumps(out), flush=True)
    del packs; torch.cuda.empty_cache()

import sys, json, itertools, torch
sys.path.insert(0, '/root/repo')
import bench
from paroquant_amd import ops
dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(3)
for model in ("qwen3-4b", "llama3-8b"):
    for name, K, sizes, _ in bench.layer_shapes(model):
        if len(sizes) != 1: continue
        nb = bench.alg_bytes(K, sum(sizes), 1); copies = max(2, min(48, int((1 << 30) // nb) + 1))
        packs = [bench.synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        x = torch.randn(1, 1, K, device=dev, dtype=torch.float16, generator=gen)
        for mode, tpw, ksp, wv in itertools.product((2,), (1, 2, 4), (1, 2, 4), (8, 16)):
            if tpw == 8 and wv == 16: continue
            try:
                for i in range(3): ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, mode)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(150): ops.w4a16_gemv_tuned(x, packs[i % copies], tpw, ksp, wv, mode)
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 150)
                print(json.dumps({"model": model, "linear": name, "mode": mode, "tpw": tpw, "ksplit": ksp, "waves": wv, "us": round(min(ts), 3)}), flush=True)
            except Exception as e:
                print(json.dumps({"model": model, "linear": name, "tpw": tpw, "ksplit": ksp, "waves": wv, "err": str(e)[:80]}), flush=True)
        del packs; torch.cuda.empty_cache()

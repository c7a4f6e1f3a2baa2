#!/usr/bin/env python3
"""Stress of GEMV mode 3 (the rotation shared inside the launch; automatic from 5 rows through `apply`): per case the same launch N times,
eagerly back to back AND from a HIP graph of 50 launches, every output compared BIT FOR BIT with the replicated rotation's (mode 0), the
workspace's status word checked at the end (a hand-over that gave up writes NaN + the status word, never a silent value).  Inputs
change every 25 iterations (a granule left by an earlier launch must not be taken); cases = the BASELINE decode shapes at 5 / 8 / 12 / 16
rows, K-split shapes, one ragged one.  Reference semantics: rotate -> INT4 linear per call, vllm/plugin.py:281-311.

    python tools/stress_shared_rot.py [iterations = 2000]
    python tools/stress_shared_rot.py --load        300 launches per case beside a second stream that keeps the CUs busy with matmuls of uneven
                                                    length (the producers are dispatched first: a consumer never waits for a workgroup behind it)"""
import sys, os, json, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from paroquant_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(17)
def auto_shape(pk, rows):
    import ctypes
    from paroquant_amd import _native as nat
    d = ops.pk_desc(pk, torch.float16)
    kn = [ctypes.c_int(v) for v in (0, 0, 0, -1)]
    nat.check(nat.load().paro_gemv_launch_shape(ctypes.byref(d), rows, *[ctypes.byref(k) for k in kn]))
    return kn[0].value, kn[1].value, kn[2].value


if len(sys.argv) > 1 and sys.argv[1] == "--load":
    res = []
    # a competing stream that keeps most CUs busy with matmuls of uneven length
    a = torch.randn(4096, 4096, device=dev, dtype=torch.float16); b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    s2 = torch.cuda.Stream(dev)
    res = []
    for K, sizes, rows in [(2560, [4096, 1024, 1024], 8), (9728, [2560], 12), (4096, [14336, 14336], 8)]:
        pk = bench.synth_packed(K, sizes, dev, gen)
        pk.bind_stream(torch.cuda.current_stream(dev)) if hasattr(pk, "bind_stream") else None
        bad = nan = 0
        for it in range(300):
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            ref = ops.w4a16_gemv_tuned(x, pk, *auto_shape(pk, rows), 0)
            torch.cuda.synchronize()
            with torch.cuda.stream(s2):
                for _ in range(1 + it % 3):
                    c = a @ b
            y = pk.apply(x)
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
                nan += int(torch.isnan(y.float()).any())
        try:
            ops.check_workspace(pk.workspace); st = "clean"
        except Exception as e:
            st = str(e)[:100]
        res.append({"K": K, "sizes": sizes, "rows": rows, "launches_beside_a_competing_stream": 300, "mismatches": bad, "with_nan": nan, "workspace_status": st})
        print(json.dumps(res[-1]), flush=True)
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = [(2560, [4096, 1024, 1024], 8), (4096, [2560], 5), (2560, [9728, 9728], 16), (9728, [2560], 12), (4096, [14336, 14336], 8),
         (14336, [4096], 8), (4096, [4096, 1024, 1024], 16), (1024, [2048, 1024, 1024], 16), (1536, [528], 7)]


out = []
t0 = time.time()
for K, sizes, rows in cases:
    pk = bench.synth_packed(K, sizes, dev, gen)
    x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
    bad = 0
    shape = auto_shape(pk, rows)          # mode 3 may run another launch shape than mode 0's rule tree: compare at ITS shape (same K-split = same bits)
    ref = ops.w4a16_gemv_tuned(x, pk, *shape, 0).clone()
    for it in range(N):
        if it % 25 == 24:
            x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
            ref = ops.w4a16_gemv_tuned(x, pk, *shape, 0).clone()
        y = pk.apply(x)                                   # the boundary's automatic route: mode 3 from 5 rows
        if not torch.equal(y, ref):
            bad += 1
    # graph: 50 launches per replay on a static input buffer that is rewritten between replays
    xs = x.clone()
    s = torch.cuda.Stream(dev); s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        pk.apply(xs)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ys = [pk.apply(xs) for _ in range(50)]
    gbad = 0
    for rep in range(max(1, N // 50)):
        xs.copy_(torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen))
        g.replay(); torch.cuda.synchronize()
        r = ops.w4a16_gemv_tuned(xs, pk, *shape, 0)
        gbad += sum(0 if torch.equal(y, r) else 1 for y in ys)
    try:
        ops.check_workspace(pk.workspace); status = "clean"
    except Exception as e:
        status = str(e)[:80]
    row = {"K": K, "sizes": sizes, "rows": rows, "eager_iterations": N, "eager_mismatches": bad, "graph_launches": 50 * max(1, N // 50),
           "graph_mismatches": gbad, "workspace_status": status}
    print(json.dumps(row), flush=True)
    out.append(row)
    del pk, g
    torch.cuda.empty_cache()
print(json.dumps({"total_eager_mismatches": sum(r["eager_mismatches"] for r in out), "total_graph_mismatches": sum(r["graph_mismatches"] for r in out),
                  "seconds": round(time.time() - t0, 1)}))

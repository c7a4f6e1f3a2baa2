#!/usr/bin/env python3
"""Determinism stress of the attention tail (ABI v18): the fused qkv + attention launch, N times eager and N times replayed from a HIP
graph (graphs of 50 launches), at several positions; every launch's slot outputs and KV-cache rows must equal, bit for bit, those of the
two-launch route (paro_w4a16_gemv_fused(parts_out) + paro_attn_decode_split).  Prints one JSON line.
    python tools/stress_attn_tail.py [N=2000]"""
import json, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paroquant_amd.decoder import ParoDecoderLM
from paroquant_amd import ops, _native as nat

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
t0 = time.time()
out = {"launches": 0, "eager_mismatches": 0, "graph_mismatches": 0}
for model in ("qwen3-4b", "llama3-8b"):
    lm = ParoDecoderLM.random(model, dev, n_layers=1, max_positions=1024)
    assert lm.fuse_qkv_attn
    c = lm.cfg; L = lm.layers[0]; pk = L.qkv; hidden = c.hidden
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(1, hidden, device=dev, dtype=torch.float16, generator=g)
    L.kcache.copy_(torch.randn(L.kcache.shape, device=dev, dtype=torch.float16, generator=g) * 0.3)
    L.vcache.copy_(torch.randn(L.vcache.shape, device=dev, dtype=torch.float16, generator=g) * 0.3)
    R = nat.PROLOGUE_RMSNORM
    for posv in (3, 130, 255, 700):
        pos = torch.tensor([posv], device=dev, dtype=torch.int32)
        pq4 = torch.zeros(pk.N + 1, 4, device=dev); pq8 = torch.zeros(pk.N + 1, 8, device=dev)
        sp_ref = torch.zeros_like(lm.attn_parts); sp = torch.zeros_like(lm.attn_parts)
        k_ref, v_ref = L.kcache.clone(), L.vcache.clone()
        ws_ref = torch.zeros_like(lm.attn_ws)
        ops.w4a16_gemv_fused(x, pk, R, c.rms_eps, parts_out=pq4)
        ops.attn_decode(pq4, k_ref, v_ref, pos, lm.rope, lm.nh, lm.nkv, c.head_dim, L.q_norm, L.k_norm, c.rms_eps, out=lm.attn_buf, workspace=ws_ref,
                        norm_dim=hidden, norm_eps=c.rms_eps, split_out=sp_ref)
        kc, vc = L.kcache.clone(), L.vcache.clone()
        tail = dict(kcache=kc, vcache=vc, pos=pos, rope=lm.rope, n_heads=lm.nh, n_kv_heads=lm.nkv, head_dim=c.head_dim, q_norm_w=L.q_norm, k_norm_w=L.k_norm,
                    eps=c.rms_eps, split_out=sp, workspace=lm.attn_ws)
        run = lambda: ops.w4a16_gemv_fused(x, pk, R, c.rms_eps, parts_out=pq8, attn_tail=tail)
        for i in range(N // 4):
            sp.zero_(); run()
            ok = torch.equal(sp, sp_ref) and torch.equal(kc, k_ref) and torch.equal(vc, v_ref)
            out["eager_mismatches"] += int(not ok); out["launches"] += 1
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(50):
                run()
        for i in range(max(1, N // 200)):
            sp.zero_(); gr.replay()
            ok = torch.equal(sp, sp_ref) and torch.equal(kc, k_ref) and torch.equal(vc, v_ref)
            out["graph_mismatches"] += int(not ok); out["launches"] += 50
    ops.check_workspace(pk.workspace)
    del lm; torch.cuda.empty_cache()
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))

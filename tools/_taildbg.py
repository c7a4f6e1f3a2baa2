import sys, torch
sys.path.insert(0, '.')
from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
from paroquant_amd import ops, _native as nat
dev = torch.device("cuda:0")
nh, nkv, hd, hidden, T = 16, 8, 128, 2048, 192
lm = ParoDecoderLM.random(DecoderConfig(hidden, 4096, nh, nkv, hd, 1, 640, 1e-6, 10000.0, True, T), dev, seed=11)
print("flags", lm.deferred, lm.deferred_qkv, lm.split_attn, lm.fuse_qkv_attn)
L = lm.layers[0]; pk = L.qkv; N = pk.N
print("shape", ops.gemv_parts_count(pk), pk.partition_sizes)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1, hidden, device=dev, dtype=torch.float16, generator=g)
for posv in (0, 5, 70):
    pos = torch.tensor([posv], device=dev, dtype=torch.int32)
    kc = torch.randn(nkv, T, hd, device=dev, dtype=torch.float16, generator=g) * 0.3
    vc = torch.randn(nkv, hd, T, device=dev, dtype=torch.float16, generator=g) * 0.3
    kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    pq4 = torch.zeros(N + 1, 4, device=dev); pq8 = torch.zeros(N + 1, 8, device=dev)
    sp1 = torch.zeros(ops.attn_parts_floats(nh, hd), device=dev); sp2 = torch.zeros_like(sp1)
    ws1 = torch.zeros_like(lm.attn_ws); ws2 = torch.zeros_like(lm.attn_ws)
    R = nat.PROLOGUE_RMSNORM
    ops.w4a16_gemv_fused(x, pk, R, 1e-6, parts_out=pq4)
    ops.attn_decode(pq4, kc1, vc1, pos, lm.rope, nh, nkv, hd, L.q_norm, L.k_norm, 1e-6, out=lm.attn_buf, workspace=ws1, norm_dim=hidden, norm_eps=1e-6, split_out=sp1)
    ops.w4a16_gemv_fused(x, pk, R, 1e-6, parts_out=pq8, attn_tail=dict(kcache=kc2, vcache=vc2, pos=pos, rope=lm.rope, n_heads=nh, n_kv_heads=nkv, head_dim=hd,
                         q_norm_w=L.q_norm, k_norm_w=L.k_norm, eps=1e-6, split_out=sp2, workspace=ws2))
    torch.cuda.synchronize()
    vals = pq8.view(N + 1, 4, 2)[:, :, 0]
    tags = pq8.view(torch.int32).view(N + 1, 4, 2)[:, :, 1]
    print("pos", posv, "granule values == parts:", torch.equal(vals, pq4), "max|d|", (vals - pq4).abs().max().item(), "tags uniq", torch.unique(tags).tolist()[:4])
    print("  kcache eq", torch.equal(kc1, kc2), "vcache eq", torch.equal(vc1, vc2), "split eq", torch.equal(sp1, sp2), "max|d| split", (sp1 - sp2).abs().max().item(),
          "nan", torch.isnan(sp2).any().item())
    no = nh * hd * 4
    d = (sp1[:no] - sp2[:no]).abs().view(nh, hd, 4)
    print("  per-head max diff", d.amax(dim=(1, 2)).tolist())
    print("  ml diff", (sp1[no:] - sp2[no:]).abs().view(nh, 8).amax(dim=1).tolist())
posv = 5
ssq = pq4[N].sum().item(); rstd = (ssq / hidden + 1e-6) ** -0.5
vexp = (pq4[(nh + nkv) * hd:(nh + nkv) * hd + 8].sum(1) * rstd)
print("expected v[0,:8]", vexp.tolist())
print("unfused  v", vc1[0, :8, 70].tolist())
print("fused    v", vc2[0, :8, 70].tolist())
kexp_raw = (pq4[nh * hd:nh * hd + 8].sum(1) * rstd)
print("k raw expected", kexp_raw.tolist())
print("unfused k", kc1[0, 70, :8].tolist()); print("fused k", kc2[0, 70, :8].tolist())
print("fused v head1", vc2[1, :8, 70].tolist(), "unfused", vc1[1, :8, 70].tolist())
allv = pq4[:N].sum(1) * rstd
for t in range(4):
    target = vc2[0, t, 70].item()
    idx = (allv - target).abs().argmin().item()
    print("fused v[0,%d]=%.4f closest element %d (%.4f); expected element %d" % (t, target, idx, allv[idx].item(), (nh + nkv) * hd + t))
# maybe a different rstd: ratio using exact element
e0 = (nh + nkv) * hd
print("ratios", [(vc2[0, t, 70].item() / allv[e0 + t].item()) for t in range(6)])
s0 = pq4[e0:e0 + 6, 0]; s1 = pq4[e0:e0 + 6, 1]
print("s0*rstd", (s0 * rstd).tolist()); print("s1*rstd", (s1 * rstd).tolist())
import os
if os.environ.get("PARO_ATTN_DBG") == "77":
    e0 = (nh + nkv) * hd
    w = ws2.view(torch.uint8)[2048:].view(torch.float32) if ws2.dtype != torch.float32 else ws2.view(-1)[512:]
    print("ws dtype", ws2.dtype, ws2.shape)
    print("dbg pv[0..3] rows:", w[:16].view(4, 4).tolist())
    print("expected slots   :", pq4[e0:e0 + 4].tolist())
    print("dbg pn", w[4096:4100].tolist(), "expected", pq4[N].tolist(), "tag/eo", w[4100:4103].view(torch.int32).tolist(), "e0*32", e0 * 32, "N*32", N * 32)

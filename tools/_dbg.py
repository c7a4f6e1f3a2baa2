import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import paro_oracle as po
from paroquant_amd import ops, _native as nat
from paroquant_amd.linear import PackedParoWeights
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K, sizes, rows = 2560, [4096, 1024, 1024], 3
L = po.make_layer(K + rows, K, sizes)
rng = np.random.default_rng(K)
w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
x = (rng.standard_normal((rows, K)) * 3.0).astype(np.float16)
pk = PackedParoWeights(t(L["qweight"]), t(L["qzeros"]), t(L["scales"]), t(L["theta"]), t(L["pairs"]), t(L["channel_scales"]), sizes).fold_norm_weight(t(w))
xn = po.rmsnorm(x, w, 1e-6)
ideal = po.paro_linear_merged(xn, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes, None, ideal=True)
wide = torch.zeros(rows, K + 64, device=dev, dtype=torch.float16); wide[:, :K] = t(x)
for trial in range(3):
    y2 = ops.w4a16_gemv_fused(wide[:, :K], pk, nat.PROLOGUE_RMSNORM, 1e-6)
    y3 = ops.w4a16_gemv_fused(t(x), pk, nat.PROLOGUE_RMSNORM, 1e-6)
    torch.cuda.synchronize()
    g2, g3 = y2.float().cpu().numpy().astype(np.float64), y3.float().cpu().numpy().astype(np.float64)
    print("trial", trial, "strided", [float(po.rel_err(g2[r:r+1], ideal[r:r+1])) for r in range(rows)], "dense", [float(po.rel_err(g3[r:r+1], ideal[r:r+1])) for r in range(rows)])
for r in (1, 2, 3, 4):
    xx = t(x[:r] if r <= 3 else np.concatenate([x, x[:1]]))
    idl = ideal[:r] if r <= 3 else np.concatenate([ideal, ideal[:1]])
    y = ops.w4a16_gemv_fused(xx, pk, nat.PROLOGUE_RMSNORM, 1e-6)
    print("rows", r, float(po.rel_err(y.float().cpu().numpy().astype(np.float64), idl)))

#!/usr/bin/env python3
"""Determinism stress of the fused decode GEMV (RMSNorm prologue, residual, strided x; 1 and 3 rows): the same launches N times, outputs
compared BIT FOR BIT with the first iteration.

    python tools/stress_fused.py [iterations = 100]          (PARO_LIB_DIR=_lib_x: an experiment build)

Exists because a compiler-level change (the krot = 8 stage loop without its per-stage compare, round 3) produced a 16-wave / 2..4-row
build whose results differed from run to run -- 136 mismatching iterations of 150 -- while every parity test passed most of the time
(profiles/NOTES.md); tests/test_gpu_parity.py::test_fused_gemv_repeated_calls_are_deterministic runs a short version."""
import numpy as np, torch, sys, os
sys.path.insert(0, ".")
from oracle import paro_oracle as po
from paroquant_amd import ops, _native as nat
from paroquant_amd.linear import PackedParoWeights
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
cases = [(2560, [4096, 1024, 1024], 3), (2560, [9728, 9728], 3), (1024, [3072, 3072], 3), (2560, [4096, 1024, 1024], 1),
         (2560, [4096, 1024, 1024], 2), (2560, [4096, 1024, 1024], 4), (4096, [4096, 1024, 1024], 3), (4096, [14336, 14336], 2)]
refs = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    for (K, sizes, rows) in cases:
        key = (K, tuple(sizes), rows)
        if key not in refs:
            L = po.make_layer(K + rows, K, sizes)
            rng = np.random.default_rng(K)
            w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
            x = (rng.standard_normal((rows, K)) * 3.0).astype(np.float16)
            res = rng.standard_normal((rows, sum(sizes))).astype(np.float16)
            pk = PackedParoWeights(t(L["qweight"]), t(L["qzeros"]), t(L["scales"]), t(L["theta"]), t(L["pairs"]), t(L["channel_scales"]), sizes).fold_norm_weight(t(w))
            wide = torch.zeros(rows, K + 64, device=dev, dtype=torch.float16)
            wide[:, :K] = t(x)
            refs[key] = (pk, t(x), t(res), None, None, wide)        # inputs stay on the device: 10 000 iterations take seconds
        pk, x, res, y_first, y2_first, wide = refs[key]
        y = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=res)
        y2 = ops.w4a16_gemv_fused(wide[:, :K], pk, nat.PROLOGUE_RMSNORM, 1e-6)
        if y_first is None:
            refs[key] = (pk, x, res, y.clone(), y2.clone(), wide)
        else:
            if not torch.equal(y, y_first) or not torch.equal(y2, y2_first):
                bad += 1
                d1 = (y.float() - y_first.float()).abs().max().item(); d2 = (y2.float() - y2_first.float()).abs().max().item()
                rowsbad = [(r, (y2[r].float() - y2_first[r].float()).abs().max().item()) for r in range(rows)]
                print("MISMATCH iter", it, key, "dy", d1, "dy2", d2, rowsbad, flush=True)
                if bad <= 6:     # the pattern of the first few: which columns (tile = col // 16), and is a whole row off by one factor (the RMSNorm scalar)?
                    for r in range(rows):
                        df = (y2[r] != y2_first[r]).nonzero().flatten()
                        if df.numel():
                            ratio = (y2[r].float()[df] / y2_first[r].float()[df].clamp_min(1e-6))
                            print("   row", r, "differing columns", int(df.numel()), "of", int(y2.shape[1]), "first", df[:12].tolist(), "tiles", sorted(set((df // 16).tolist()))[:12],
                                  "ratio min/max", float(ratio.min()), float(ratio.max()), flush=True)
# ---- the deferred K-split reduction: producer (parts_out) -> consumer (RMSNorm prologue + parts_in + x_out), qkv as an RMSNorm producer
Lp, Lc = po.make_layer(91, 4096, [2560]), po.make_layer(92, 2560, [4096, 1024, 1024])
mk = lambda L_: PackedParoWeights(t(L_["qweight"]), t(L_["qzeros"]), t(L_["scales"]), t(L_["theta"]), t(L_["pairs"]), t(L_["channel_scales"]), L_["sizes"])
pp, pc = mk(Lp), mk(Lc)
rng = np.random.default_rng(5)
xa, h0 = t(rng.standard_normal((1, 4096)).astype(np.float16)), t(rng.standard_normal((1, 2560)).astype(np.float16))
parts, partsq = torch.zeros(2560, 4, device=dev), torch.zeros(6144 + 1, 4, device=dev)
h1 = torch.zeros(2560, device=dev, dtype=torch.float16)
first = None
bad2 = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    ops.w4a16_gemv_fused(xa, pp, 0, parts_out=parts)
    y = ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_RMSNORM, 1e-6, parts_in=parts, x_out=h1)
    ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_RMSNORM, 1e-6, parts_in=parts, parts_out=partsq)
    torch.cuda.synchronize()
    cur = (parts.clone(), y.clone(), h1.clone(), partsq.clone())
    if first is None:
        first = cur
    elif not all(torch.equal(a, b) for a, b in zip(cur, first)):
        bad2 += 1
        print("MISMATCH (deferred) iter", it, [bool(torch.equal(a, b)) for a, b in zip(cur, first)], flush=True)
print("lib", os.environ.get("PARO_LIB_DIR", "_lib"), "mismatches", bad, "deferred-route mismatches", bad2)

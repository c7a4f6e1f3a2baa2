"""Mixture-of-experts prefill as one launch sequence: shared rotation applied once, device-side
sort by expert, grouped W4A16 GEMM over the expert segments -- against oracle rows; and the product packer's `quantize_moe`
against the reference-generated golden G9."""
import os

import numpy as np
import pytest
import torch

from oracle import paro_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    import paroquant_amd  # noqa: F401
    from paroquant_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _build(E, H, I, seed, dev):
    from paroquant_amd.moe import ParoMoEExperts
    experts, rot = po.make_moe(seed, E, H, I)
    tensors = {}
    for proj, d in experts.items():
        for name, stack in d.items():
            for e in range(E):
                tensors[f"{e}.{proj}.{name}"] = torch.from_numpy(stack[e])
    for name, v in rot.items():
        tensors[name] = torch.from_numpy(v)
    return ParoMoEExperts(tensors, E, dev), experts, rot


def _oracle_pairs(x, idx, experts, rot, pairs):
    """float64 forward of the routed experts for the sampled (token, slot) pairs only (oracle.moe_experts_forward per pair)."""
    xr = po.rotate(x.astype(np.float64), rot["gate_up_weight_pairs"], rot["gate_up_weight_theta"].astype(np.float64),
                   rot["gate_up_weight_channel_scales"].astype(np.float64).reshape(-1), 128, "ideal")
    deq = lambda proj, e: po.dequant_awq(experts[proj]["qweight"][e], experts[proj]["qzeros"][e], experts[proj]["scales"][e], 128, np.float16).astype(np.float64)
    out = []
    for t, s in pairs:
        e = int(idx[t, s])
        g, u = xr[t] @ deq("gate_proj", e), xr[t] @ deq("up_proj", e)
        act = (g / (1.0 + np.exp(-g)) * u)[None, :]
        ar = po.rotate(act, rot["down_weight_pairs"], rot["down_weight_theta"].astype(np.float64),
                       rot["down_weight_channel_scales"].astype(np.float64).reshape(-1), 128, "ideal")
        out.append(ar[0] @ deq("down_proj", e))
    return np.stack(out)


@pytest.mark.parametrize("E,H,I,T,k", [(64, 1024, 512, 512, 8), (64, 1024, 512, 4096, 8), (8, 512, 256, 100, 2), (16, 256, 128, 33, 3)])
def test_moe_grouped_prefill_matches_oracle_rows(dev, E, H, I, T, k):
    moe, experts, rot = _build(E, H, I, E * 1000 + T, dev)
    rng = np.random.default_rng(T + k)
    x = rng.standard_normal((T, H)).astype(np.float16)
    # skewed routing: some experts get many tokens, some none (E = 64: expert 63 never routed to)
    p = rng.dirichlet(np.full(E, 0.6))
    if E >= 64:
        p[-1] = 0.0
    p /= p.sum()
    idx = np.stack([rng.choice(E, size=k, replace=False, p=p) for _ in range(T)]).astype(np.int64)
    xt, it = _t(x, dev), _t(idx, dev)
    y = moe(xt, it)
    assert y.shape == (T, k, H) and torch.isfinite(y.float()).all()
    sample = [(int(rng.integers(T)), int(rng.integers(k))) for _ in range(48)]
    ref = _oracle_pairs(x, idx, experts, rot, sample)
    got = np.stack([_np(y[t, s]) for t, s in sample])
    assert po.rel_err(got, ref) < 4e-3
    # the whole tensor against the per-expert route (the round-2 prefill path: same kernels per expert, host loop)
    y_loop = moe.per_expert_prefill(xt, it)
    assert po.rel_err(_np(y), _np(y_loop)) < 2e-3
    # capturable: nothing of the grouped route is read back to the host; other routings replay through the same graph
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        moe(xt, it)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = moe(xt, it)
    it.copy_(torch.flip(it, dims=[1]))
    g.replay()
    torch.cuda.synchronize()
    assert po.rel_err(_np(yg), _np(torch.flip(y, dims=[1]))) < 1e-6 or torch.equal(yg, torch.flip(y, dims=[1]))
    # expert ids are checked ON THE DEVICE (paro_experts_t.n_experts; no host sync per MoE block, and the same guarantee under graph
    # replay): an id outside [0, E) never reads out of bounds and its slot comes back NaN; the other slots are untouched
    bad = it.clone()
    bad[0, 0] = E
    yb = moe(xt, bad)
    torch.cuda.synchronize()
    assert torch.isnan(yb[0, 0]).all() and torch.equal(yb[0, 1:], moe(xt, it)[0, 1:]) and torch.isfinite(yb[1:]).all()
    bad[0, 0] = -1
    assert torch.isnan(moe(xt, bad)[0, 0]).all()


def test_pack_quantize_moe_matches_golden_g9(dev):
    """paroquant_amd.pack.quantize_moe on the GPU against the fixture the REFERENCE's `_quantize_moe` produced
    (tests/golden/make_golden_g9.py): which gate_up rows are gate / up, the per-expert AWQ stacking, the shared rotation
    buffers and their names; numbered `*_pairs_grouped.N` lists and the `quantizer.n_bits` spelling are accepted."""
    from paroquant_amd import pack
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quantize_moe.npz"))
    sd = {k[3:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    bufs, rot = pack.quantize_moe(sd, dev)
    for proj in ("gate_proj", "up_proj", "down_proj"):
        for key in ("qzeros", "scales"):
            a, b = bufs[proj][key].cpu().numpy(), g[f"out_{proj}_{key}"]
            assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (proj, key)
        # fp32 rotation on the GPU vs the fixture's: a weight on a quantisation boundary may land on the neighbouring level
        qa = np.stack([pack.unpack_awq(bufs[proj]["qweight"][e]).cpu().numpy() for e in range(bufs[proj]["qweight"].shape[0])]).astype(np.int64)
        qb = np.stack([po.unpack_awq(g[f"out_{proj}_qweight"][e]) for e in range(qa.shape[0])]).astype(np.int64)
        assert qa.shape == qb.shape and np.abs(qa - qb).max() <= 1 and ((qa - qb) != 0).mean() < 1e-3, proj
    for key in ("gate_up_weight_theta", "gate_up_weight_pairs", "gate_up_weight_channel_scales", "down_weight_theta", "down_weight_pairs",
                "down_weight_channel_scales"):
        a, b = rot[key].cpu().numpy(), g["rot_" + key]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), key
    # the optimiser's other spellings (cli/convert.py:127-146)
    sd2 = dict(sd)
    sd2["quantizer.n_bits"], sd2["quantizer.group_size"] = sd2.pop("n_bits"), sd2.pop("group_size")
    pg = sd2.pop("gate_up_pairs_grouped")
    for i in range(pg.shape[0]):
        sd2[f"gate_up_pairs_grouped.{i}"] = pg[i]
    bufs2, rot2 = pack.quantize_moe(sd2, dev)
    assert torch.equal(bufs2["down_proj"]["qweight"], bufs["down_proj"]["qweight"]) and torch.equal(rot2["gate_up_weight_pairs"], rot["gate_up_weight_pairs"])
    with pytest.raises(KeyError):
        pack.state_value({}, "n_bits", "quantizer.n_bits")

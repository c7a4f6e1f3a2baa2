"""Split decode attention (paro_attn_decode_split, ABI v14): the merge over position chunks is left to the consumer -- paro_attn_finish,
or the attn_in prologue of the fused GEMV (o_proj).  Against the float64 oracle (oracle/paro_oracle.py: attention_decode, paro_linear_merged)
and, bit for bit, against the route that finishes in its own launch."""
import os

import numpy as np
import pytest
import torch

from oracle import paro_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _case(seed, hd, Hq, Hkv, T, qk_norm, dev, dtype=np.float16):
    rng = np.random.default_rng(seed)
    qkv = rng.standard_normal((Hq + 2 * Hkv) * hd).astype(dtype)
    kc = rng.standard_normal((Hkv, T, hd)).astype(dtype)
    vc = rng.standard_normal((Hkv, T, hd)).astype(dtype)
    qw = (1 + 0.2 * rng.standard_normal(hd)).astype(dtype) if qk_norm else None
    kw = (1 + 0.2 * rng.standard_normal(hd)).astype(dtype) if qk_norm else None
    cos, sin = po.rope_tables(hd, T, 1e4)
    rope = torch.from_numpy(np.concatenate([cos, sin], axis=-1).astype(np.float32)).to(dev)
    return qkv, kc, vc, qw, kw, cos, sin, rope


@pytest.mark.parametrize("hd,Hq,Hkv,qk_norm", [(128, 8, 2, True), (64, 4, 2, False), (128, 32, 8, True), (128, 4, 4, False), (128, 16, 2, True),
                                             (64, 16, 2, True)])      # (the last two: eight query heads per KV head, Llama-3-70B's ratio)
@pytest.mark.parametrize("T,pos", [(264, 0), (264, 5), (264, 127), (264, 128), (264, 255), (264, 263), (1024, 300), (1024, 511), (1024, 512),
                                    (1024, 1023), (4096, 2500), (4096, 4095)])
def test_split_attention_and_finish_match_oracle(dev, hd, Hq, Hkv, qk_norm, T, pos):
    """Every slot regime: one active chunk, one chunk per slot (<= 512 positions), several chunks per slot behind the per-slot ticket."""
    from paroquant_amd import ops
    qkv, kc, vc, qw, kw, cos, sin, rope = _case(hd + Hq + pos + T, hd, Hq, Hkv, T, qk_norm, dev)
    kct, vct = _t(kc, dev), _t(vc.transpose(0, 2, 1), dev)
    k2, v2 = kct.clone(), vct.clone()
    pt = torch.tensor([pos], dtype=torch.int32, device=dev)
    nw = (None if qw is None else _t(qw, dev), None if kw is None else _t(kw, dev))
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
    ws = ops.attn_workspace(dev, Hq, Hkv, hd, T)
    r = ops.attn_decode(_t(qkv, dev), kct, vct, pt, rope, Hq, Hkv, hd, *nw, 1e-6, workspace=ws, split_out=sp)
    assert r is sp
    out = ops.attn_finish(sp, Hq, hd)
    ref, k_new, v_new = po.attention_decode(qkv, kc, vc, pos, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
    assert po.rel_err(out.float().cpu().numpy(), ref) < 4e-3          # the in-launch merge's bound (test_attn_decode_matches_oracle)
    assert po.rel_err(kct[:, pos].float().cpu().numpy(), k_new) < 2e-3 and po.rel_err(vct[:, :, pos].float().cpu().numpy(), v_new) < 1e-6
    # the in-launch merge of the same kernel family agrees to rounding, and the caches end up identical
    o2 = ops.attn_decode(_t(qkv, dev), k2, v2, pt, rope, Hq, Hkv, hd, *nw, 1e-6, workspace=ws)
    assert po.rel_err(out.float().cpu().numpy(), o2.float().cpu().numpy().astype(np.float64)) < 2e-3
    assert torch.equal(kct, k2) and torch.equal(vct, v2)
    # slots nobody filled carry the sentinel
    ml = sp[Hq * hd * 4:].view(Hq, 8)
    env = os.environ.get("PARO_ATTN_SPLIT_CHUNK")                                   # (A/B knob of attn.hip; default: by position)
    chunk = int(env) if env in ("64", "128") else (64 if pos < 256 else 128)      # positions per workgroup of the split launch
    n_act = pos // chunk + 1
    per = (n_act + 3) // 4
    used = (n_act + per - 1) // per
    assert bool((ml[:, 4:4 + used] > 0).all()) and bool((ml[:, 4 + used:] == 0).all()) and bool((ml[:, used:4] < -1e37).all())
    # the tickets are back at zero: the same workspace serves the next launch
    assert int(ws.view(torch.int32)[:512].abs().sum()) == 0


def test_split_attention_reuses_stale_slots(dev):
    """A long sequence, then a short one on the SAME slot buffer: slots 1..3 hold stale (even non-finite) values and must not leak."""
    from paroquant_amd import ops
    hd, Hq, Hkv, T = 128, 8, 2, 1024
    qkv, kc, vc, qw, kw, cos, sin, rope = _case(5, hd, Hq, Hkv, T, True, dev)
    kct, vct = _t(kc, dev), _t(vc.transpose(0, 2, 1), dev)
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
    nw = (_t(qw, dev), _t(kw, dev))
    ops.attn_decode(_t(qkv, dev), kct.clone(), vct.clone(), torch.tensor([900], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd, *nw, 1e-6, split_out=sp)
    sp[:Hq * hd * 4].view(-1, 4)[:, 1:] = float("nan")        # whatever an earlier launch left in the outputs of slots 1..3
    ops.attn_decode(_t(qkv, dev), kct, vct, torch.tensor([17], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd, *nw, 1e-6, split_out=sp)
    out = ops.attn_finish(sp, Hq, hd)
    ref, _, _ = po.attention_decode(qkv, kc, vc, 17, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
    assert bool(torch.isfinite(out).all()) and po.rel_err(out.float().cpu().numpy(), ref) < 4e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hd,Hq,Hkv,N,pos", [(128, 32, 8, 2560, 200), (128, 32, 8, 4096, 700), (64, 16, 8, 1024, 130), (128, 8, 2, 512, 3)])
def test_attn_in_prologue_equals_finish_then_linear(dev, dtype, hd, Hq, Hkv, N, pos):
    """o_proj on the un-merged slots (paro_fusion_t.attn_in) == paro_attn_finish -> the plain launch, bit for bit -- as y and as the K-split
    partial sums a deferred reduction leaves; and against the float64 oracle end to end (attention -> linear)."""
    from paroquant_amd import ops, _native as nat
    from paroquant_amd.linear import PackedParoWeights
    T = 1024
    K = Hq * hd
    npdt = np.float16
    qkv, kc, vc, qw, kw, cos, sin, rope = _case(N + pos, hd, Hq, Hkv, T, True, dev, npdt)
    L = po.make_layer(K + N, K, [N])
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev), _t(L["pairs"], dev),
                           _t(L["channel_scales"], dev), L["sizes"], None)
    to = lambda a: _t(a, dev).to(dtype)
    kct, vct = to(kc), to(vc.transpose(0, 2, 1))
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
    ops.attn_decode(to(qkv), kct, vct, torch.tensor([pos], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd, to(qw), to(kw), 1e-6, split_out=sp)
    x = ops.attn_finish(sp, Hq, hd, dtype=dtype)
    y_ref = ops.w4a16_gemv_fused(x.view(1, K), pk, 0)
    y = ops.w4a16_gemv_fused(None, pk, 0, attn_in=sp, attn_head_dim=hd, dtype=dtype)
    assert y.dtype == dtype and torch.equal(y, y_ref)
    n = ops.gemv_parts_count(pk, dtype)
    if n >= 2:
        p_ref = torch.zeros(N, 4, dtype=torch.float32, device=dev)
        p = torch.zeros_like(p_ref)
        ops.w4a16_gemv_fused(x.view(1, K), pk, 0, parts_out=p_ref)
        ops.w4a16_gemv_fused(None, pk, 0, parts_out=p, attn_in=sp, attn_head_dim=hd, dtype=dtype)
        assert torch.equal(p, p_ref)
    if dtype == torch.float16:
        att, _, _ = po.attention_decode(qkv, kc, vc, pos, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
        ideal = po.paro_linear_merged(att.reshape(1, K), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], [N],
                                      None, ideal=True)
        assert po.rel_err(y.float().cpu().numpy(), ideal) < 1e-2      # north-star tolerance
    torch.cuda.synchronize()
    ops.check_workspace(pk.workspace)


def test_attn_in_argument_errors(dev):
    from paroquant_amd import ops, _native as nat
    from paroquant_amd.linear import PackedParoWeights
    hd, Hq, N = 128, 4, 256
    K = Hq * hd
    L = po.make_layer(3, K, [N])
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev), _t(L["pairs"], dev),
                           _t(L["channel_scales"], dev), L["sizes"], None)
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        ops.w4a16_gemv_fused(None, pk, 0, attn_in=sp[:-1].contiguous(), attn_head_dim=hd)
    with pytest.raises(ValueError):
        ops.w4a16_gemv_fused(None, pk, 0, attn_in=sp, attn_head_dim=96)
    with pytest.raises(RuntimeError, match="no prologue"):
        ops.w4a16_gemv_fused(None, pk, nat.PROLOGUE_RMSNORM, attn_in=sp, attn_head_dim=hd)
    with pytest.raises(RuntimeError, match="no prologue"):
        ops.w4a16_gemv_fused(None, pk, 0, attn_in=sp, attn_head_dim=hd, residual=torch.zeros(1, N, dtype=torch.float16, device=dev))


def test_split_attention_graph_replay(dev):
    """The split launch + the attn_in consumer in one captured graph, replayed over growing positions: equal to the eager pair."""
    from paroquant_amd import ops
    from paroquant_amd.linear import PackedParoWeights
    hd, Hq, Hkv, T, N = 128, 8, 2, 264, 1024
    K = Hq * hd
    qkv, kc, vc, qw, kw, cos, sin, rope = _case(11, hd, Hq, Hkv, T, True, dev)
    L = po.make_layer(K + N, K, [N])
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev), _t(L["pairs"], dev),
                           _t(L["channel_scales"], dev), L["sizes"], None)
    kct, vct = _t(kc, dev), _t(vc.transpose(0, 2, 1), dev)
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
    ws = ops.attn_workspace(dev, Hq, Hkv, hd, T)
    pt = torch.zeros(1, dtype=torch.int32, device=dev)
    q, nw = _t(qkv, dev), (_t(qw, dev), _t(kw, dev))
    y = torch.empty(1, N, dtype=torch.float16, device=dev)

    def step():
        ops.attn_decode(q, kct, vct, pt, rope, Hq, Hkv, hd, *nw, 1e-6, workspace=ws, split_out=sp)
        ops.w4a16_gemv_fused(None, pk, 0, attn_in=sp, attn_head_dim=hd, out=y)
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for pos in (0, 100, 127, 128, 200, 255, 256, 263, 40):
        pt.fill_(pos)
        g.replay()
        got = y.clone()
        step()
        assert torch.equal(got, y), pos
        ref, k_new, v_new = po.attention_decode(qkv, kc, vc, pos, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
        assert po.rel_err(ops.attn_finish(sp, Hq, hd).float().cpu().numpy(), ref) < 4e-3, pos
        kc[:, pos], vc[:, pos] = k_new.astype(np.float16), v_new.astype(np.float16)     # the launch appended this position to the caches


@pytest.mark.parametrize("split", [True, False])
def test_attention_ticket_merge_is_deterministic(dev, split):
    """The in-launch merge of several chunks (write-through stores -> ticket -> agent-scope loads, no cache-wide fence): 3000 graph
    replays at a long position, every result compared on the device with the first one -- a lost or stale partial result would show
    as a mismatch (or a NaN)."""
    from paroquant_amd import ops
    hd, Hq, Hkv, T, pos = 128, 32, 8, 2048, 2047
    qkv, kc, vc, qw, kw, cos, sin, rope = _case(77, hd, Hq, Hkv, T, True, dev)
    kct, vct = _t(kc, dev), _t(vc.transpose(0, 2, 1), dev)
    q, nw = _t(qkv, dev), (_t(qw, dev), _t(kw, dev))
    pt = torch.tensor([pos], dtype=torch.int32, device=dev)
    ws = ops.attn_workspace(dev, Hq, Hkv, hd, T)
    sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev) if split else None
    out = torch.empty(Hq * hd, dtype=torch.float16, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        ops.attn_decode(q, kct, vct, pt, rope, Hq, Hkv, hd, *nw, 1e-6, out=out, workspace=ws, split_out=sp)
        if split:
            ops.attn_finish(sp, Hq, hd, out=out)
    step()
    first = out.clone()
    ref, _, _ = po.attention_decode(qkv, kc, vc, pos, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
    assert po.rel_err(first.float().cpu().numpy(), ref) < 4e-3
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            step()
            bad.add_((out != first).sum())
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    assert int(ws.view(torch.int32)[:512].abs().sum()) == 0


def test_split_attention_bf16(dev):
    """bf16 caches and activations through the split launch and its completion (probabilities rounded to bf16 for P V as HF does)."""
    from paroquant_amd import ops
    bf = torch.bfloat16
    hd, Hq, Hkv, T = 128, 16, 4, 1024
    rng = np.random.default_rng(9)
    f = lambda t: t.float().cpu().numpy()
    for pos in (37, 200, 255, 256, 800):
        qkv = torch.from_numpy(rng.standard_normal((Hq + 2 * Hkv) * hd).astype(np.float32)).to(dev).to(bf)
        kc = torch.from_numpy(rng.standard_normal((Hkv, T, hd)).astype(np.float32)).to(dev).to(bf)
        vc = torch.from_numpy(rng.standard_normal((Hkv, T, hd)).astype(np.float32)).to(dev).to(bf)
        qw = torch.from_numpy((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32)).to(dev).to(bf)
        kw = torch.from_numpy((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32)).to(dev).to(bf)
        cos, sin = po.rope_tables(hd, T, 1e4)
        rope = torch.from_numpy(np.concatenate([cos, sin], axis=-1).astype(np.float32)).to(dev)
        kct, vct = kc.clone(), vc.transpose(1, 2).contiguous()
        sp = torch.zeros(ops.attn_parts_floats(Hq, hd), dtype=torch.float32, device=dev)
        ops.attn_decode(qkv, kct, vct, torch.tensor([pos], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd, qw, kw, 1e-6, split_out=sp)
        out = ops.attn_finish(sp, Hq, hd, dtype=bf)
        ref, k_new, v_new = po.attention_decode(f(qkv), f(kc), f(vc), pos, Hq, Hkv, hd, cos, sin, f(qw), f(kw), 1e-6)
        assert out.dtype == bf and po.rel_err(f(out), ref) < 3e-2, pos          # the bound of test_fused_prologues_and_attention_bf16
        assert po.rel_err(f(kct[:, pos]), k_new) < 2e-2 and po.rel_err(f(vct[:, :, pos]), v_new) < 1e-6

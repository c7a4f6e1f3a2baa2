"""GPU tests of the deferred K-split reduction (include/paro_abi.h v12, ``paro_fusion_t.parts_out / parts_in / x_out``):
the K-split producer (o_proj, down_proj) leaves fp32 partial sums, the RMSNorm-prologue consumer behind it completes
``x' = round(x + sum(parts))`` while it seeds its rotation.  Checked (1) against the CPU oracle on the same seeded inputs
(tolerance of tests/test_gpu_parity.py), (2) BIT FOR BIT against the ordinary route (in-launch reducer + residual epilogue, then the
RMSNorm-prologue launch): same summation order, same single rounding."""
import numpy as np
import pytest
import torch

from oracle import paro_oracle as po
from tests.test_gpu_parity import TIGHT_BF16, TIGHT_F16, _np, _packed, _t, dev  # noqa: F401

pytestmark = pytest.mark.gpu

# (producer K, producer N = consumer K, consumer partitions): Qwen3-4B o -> gate_up, down -> qkv; Llama-3-8B down -> qkv; a small case
PAIRS = [(4096, 2560, [9728, 9728]), (9728, 2560, [4096, 1024, 1024]), (14336, 4096, [4096, 1024, 1024]), (1024, 512, [208, 48])]


def _layers(Kp, H, sizes, seed):
    return po.make_layer(seed, Kp, [H]), po.make_layer(seed + 1, H, sizes)


@pytest.mark.parametrize("Kp,H,sizes", PAIRS)
@pytest.mark.parametrize("dtype,tol", [(torch.float16, TIGHT_F16), (torch.bfloat16, TIGHT_BF16)])
@pytest.mark.parametrize("n_parts", [0, 2, 3])
def test_parts_route_matches_oracle_and_fused_route(dev, Kp, H, sizes, dtype, tol, n_parts):
    from paroquant_amd import ops, _native as nat
    Lp, Lc = _layers(Kp, H, sizes, Kp + H)
    rng = np.random.default_rng(Kp * 3 + H)
    xa = _t(rng.standard_normal((1, Kp)).astype(np.float32), dev, dtype)              # the producer's input (attention output / act)
    h0 = _t((rng.standard_normal((1, H)) * 2.0).astype(np.float32), dev, dtype)       # the residual stream in front of the producer
    w = (1.0 + 0.2 * rng.standard_normal(H)).astype(np.float16)
    pp = _packed(Lp, dev)
    pc = _packed(Lc, dev).fold_norm_weight(_t(w, dev))
    n = n_parts if n_parts else ops.gemv_parts_count(pp, dtype)
    if n == 0:
        assert Kp < 4096                       # every decoder-sized o / down splits; the small case does not on its own
        n = 2
    # ---- the ordinary route: producer reduces in its launch and adds the residual; consumer with the RMSNorm prologue
    h1_ref = ops.w4a16_gemv_fused(xa, pp, nat.PROLOGUE_NONE, residual=h0)
    y_ref = ops.w4a16_gemv_fused(h1_ref, pc, nat.PROLOGUE_RMSNORM, 1e-6)
    # ---- the parts route
    parts = torch.full((H, 4), float("nan"), device=dev, dtype=torch.float32)
    assert ops.w4a16_gemv_fused(xa, pp, nat.PROLOGUE_NONE, parts_out=parts, parts_n=n) is parts
    h1 = torch.zeros(H, device=dev, dtype=dtype)
    y = ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_RMSNORM, 1e-6, parts_in=parts, x_out=h1)
    torch.cuda.synchronize()
    assert torch.isfinite(parts).all() and (parts[:, n:] == 0).all()          # unused slots are zeroed by the producer
    # oracle: the producer's linear, the residual add, the norm, the consumer's linear
    yo = po.paro_linear_merged(_np(xa), Lp["qweight"], Lp["qzeros"], Lp["scales"], Lp["theta"], Lp["pairs"], Lp["channel_scales"], [H],
                               None, ideal=True)
    assert po.rel_err(parts.double().sum(1)[None, :].cpu().numpy(), yo) < tol
    assert po.rel_err(_np(h1)[None, :], yo + _np(h0)) < tol
    xn = po.rmsnorm(_np(h1)[None, :].astype(np.float32), w, 1e-6)
    ideal = po.paro_linear_merged(xn, Lc["qweight"], Lc["qzeros"], Lc["scales"], Lc["theta"], Lc["pairs"], Lc["channel_scales"], sizes,
                                  None, ideal=True)
    assert po.rel_err(_np(y), ideal) < 2 * tol
    # bit identity with the ordinary route whenever that route split K the same way (its automatic shape)
    if n_parts == 0 and ops.gemv_parts_count(pp, dtype) == n:
        assert torch.equal(h1.view(-1), h1_ref.view(-1))
        assert torch.equal(y, y_ref)
    else:
        assert po.rel_err(_np(h1)[None, :], _np(h1_ref)) < (2e-3 if dtype == torch.float16 else 2e-2)
    # the stand-alone completion (no linear behind the producer) gives the same residual stream
    assert torch.equal(ops.parts_finish(parts, h0.view(-1)), h1)
    # the plain (no norm) consumer, without x_out
    # (the ordinary launch may K-split -- Llama-3-8B's qkv does --, the consumer of partial sums never does: same numbers, not the same bits)
    y_plain = ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_NONE, parts_in=parts)
    assert po.rel_err(_np(y_plain), _np(ops.w4a16_gemv_fused(h1.view(1, -1), pc, nat.PROLOGUE_NONE))) < (1e-3 if dtype == torch.float16 else 1e-2)
    ops.check_workspace(pp.workspace)


def test_parts_with_silu_producer_and_graph_replay(dev):
    """down_proj (SiLU * mul prologue) as the producer, the next layer's qkv as the consumer, captured in a HIP graph and replayed
    on fresh inputs: nothing is armed or reset between launches."""
    from paroquant_amd import ops, _native as nat
    Kp, H, sizes = 9728, 2560, [4096, 1024, 1024]
    Lp, Lc = _layers(Kp, H, sizes, 77)
    pp, pc = _packed(Lp, dev), _packed(Lc, dev)
    n = ops.gemv_parts_count(pp)
    assert 2 <= n <= nat.PARO_MAX_PARTIALS
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    gu = torch.randn(1, 2 * Kp, device=dev, dtype=torch.float16, generator=gen)
    h0 = torch.randn(1, H, device=dev, dtype=torch.float16, generator=gen)
    parts = torch.zeros(H, 4, device=dev, dtype=torch.float32)
    h1 = torch.zeros(H, device=dev, dtype=torch.float16)
    y = torch.zeros(1, sum(sizes), device=dev, dtype=torch.float16)

    def step():
        ops.w4a16_gemv_fused(gu, pp, nat.PROLOGUE_SILU_MUL, parts_out=parts)
        ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_RMSNORM, 1e-6, parts_in=parts, x_out=h1, out=y)

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for it in range(3):
        gu.copy_(torch.randn(1, 2 * Kp, device=dev, dtype=torch.float16, generator=gen))
        h0.copy_(torch.randn(1, H, device=dev, dtype=torch.float16, generator=gen))
        g.replay()
        torch.cuda.synchronize()
        h1_ref = ops.w4a16_gemv_fused(gu, pp, nat.PROLOGUE_SILU_MUL, residual=h0)
        y_ref = ops.w4a16_gemv_fused(h1_ref, pc, nat.PROLOGUE_RMSNORM, 1e-6)
        assert torch.equal(h1, h1_ref.view(-1)) and torch.equal(y, y_ref), it


def test_parts_argument_errors(dev):
    from paroquant_amd import ops, _native as nat
    Lp, Lc = _layers(1024, 512, [208, 48], 9)
    pp, pc = _packed(Lp, dev), _packed(Lc, dev)
    x = torch.zeros(1, 1024, device=dev, dtype=torch.float16)
    h = torch.zeros(1, 512, device=dev, dtype=torch.float16)
    parts = torch.zeros(512, 4, device=dev, dtype=torch.float32)
    with pytest.raises(RuntimeError, match="batch-1"):
        ops.w4a16_gemv_fused(torch.zeros(2, 1024, device=dev, dtype=torch.float16), pp, parts_out=parts, parts_n=2)
    with pytest.raises(RuntimeError, match="residual and bias"):
        ops.w4a16_gemv_fused(x, pp, residual=h, parts_out=parts, parts_n=2)
    with pytest.raises(RuntimeError, match="residual and bias"):
        ops.w4a16_gemv_fused(x, pp, bias=torch.zeros(512, device=dev, dtype=torch.float16), parts_out=parts, parts_n=2)
    with pytest.raises(ValueError, match="float32"):
        ops.w4a16_gemv_fused(x, pp, parts_out=parts.half())
    with pytest.raises(ValueError, match="float32"):
        ops.w4a16_gemv_fused(x, pp, parts_out=torch.zeros(512, 5, device=dev))
    with pytest.raises(RuntimeError, match="does not K-split on its own"):
        ops.w4a16_gemv_fused(x, pp, parts_out=parts)
    with pytest.raises(RuntimeError, match="parts_out_n must be in 2"):
        ops.w4a16_gemv_fused(x, pp, parts_out=parts, parts_n=5)
    with pytest.raises(RuntimeError, match="non-empty slices|parts_out_n"):
        ops.w4a16_gemv_fused(torch.zeros(1, 128, device=dev, dtype=torch.float16), _packed(po.make_layer(1, 128, [512]), dev), parts_out=parts, parts_n=2)
    with pytest.raises(RuntimeError, match="plain or the RMSNorm"):
        ops.w4a16_gemv_fused(torch.zeros(1, 1024, device=dev, dtype=torch.float16), _packed(po.make_layer(2, 512, [64]), dev),
                             nat.PROLOGUE_SILU_MUL, parts_in=parts)
    with pytest.raises(RuntimeError, match="alias"):
        ops.w4a16_gemv_fused(h, pc, nat.PROLOGUE_RMSNORM, parts_in=parts, x_out=h.view(-1))
    with pytest.raises(ValueError, match="513"):            # an RMSNorm-prologue producer also leaves its sums of squares: one more row
        ops.w4a16_gemv_fused(h, _packed(po.make_layer(3, 512, [512]), dev), nat.PROLOGUE_RMSNORM, parts_in=parts, parts_out=parts.clone(), parts_n=2)
    with pytest.raises(ValueError, match="x_out"):
        ops.w4a16_gemv_fused(h, pc, x_out=h.view(-1).clone())
    assert ops.gemv_parts_count(_packed(po.make_layer(4, 256, [4096]), dev)) == 0      # wide output, shallow K: no split


@pytest.mark.parametrize("dtype,tol", [(torch.float16, TIGHT_F16), (torch.bfloat16, TIGHT_BF16)])
def test_parts_chain_consumer_is_producer(dev, dtype, tol):
    """A chain of plain linears (bench.py's stack): down -> qkv -> o, every K-split deferred.  qkv both RECEIVES down's partial sums and
    LEAVES its own (2-way split of the launch shape nobody polls in), o receives them and reduces its 4-way split in the launch.
    Against the oracle on the chain, and bit for bit against the ordinary launches wherever the split is the same."""
    from paroquant_amd import ops, _native as nat
    Kd, H, qkv_sizes, Ko = 9728, 2560, [4096, 1024, 1024], 4096
    Ld, Lq, Lo = po.make_layer(31, Kd, [H]), po.make_layer(32, H, qkv_sizes), po.make_layer(33, Ko, [H])
    pd, pq, pko = _packed(Ld, dev), _packed(Lq, dev), _packed(Lo, dev)
    assert ops.gemv_parts_count(pd, dtype) == 4 and ops.gemv_parts_count(pq, dtype) == 2 and ops.gemv_parts_count(pko, dtype) == 4
    rng = np.random.default_rng(8)
    xd = _t(rng.standard_normal((1, Kd)).astype(np.float32), dev, dtype)
    zeros = torch.zeros(1, 4096, device=dev, dtype=dtype)
    # ordinary launches
    y_d = pd.apply(xd)
    y_q = pq.apply(y_d)
    y_o = pko.apply(y_q[:, :Ko].contiguous())
    # deferred chain
    p_d = torch.full((H, 4), float("nan"), device=dev, dtype=torch.float32)
    p_q = torch.full((sum(qkv_sizes), 4), float("nan"), device=dev, dtype=torch.float32)
    ops.w4a16_gemv_fused(xd, pd, 0, parts_out=p_d)
    ops.w4a16_gemv_fused(zeros[:, :H], pq, 0, parts_in=p_d, parts_out=p_q)               # consumer AND producer
    y_o2 = ops.w4a16_gemv_fused(zeros[:, :Ko], pko, 0, parts_in=p_q[:Ko])                 # consumer with its own (polled) 4-way split
    torch.cuda.synchronize()
    assert torch.equal(ops.parts_finish(p_d, dtype=dtype), y_d.view(-1))                  # same 4-way split: the reducer's bits
    yq2 = ops.parts_finish(p_q, dtype=dtype)
    assert (p_q[:, 2:] == 0).all()
    assert po.rel_err(_np(yq2)[None, :], _np(y_q)) < (1e-3 if dtype == torch.float16 else 1e-2)     # 2-way split vs unsplit: fp32 order only
    # o on the completed qkv output: the same launch shape in both routes -> the same bits
    assert torch.equal(y_o2, pko.apply(yq2[None, :Ko].contiguous()))
    ideal = po.paro_linear_merged(_np(yq2)[None, :Ko], Lo["qweight"], Lo["qzeros"], Lo["scales"], Lo["theta"], Lo["pairs"], Lo["channel_scales"],
                                  [H], None, ideal=True)
    assert po.rel_err(_np(y_o2), ideal) < tol
    assert po.rel_err(_np(y_o2), _np(y_o)) < 20 * tol          # two chained linears amplify the rounding differences of their inputs
    ops.check_workspace(pko.workspace)


def _parts_fuzz_cases(n=48):
    rng = np.random.default_rng(1203)
    out = []
    for i in range(n):
        Gp = int(rng.integers(2, 40))                       # producer groups (K = 128 Gp)
        n_parts = int(rng.integers(2, min(4, Gp) + 1))
        gps = -(-Gp // n_parts)
        n_parts = -(-Gp // gps)                             # empty slices are dropped by the launch: ask for what K / 128 divides into
        H = int(rng.integers(1, 17)) * 128                  # producer N = consumer K
        P = int(rng.integers(1, 4))
        sizes = [int(rng.integers(1, 60)) * 16 for _ in range(P)]
        if i % 6 == 0:
            sizes[0] = int(rng.integers(64, 200)) * 16      # a wide consumer now and then
        out.append((i, Gp * 128, n_parts, H, tuple(sizes), i % 3, 64 if i % 5 == 1 else 128, bool(i % 2), int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("case,Kp,n_parts,H,sizes,consumer,gs,use_bf16,producer_prologue", _parts_fuzz_cases())
def test_random_shapes_deferred_reduction(dev, case, Kp, n_parts, H, sizes, consumer, gs, use_bf16, producer_prologue):
    """Seeded random (producer, consumer) pairs: any split count K / 128 allows, ragged consumers, both group sizes, both dtypes;
    the consumer with the RMSNorm prologue (+ x_out), plain with x_out (unsplit), or plain without (it may split K itself).
    The producer plain or behind the SiLU * mul prologue.  Against the oracle, stage by stage on the GPU's own intermediate."""
    from paroquant_amd import ops, _native as nat
    sizes = list(sizes)
    dt = torch.bfloat16 if use_bf16 else torch.float16
    tol = 2e-2 if use_bf16 else TIGHT_F16
    Lp, Lc = po.make_layer(5000 + case, Kp, [H], group_size=gs), po.make_layer(6000 + case, H, sizes, group_size=gs)
    pp, pc = _packed(Lp, dev), _packed(Lc, dev)
    rng = np.random.default_rng(case)
    h0 = _t((rng.standard_normal((1, H)) * 1.5).astype(np.float32), dev).to(dt)
    parts = torch.full((H, 4), float("nan"), device=dev, dtype=torch.float32)
    if producer_prologue == 2:
        xin = _t(rng.standard_normal((1, 2 * Kp)).astype(np.float32), dev).to(dt)
        xeff = po.silu_mul(xin.float().cpu().numpy(), Kp)
        ops.w4a16_gemv_fused(xin, pp, nat.PROLOGUE_SILU_MUL, parts_out=parts, parts_n=n_parts)
    else:
        xin = _t(rng.standard_normal((1, Kp)).astype(np.float32), dev).to(dt)
        xeff = xin.float().cpu().numpy()
        ops.w4a16_gemv_fused(xin, pp, 0, parts_out=parts, parts_n=n_parts)
    torch.cuda.synchronize()
    assert torch.isfinite(parts).all() and (parts[:, n_parts:] == 0).all()
    y_p = po.paro_linear_merged(xeff, Lp["qweight"], Lp["qzeros"], Lp["scales"], Lp["theta"], Lp["pairs"], Lp["channel_scales"], [H], None,
                                group_size=gs, ideal=True)
    assert po.rel_err(parts.double().sum(1)[None, :].cpu().numpy(), y_p) < tol
    h1_ref = ops.parts_finish(parts, h0.view(-1))
    assert po.rel_err(_np(h1_ref)[None, :], y_p + _np(h0)) < tol
    h1 = torch.zeros(H, device=dev, dtype=dt)
    if consumer == 0:      # RMSNorm prologue, the new residual stream written
        w = (1.0 + 0.2 * rng.standard_normal(H)).astype(np.float16)
        pc = pc.fold_norm_weight(_t(w, dev))
        y = ops.w4a16_gemv_fused(h0, pc, nat.PROLOGUE_RMSNORM, 1e-6, parts_in=parts, x_out=h1)
        x_c = po.rmsnorm(_np(h1_ref)[None, :].astype(np.float32), w, 1e-6)
        assert torch.equal(h1, h1_ref)
    elif consumer == 1:    # plain, unsplit (x_out asked for)
        y = ops.w4a16_gemv_fused(h0, pc, 0, parts_in=parts, x_out=h1)
        x_c = _np(h1_ref)[None, :]
        assert torch.equal(h1, h1_ref)
    else:                  # plain, free to split K
        y = ops.w4a16_gemv_fused(h0, pc, 0, parts_in=parts)
        x_c = _np(h1_ref)[None, :]
    ideal = po.paro_linear_merged(x_c, Lc["qweight"], Lc["qzeros"], Lc["scales"], Lc["theta"], Lc["pairs"], Lc["channel_scales"], sizes, None,
                                  group_size=gs, ideal=True)
    got = _np(y)
    assert y.dtype == dt and np.isfinite(got).all()
    assert po.rel_err(got, ideal) < tol
    ops.check_workspace(pc.workspace)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, TIGHT_F16), (torch.bfloat16, TIGHT_BF16)])
@pytest.mark.parametrize("Hq,Hkv,hd,qk_norm,pos", [(32, 8, 128, True, 300), (16, 2, 128, False, 0), (8, 4, 64, True, 37)])
def test_qkv_partial_sums_completed_by_attention(dev, dtype, tol, Hq, Hkv, hd, qk_norm, pos):
    """The qkv projection with the RMSNorm prologue as a 2-way K-split producer: un-normalised partial sums + the K-slices' sums of squares
    (row N); the attention kernel completes q / k / v as it reads them.  Against the oracle (norm -> linear -> attention) and against
    the attention kernel on the ordinary projection's output."""
    from paroquant_amd import ops, _native as nat
    K, sizes = 2560, [Hq * hd, Hkv * hd, Hkv * hd]
    N = sum(sizes)
    L = po.make_layer(Hq + pos, K, sizes)
    rng = np.random.default_rng(Hq * 7 + pos)
    w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
    pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
    x = _t((rng.standard_normal((1, K)) * 2.0).astype(np.float32), dev, dtype)
    assert ops.gemv_parts_count(pk, dtype) == 2
    parts = torch.full((N + 1, 4), float("nan"), device=dev, dtype=torch.float32)
    ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, parts_out=parts)
    y_ref = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6)                      # the ordinary launch: unsplit, norm applied in its epilogue
    torch.cuda.synchronize()
    assert torch.isfinite(parts).all() and (parts[:, 2:] == 0).all()
    ss = float(parts[N].sum())
    assert abs(ss - float((x.float() ** 2).sum())) < 1e-4 * ss                           # row N: the K-slices' sums of squares
    y_c = (parts[:N].sum(1) * torch.rsqrt(parts[N].sum() / K + 1e-6)).to(dtype)
    assert po.rel_err(_np(y_c)[None, :], _np(y_ref)) < (1e-3 if dtype == torch.float16 else 1e-2)
    ideal = po.paro_linear_merged(po.rmsnorm(_np(x).astype(np.float32), w, 1e-6), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, None, ideal=True)
    assert po.rel_err(_np(y_c)[None, :], ideal) < tol
    with pytest.raises(ValueError, match="float32"):                                     # the RMSNorm producer needs the extra row
        ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, parts_out=parts[:N])
    # ---- attention on the partial sums vs on the ordinary projection's output
    T = 512
    kc = _t(rng.standard_normal((Hkv, T, hd)).astype(np.float32), dev, dtype)
    vc = _t(rng.standard_normal((Hkv, hd, T)).astype(np.float32), dev, dtype)
    qw = _t((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32), dev, dtype) if qk_norm else None
    kw = _t((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32), dev, dtype) if qk_norm else None
    cos, sin = po.rope_tables(hd, T, 1e4)
    rope = torch.from_numpy(np.concatenate([cos, sin], axis=-1).astype(np.float32)).to(dev)
    pt = torch.tensor([pos], dtype=torch.int32, device=dev)
    k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    o_parts = ops.attn_decode(parts, k1, v1, pt, rope, Hq, Hkv, hd, qw, kw, 1e-6, norm_dim=K, norm_eps=1e-6)
    o_ref = ops.attn_decode(y_c.contiguous(), k2, v2, pt, rope, Hq, Hkv, hd, qw, kw, 1e-6)
    torch.cuda.synchronize()
    # (the kernel's rsqrt is the hardware approximation: an element may round the other way now and then)
    atol = 2e-3 if dtype == torch.float16 else 2e-2
    assert po.rel_err(_np(o_parts)[None, :], _np(o_ref)[None, :]) < atol
    assert po.rel_err(_np(k1[:, pos]), _np(k2[:, pos])) < atol and po.rel_err(_np(v1[:, :, pos]), _np(v2[:, :, pos])) < atol
    if pos > 0:
        assert torch.equal(k1[:, :pos], kc[:, :pos]) and torch.equal(v1[:, :, :pos], vc[:, :, :pos])      # the rest of the cache is untouched
    ref, _, _ = po.attention_decode(_np(y_c), _np(kc), np.ascontiguousarray(_np(vc).transpose(0, 2, 1)), pos, Hq, Hkv, hd, cos, sin,
                                    None if qw is None else _np(qw), None if kw is None else _np(kw), 1e-6)
    assert po.rel_err(_np(o_parts), ref) < (4e-3 if dtype == torch.float16 else 3e-2)
    # norm_dim = 0: plain sums (a projection without the norm prologue)
    p0 = torch.zeros(N + 1, 4, device=dev, dtype=torch.float32)
    p0[:N, 0] = y_c.float() * 0.25
    p0[:N, 1] = y_c.float() * 0.75
    o_plain = ops.attn_decode(p0, kc.clone(), vc.clone(), pt, rope, Hq, Hkv, hd, qw, kw, 1e-6)
    assert po.rel_err(_np(o_plain)[None, :], _np(o_ref)[None, :]) < atol


@pytest.mark.parametrize("K,sizes,n", [(1024, [256, 64, 64], 2), (1024, [512], 4), (384, [48, 16], 3), (4096, [1024, 512, 512], 2), (2048, [2560], 0)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rmsnorm_producer_leaves_sums_of_squares(dev, K, sizes, n, dtype):
    """RMSNorm prologue on a launch that leaves partial sums: un-normalised partials + the K-slices' sums of squares in row N, for splits
    with fewer groups than waves, ragged partitions, forced and automatic split counts -- completed on the host they equal the ordinary
    launch (norm applied in its own epilogue)."""
    from paroquant_amd import ops, _native as nat
    N = sum(sizes)
    L = po.make_layer(K + N + n, K, sizes)
    rng = np.random.default_rng(K + n)
    w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
    pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
    x = _t((rng.standard_normal((1, K)) * 3.0).astype(np.float32), dev, dtype)
    n_eff = n or ops.gemv_parts_count(pk, dtype)
    assert n_eff >= 2
    parts = torch.full((N + 1, 4), float("nan"), device=dev, dtype=torch.float32)
    ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, parts_out=parts, parts_n=n)
    y_ref = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6)
    torch.cuda.synchronize()
    assert torch.isfinite(parts).all() and (parts[:, n_eff:] == 0).all() and (parts[N, :n_eff] > 0).all()
    ss = float(parts[N].sum())
    assert abs(ss - float((x.float() ** 2).sum())) < 1e-4 * ss
    y_c = (parts[:N].sum(1) * torch.rsqrt(parts[N].sum() / K + 1e-6)).to(dtype)
    assert po.rel_err(_np(y_c)[None, :], _np(y_ref)) < (1e-3 if dtype == torch.float16 else 1e-2)

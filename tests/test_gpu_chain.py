"""GPU parity of the decode-chain GEMV (``paro_w4a16_gemv_chain``, csrc/chain_impl.hpp) against the CPU oracle.

The chain computes the SAME function as the reference's per-linear ``rotate -> dequant matmul``
(transformers/modules.py:57-71, vllm/plugin.py:281-311); only WHERE the rotation runs differs (the epilogue of the
launch that produces the activation).  Every test therefore compares with ``oracle.paro_linear_merged`` applied linear
by linear on the un-rotated activations (float64 ideal), never with the in-kernel-rotation path of this library only.
"""
import numpy as np
import pytest
import torch

from oracle import paro_oracle as po

pytestmark = pytest.mark.gpu

TIGHT_F16 = 3e-3
TIGHT_BF16 = 8e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    import paroquant_amd  # noqa: F401
    from paroquant_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _packed(L, dev):
    from paroquant_amd.linear import PackedParoWeights
    return PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev),
                             _t(L["pairs"], dev), _t(L["channel_scales"], dev), L["sizes"])


def _ideal(L, x, bias=None):
    return po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"],
                                 L["sizes"], bias, ideal=True)


def _h(a):
    """Round to fp16 and back (what an fp16 activation buffer holds)."""
    return np.asarray(a, dtype=np.float64).astype(np.float16)


SHAPES = [
    (2560, [4096, 1024, 1024]),    # Qwen3-4B qkv
    (4096, [2560]),                # Qwen3-4B o
    (2560, [9728, 9728]),          # Qwen3-4B gate_up
    (9728, [2560]),                # Qwen3-4B down
    (4096, [4096, 1024, 1024]),    # Llama-3-8B qkv
    (14336, [4096]),               # Llama-3-8B down
    (1024, [2048, 1024, 1024]),    # Qwen3-0.6B qkv
    (256, [128]),                  # tiny: 2 groups, 1 block
    (128, [256, 128]),             # one group
]


@pytest.mark.parametrize("K,sizes", SHAPES)
@pytest.mark.parametrize("rows", [1, 3, 4, 7, 16])
def test_chain_linear_matches_oracle(dev, K, sizes, rows):
    """Head of a chain + one linear: rotate_parts -> chain_gemv == oracle linear, automatic launch shape."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows + len(sizes), K, sizes)
    pk = _packed(L, dev)
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    xr = ops.rotate_parts(_t(x, dev), pk)
    y, nx = ops.chain_gemv(xr, pk)
    assert nx is None
    assert po.rel_err(_np(y), _ideal(L, x)) < TIGHT_F16
    torch.cuda.synchronize()
    ops.check_workspace(pk.workspace)


@pytest.mark.parametrize("K,sizes", [(4096, [2560]), (2560, [4096, 1024, 1024]), (1024, [256])])
@pytest.mark.parametrize("ksplit,waves", [(1, 4), (1, 8), (2, 4), (3, 8), (8, 4), (16, 4), (16, 8)])
@pytest.mark.parametrize("rows", [1, 5])
def test_chain_launch_shapes(dev, K, sizes, ksplit, waves, rows):
    """Every K-split / workgroup size computes the same function; repeated launches on one workspace advance the
    per-block epochs (no re-arm store), interleaved with the tag-1 K-split of the fused GEMV on the SAME workspace."""
    from paroquant_amd import ops
    L = po.make_layer(K + ksplit, K, sizes)
    pk = _packed(L, dev)
    rng = np.random.default_rng(ksplit * 10 + waves)
    gps = -(-(K // 128) // min(ksplit, K // 128))
    grid = (sum(sizes) // 128) * -(-(K // 128) // gps)
    if grid > 256:
        # a caller-fixed split whose grid cannot be resident at once is refused (the automatic one shrinks instead); how many
        # workgroups a CU holds depends on the instantiation, one per CU is the guaranteed minimum
        try:
            ops.chain_gemv(ops.rotate_parts(torch.zeros(rows, K, device=dev, dtype=torch.float16), pk), pk, ksplit=ksplit, waves=waves)
        except RuntimeError as e:
            assert "resident at once" in str(e)
            return
    for it in range(6):
        x = rng.standard_normal((rows, K)).astype(np.float16)
        xr = ops.rotate_parts(_t(x, dev), pk)
        y, _ = ops.chain_gemv(xr, pk, ksplit=ksplit, waves=waves)
        assert po.rel_err(_np(y), _ideal(L, x)) < TIGHT_F16
        if it % 2 == 1 and rows <= 4:     # the old family's K-split on the same granule area in between
            y2 = ops.w4a16_gemv_tuned(_t(x, dev), pk, 4, 4, 8, 0)
            assert po.rel_err(_np(y2), _ideal(L, x)) < TIGHT_F16
    torch.cuda.synchronize()
    ops.check_workspace(pk.workspace)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows", [1, 2, 6])
def test_chain_producer_rotation_bench_chain(dev, dtype, rows):
    """The bench's chain on small dims: qkv -> (q part) -> o -> gate_up -> (gate part) -> down -> next qkv, every hand-over
    through the producer's epilogue rotation; each stage against the oracle applied to the previous stage's actual output."""
    from paroquant_amd import ops
    h, inter, q, kv = 512, 1280, 768, 256
    Ls = [po.make_layer(1, h, [q, kv, kv]), po.make_layer(2, q, [h]), po.make_layer(3, h, [inter, inter]), po.make_layer(4, inter, [h]),
          po.make_layer(5, h, [q, kv, kv])]
    pks = [_packed(L, dev) for L in Ls]
    tol = TIGHT_F16 if dtype == torch.float16 else TIGHT_BF16
    xt = _t(np.random.default_rng(rows).standard_normal((rows, h)).astype(np.float32), dev, dtype)
    x = _np(xt)
    xr = ops.rotate_parts(xt, pks[0])
    # every stage against the float64 oracle linear applied to the PREVIOUS stage's actual (activation-dtype) output
    g0, x1 = ops.chain_gemv(xr, pks[0], next_pk=pks[1], next_col0=0)
    y0 = _np(g0)
    assert po.rel_err(y0, _ideal(Ls[0], x)) < tol
    # the handed-over vector itself: rotate_1(y0[:, :q] * cs_1) -- against the oracle's rotation
    ref_x1 = po.rotate(y0[:, :q], Ls[1]["pairs"][0], Ls[1]["theta"][0].astype(np.float64),
                       Ls[1]["channel_scales"][0].reshape(-1).astype(np.float64), 128, mode="ideal")
    assert po.rel_err(_np(x1[0]), ref_x1) < tol
    g1, x2 = ops.chain_gemv(x1, pks[1], next_pk=pks[2])
    y1 = _np(g1)
    assert po.rel_err(y1, _ideal(Ls[1], y0[:, :q])) < tol
    assert tuple(x2.shape) == (2, rows, h)
    g2, x3 = ops.chain_gemv(x2, pks[2], next_pk=pks[3], next_col0=0)
    y2 = _np(g2)
    assert po.rel_err(y2, _ideal(Ls[2], y1)) < tol
    g3, x4 = ops.chain_gemv(x3, pks[3], next_pk=pks[4])
    y3 = _np(g3)
    assert po.rel_err(y3, _ideal(Ls[3], y2[:, :inter])) < tol
    assert tuple(x4.shape) == (3, rows, h)
    g4, _ = ops.chain_gemv(x4, pks[4])
    assert po.rel_err(_np(g4), _ideal(Ls[4], y3)) < tol
    # a consumer that reads a block range in the MIDDLE of the producer's output (the k part of qkv -> a 256-wide linear)
    Lk = po.make_layer(9, kv, [384])
    pkk = _packed(Lk, dev)
    _, xk = ops.chain_gemv(xr, pks[0], next_pk=pkk, next_col0=q, write_y=False)
    gk, _ = ops.chain_gemv(xk, pkk)
    assert po.rel_err(_np(gk), _ideal(Lk, y0[:, q:q + kv])) < tol
    torch.cuda.synchronize()
    ops.check_workspace(pks[0].workspace)


@pytest.mark.parametrize("h,inter", [(512, 1280), (2560, 9728), (1024, 3072)])
@pytest.mark.parametrize("rows", [1, 4, 11])
def test_chain_decoder_layer_norms_residual_silu(dev, h, inter, rows):
    """The real decoder's MLP half through the chain: o (+ residual, sum of squares out, gate_up's two rotations) ->
    gate_up (RMSNorm scalar from the sums, silu(gate) * up + down's rotation in the epilogue, K-split) -> down (+ residual,
    next layer's qkv rotations) -- against oracle linears around explicit RMSNorm / SiLU * mul / residual adds."""
    from paroquant_amd import ops, _native as nat
    q = 384
    Lo, Lgu, Ld, Lq = po.make_layer(11, q, [h]), po.make_layer(12, h, [inter, inter]), po.make_layer(13, inter, [h]), po.make_layer(14, h, [q, 128, 128])
    rng = np.random.default_rng(h + rows)
    w_post = (1.0 + 0.2 * rng.standard_normal(h)).astype(np.float16)
    w_in = (1.0 + 0.2 * rng.standard_normal(h)).astype(np.float16)
    po_, pgu, pd, pq = _packed(Lo, dev), _packed(Lgu, dev).fold_norm_weight(_t(w_post, dev)), _packed(Ld, dev), _packed(Lq, dev).fold_norm_weight(_t(w_in, dev))
    attn = rng.standard_normal((rows, q)).astype(np.float16)
    res = (rng.standard_normal((rows, h)) * 2.0).astype(np.float16)
    # chain; every stage against the float64 oracle applied to the previous stage's actual fp16 output
    xa = ops.rotate_parts(_t(attn, dev), po_)
    ssq1 = torch.zeros(rows, h // 128, device=dev)
    g_h2, x_gu = ops.chain_gemv(xa, po_, residual=_t(res, dev), ssq_out=ssq1, next_pk=pgu)
    h2 = _np(g_h2)
    assert po.rel_err(h2, _ideal(Lo, attn) + res.astype(np.float64)) < TIGHT_F16
    np.testing.assert_allclose(_np(ssq1).sum(1), (h2 ** 2).sum(1), rtol=1e-5)
    g_gu, x_d = ops.chain_gemv(x_gu, pgu, ssq_in=ssq1, norm_dim=h, eps=1e-6, next_pk=pd, act=nat.CHAIN_ACT_SILU_MUL)
    gu = _np(g_gu)
    assert po.rel_err(gu, _ideal(Lgu, po.rmsnorm(h2, w_post, 1e-6))) < TIGHT_F16
    ssq2 = torch.zeros(rows, h // 128, device=dev)
    g_h3, x_q = ops.chain_gemv(x_d, pd, residual=g_h2, ssq_out=ssq2, next_pk=pq)
    h3 = _np(g_h3)
    assert po.rel_err(h3, _ideal(Ld, po.silu_mul(gu, inter)) + h2) < TIGHT_F16
    g_qkv, _ = ops.chain_gemv(x_q, pq, ssq_in=ssq2, norm_dim=h, eps=1e-6)
    assert po.rel_err(_np(g_qkv), _ideal(Lq, po.rmsnorm(h3, w_in, 1e-6))) < TIGHT_F16
    # the same last linear through the in-kernel-rotation family (RMSNorm prologue): two routes, one function
    if rows <= 4:
        alt = ops.w4a16_gemv_fused(g_h3, pq, nat.PROLOGUE_RMSNORM, 1e-6)
        assert po.rel_err(_np(g_qkv), _np(alt)) < TIGHT_F16
    torch.cuda.synchronize()
    ops.check_workspace(po_.workspace)


def test_chain_graph_replay_and_epoch_wrap(dev):
    """A captured chain replays (the epochs live in device memory, the tags change per replay) and survives the epoch
    counter's wrap: the per-block epoch words are preset to just below 2^20."""
    from paroquant_amd import ops
    K, sizes = 4096, [1024]
    L = po.make_layer(77, K, sizes)
    L2 = po.make_layer(78, 1024, [512, 512])
    pk, pk2 = _packed(L, dev), _packed(L2, dev)
    x = np.random.default_rng(5).standard_normal((1, K)).astype(np.float16)
    xt = _t(x, dev)
    xr = torch.empty(1, 1, K, device=dev, dtype=torch.float16)
    y = torch.empty(1, 1024, device=dev, dtype=torch.float16)
    nx = torch.empty(2, 1, 1024, device=dev, dtype=torch.float16)
    y2 = torch.empty(1, 1024, device=dev, dtype=torch.float16)

    def step():
        ops.rotate_parts(xt, pk, out=xr)
        ops.chain_gemv(xr, pk, out=y, next_pk=pk2, next_x=nx, ksplit=8)
        ops.chain_gemv(nx, pk2, out=y2, ksplit=4)

    step()
    torch.cuda.synchronize()
    ref1 = _ideal(L, x)
    ref2 = _ideal(L2, _h(ref1))
    assert po.rel_err(_np(y), ref1) < TIGHT_F16 and po.rel_err(_np(y2), ref2) < 2 * TIGHT_F16
    pk.workspace[:16 * 4].view(torch.int32).fill_((1 << 20) - 3)     # epochs of blocks 0..15: three launches from the wrap
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for i in range(12):
        x2 = np.random.default_rng(100 + i).standard_normal((1, K)).astype(np.float16)
        xt.copy_(_t(x2, dev))
        g.replay()
        torch.cuda.synchronize()
        r1 = _ideal(L, x2)
        assert po.rel_err(_np(y), r1) < TIGHT_F16
        assert po.rel_err(_np(y2), _ideal(L2, _h(r1))) < 2 * TIGHT_F16
    ops.check_workspace(pk.workspace)
    pk.workspace[:16 * 4].zero_()


def test_chain_argument_errors(dev):
    from paroquant_amd import ops, _native as nat
    L = po.make_layer(1, 256, [128, 48])       # a partition that is not a multiple of 128 columns
    pk = _packed(L, dev)
    x = torch.zeros(2, 1, 256, device=dev, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="128-column blocks"):
        ops.chain_gemv(x, pk)
    L = po.make_layer(2, 256, [256])
    pk = _packed(L, dev)
    with pytest.raises(ValueError, match="x_rot must be"):
        ops.chain_gemv(torch.zeros(1, 256, device=dev, dtype=torch.float16), pk)
    with pytest.raises(RuntimeError, match="1..16 rows"):
        ops.chain_gemv(torch.zeros(1, 17, 256, device=dev, dtype=torch.float16), pk)
    Ln = po.make_layer(3, 512, [128])
    with pytest.raises(RuntimeError, match="must lie inside"):
        ops.chain_gemv(torch.zeros(1, 1, 256, device=dev, dtype=torch.float16), pk, next_pk=_packed(Ln, dev))
    with pytest.raises(RuntimeError, match="merged gate"):
        ops.chain_gemv(torch.zeros(1, 1, 256, device=dev, dtype=torch.float16), pk, next_pk=_packed(po.make_layer(4, 128, [128]), dev),
                       act=nat.CHAIN_ACT_SILU_MUL)
    L64 = po.make_layer(5, 256, [128], group_size=64)
    with pytest.raises(RuntimeError, match="group_size 128"):
        ops.chain_gemv(torch.zeros(1, 1, 256, device=dev, dtype=torch.float16), _packed(L64, dev))


@pytest.mark.parametrize("rows", [1, 3])
def test_gelu_tanh_mul_gemma_mlp_both_families(dev, rows):
    """The Gemma MLP (gelu_tanh(gate) * up, RMSNorm weights stored as w with y = x_hat (1 + w)): the fused family's
    GELU_TANH_MUL prologue and the chain family's epilogue activation, both against the oracle."""
    from paroquant_amd import ops, _native as nat
    h, inter = 512, 1280
    Lgu, Ld = po.make_layer(31, h, [inter, inter]), po.make_layer(32, inter, [h])
    rng = np.random.default_rng(rows)
    w = (0.1 * rng.standard_normal(h)).astype(np.float16)                 # stored weight: the norm multiplies by (1 + w)
    pgu, pd = _packed(Lgu, dev).fold_norm_weight(_t(w, dev), plus_one=True), _packed(Ld, dev)
    x = (rng.standard_normal((rows, h)) * 2.0).astype(np.float16)
    # fused family: gate_up with the RMSNorm prologue, down with the GELU prologue
    gu = ops.w4a16_gemv_fused(_t(x, dev), pgu, nat.PROLOGUE_RMSNORM, 1e-6)
    ref_gu = _ideal(Lgu, po.rmsnorm(x, 1.0 + w.astype(np.float64), 1e-6))
    assert po.rel_err(_np(gu), ref_gu) < TIGHT_F16
    y = ops.w4a16_gemv_fused(gu, pd, nat.PROLOGUE_GELU_TANH_MUL)
    ref_y = _ideal(Ld, po.gelu_tanh_mul(_np(gu), inter))
    assert po.rel_err(_np(y), ref_y) < TIGHT_F16
    assert po.rel_err(_np(y), _ideal(Ld, po.silu_mul(_np(gu), inter))) > 5e-2        # ... and it is not SiLU
    # chain family: the same two linears, the activation in gate_up's epilogue
    xn = po.rmsnorm(x, 1.0 + w.astype(np.float64), 1e-6).astype(np.float16)
    Lgu_plain = _packed(Lgu, dev)
    g2, x_d = ops.chain_gemv(ops.rotate_parts(_t(xn, dev), Lgu_plain), Lgu_plain, next_pk=pd, act=nat.CHAIN_ACT_GELU_TANH_MUL)
    y2, _ = ops.chain_gemv(x_d, pd)
    assert po.rel_err(_np(y2), _ideal(Ld, po.gelu_tanh_mul(_np(g2), inter))) < TIGHT_F16

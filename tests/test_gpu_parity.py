"""GPU parity tests (run on an MI355X with ``-m gpu``): every HIP kernel, called through the C ABI
(ctypes -> libparo_mi355x.so), against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): outputs within 1e-2 relative of the CPU
dequant-then-fp16-matmul oracle, measured as max|y - ref| / max|ref|.  Integer/byte work (the
repack and the (q - z) * s dequant) must be bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import paro_oracle as po

pytestmark = pytest.mark.gpu

REL_TOL = 1e-2          # the north-star gate
TIGHT_F16 = 3e-3        # what we actually expect for fp16 activations
TIGHT_BF16 = 8e-3       # bf16 activations carry 8 mantissa bits: one rounding of the rotated input + one of the output (was 2e-2)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    import paroquant_amd  # noqa: F401
    from paroquant_amd import _native
    _native.load()        # the product path must fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _packed(L, dev, bias=None):
    from paroquant_amd.linear import PackedParoWeights
    return PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev),
                             _t(L["pairs"], dev), _t(L["channel_scales"], dev), L["sizes"],
                             None if bias is None else _t(bias, dev))


# ---------------------------------------------------------------- rotation::rotate

# tol: against the float64 ideal (ONE rounding to the activation type here); ftol: against the reference-faithful mode, whose
# bf16 leg is the lossy one -- the reference casts theta and the channel scales to bf16 (rotation.cu:75-78) and re-rounds the state
# after every stage (rotation.cuh:143-153): 8 stages x 2^-9 plus a 2^-9 angle error is what 6e-2 allows for, not this kernel
@pytest.mark.parametrize("dtype,mode,tol,ftol", [(torch.float16, "f16", 4e-3, 8e-3), (torch.bfloat16, "bf16", 1e-2, 6e-2),
                                                 (torch.float32, "f32", 2e-5, 4e-5)])
@pytest.mark.parametrize("rows,hidden,gs,krot,with_scale", [
    (1, 128, 128, 8, True), (1, 4096, 128, 8, True), (3, 512, 128, 8, False), (4, 2560, 128, 8, True),
    (7, 1024, 128, 1, True), (33, 1024, 128, 3, True), (5, 256, 64, 8, True), (2, 192, 64, 1, False),
    (4100, 256, 128, 8, True), (0, 256, 128, 8, True),
])
def test_rotate_matches_oracle(dev, dtype, mode, tol, ftol, rows, hidden, gs, krot, with_scale):
    rng = np.random.default_rng(rows * 7919 + hidden + krot)
    x = rng.standard_normal((rows, hidden)).astype(np.float32)
    G = hidden // gs
    idx = np.stack([np.concatenate([rng.permutation(gs) for _ in range(G)]) for _ in range(krot)]).astype(np.int16)
    theta = (rng.standard_normal((krot, hidden // 2)) * 0.3).astype(np.float16)
    sc = rng.uniform(0.5, 2, hidden).astype(np.float16) if with_scale else None
    xt = _t(x, dev, dtype)
    out = torch.ops.rotation.rotate(xt, _t(idx, dev), _t(theta, dev), None if sc is None else _t(sc[None, :], dev), gs)
    assert out.shape == xt.shape and out.dtype == dtype
    if rows == 0:
        return
    xin = _np(xt)
    ideal = po.rotate(xin, idx, theta.astype(np.float64), None if sc is None else sc.astype(np.float64), gs, "ideal")
    faithful = po.rotate(xin, idx, theta, sc, gs, mode)
    got = _np(out)
    assert po.rel_err(got, ideal) < tol
    assert po.rel_err(got, faithful) < ftol
    if dtype != torch.float32:   # one rounding at the end: at least as close to the ideal as the reference-faithful path
        assert po.rel_err(got, ideal) <= po.rel_err(faithful, ideal) * 1.5 + 1e-4


@pytest.mark.parametrize("dtype,mode,tol,ftol", [(torch.float16, "f16", 4e-3, 8e-3), (torch.bfloat16, "bf16", 1e-2, 6e-2)])
@pytest.mark.parametrize("rows,K,sizes,krot", [(1, 128, [16], 8), (5, 1024, [256, 64], 8), (17, 2560, [512, 128, 128], 8), (32, 512, [64], 3),
                                                (33, 1536, [400, 112], 1), (100, 256, [32, 32], 8), (255, 384, [48], 8)])
def test_schedule_prepass_matches_oracle(dev, dtype, mode, tol, ftol, rows, K, sizes, krot):
    """The rotation as a launch of its own below 256 rows (rotate.hip `prerot_sched_kernel`, round 6: one wave per (partition, group, 4 rows)
    on the PACKED schedule, the in-kernel rotation's arithmetic) through `paro_rotate_parts` -- plain rows [P][rows][K] -- against the
    oracle's rotate per merged partition (rotation.cuh:91-173; one rotation per partition: plugin.py:288-306): the float64 ideal and the
    reference-faithful mode, the tolerances of `test_rotate_matches_oracle`.  Ragged row counts (not multiples of 4), 1..3 partitions,
    short schedules (krot 1 / 3)."""
    from paroquant_amd import ops
    L = po.make_layer(rows * 31 + K + krot, K, sizes, krot=krot)
    pk = _packed(L, dev)
    x = np.random.default_rng(rows + K).standard_normal((rows, K)).astype(np.float32)
    xt = _t(x, dev, dtype)
    xr = ops.rotate_parts(xt, pk)
    assert xr.shape == (len(sizes), rows, K) and xr.dtype == dtype and torch.isfinite(xr.float()).all()
    xin = _np(xt)
    for p in range(len(sizes)):
        cs = L["channel_scales"][p].reshape(-1)
        ideal = po.rotate(xin, L["pairs"][p], L["theta"][p].astype(np.float64), cs.astype(np.float64), 128, "ideal")
        faithful = po.rotate(xin, L["pairs"][p], L["theta"][p], cs, 128, mode)
        got = _np(xr[p])
        assert po.rel_err(got, ideal) < tol, p
        assert po.rel_err(got, faithful) < ftol, p
    # and the stage kernel behind rotation::rotate agrees with it to the rotation's last-place rounding
    ref = torch.ops.rotation.rotate(xt, pk.pairs[0], pk.theta[0], pk.channel_scales.reshape(len(sizes), 1, K)[0])
    assert po.rel_err(_np(xr[0]), _np(ref).astype(np.float64)) < tol


def test_rotate_kats_and_errors(dev):
    # K1 quarter turn (rotation.cuh:55-56), K2 zero theta
    x = torch.arange(1, 129, device=dev, dtype=torch.float32)[None, :] / 16
    idx = torch.arange(128, device=dev, dtype=torch.int16)[None, :]
    th = torch.zeros(1, 64, device=dev); th[0, 0] = np.pi / 2
    y = torch.ops.rotation.rotate(x, idx, th)
    assert abs(y[0, 0].item() - x[0, 1].item()) < 1e-5 and abs(y[0, 1].item() + x[0, 0].item()) < 1e-5
    assert torch.equal(y[0, 2:], x[0, 2:])
    # 3-D input keeps its shape; non-contiguous input is handled
    x3 = torch.randn(2, 3, 256, device=dev, dtype=torch.float16)
    idx2 = _t(po.random_pairs(np.random.default_rng(0), 8, 256), dev)
    th2 = torch.zeros(8, 128, device=dev, dtype=torch.float16)
    assert torch.equal(torch.ops.rotation.rotate(x3, idx2, th2), x3)
    xt = x3.transpose(0, 1)
    assert torch.equal(torch.ops.rotation.rotate(xt, idx2, th2), xt)
    # error behaviour of rotate_dynamic / rotate_launcher (rotation.cu:66,114,123)
    with pytest.raises(RuntimeError, match="must equal"):
        torch.ops.rotation.rotate(x3, idx2, th2[:4])
    with pytest.raises(RuntimeError, match="group_size"):
        torch.ops.rotation.rotate(x3, idx2, th2, None, 32)
    with pytest.raises(RuntimeError, match="divisible"):
        torch.ops.rotation.rotate(torch.zeros(1, 200, device=dev, dtype=torch.float16),
                                  torch.zeros(8, 200, device=dev, dtype=torch.int16),
                                  torch.zeros(8, 100, device=dev, dtype=torch.float16))


def test_rotate_inverse_and_norm_full_size(dev):
    """Size-independent properties at a BASELINE-size prefill shape (M = 8192, K = 4096)."""
    rng = np.random.default_rng(3)
    K = 4096
    idx = _t(po.random_pairs(rng, 8, K), dev)
    th = _t((rng.standard_normal((8, K // 2)) * 0.3).astype(np.float32), dev)
    x = torch.randn(8192, K, device=dev, dtype=torch.float32)
    y = torch.ops.rotation.rotate(x, idx, th)
    n0 = x.view(8192, -1, 128).norm(dim=-1)
    n1 = y.view(8192, -1, 128).norm(dim=-1)
    assert torch.allclose(n0, n1, rtol=1e-4, atol=1e-4)
    back = torch.ops.rotation.rotate(y, torch.flip(idx, [0]), -torch.flip(th, [0]))
    assert (back - x).abs().max().item() < 2e-4


# ---------------------------------------------------------------- repack / dequant (bit-exact)

@pytest.mark.parametrize("K,sizes", [(128, [16]), (256, [64]), (512, [208, 48, 16]), (4096, [1024])])
def test_repack_dequant_bit_exact(dev, K, sizes):
    """Integer/byte work: the repack + (q - z) * s dequant are bit-exact against the oracle, and the
    packed words match the layout spec of include/paro_abi.h."""
    from paroquant_amd import ops
    N = sum(sizes)
    L = po.make_layer(K + N, K, sizes)
    wq, sz = torch.ops.paro.repack_awq(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), sizes)
    w = ops.dequant_packed(wq, sz, K, sizes, torch.float16).cpu().numpy()
    ref = po.dequant_awq(L["qweight"], L["qzeros"], L["scales"], 128, np.float16)
    assert np.array_equal(w.view(np.uint16), ref.view(np.uint16))
    q = po.unpack_awq(L["qweight"]).astype(np.uint32)          # [K, N]
    got = wq.cpu().numpy().view(np.uint32).reshape(N // 16, K // 128, 64, 4)
    t, g, lane, i = N // 16 - 1, K // 128 - 1, 37, 2
    n, kb = lane & 15, lane >> 4
    word = 0
    for e in range(8):
        word |= int(q[g * 128 + 32 * i + 8 * kb + e, t * 16 + n]) << (4 * ((e >> 1) + 4 * (e & 1)))
    assert int(got[t, g, lane, i]) == word
    # scale/zero words: padded tile space, partition p starts at sum(pad8(tiles_q), q < p)
    z = po.unpack_awq(L["qzeros"]).astype(np.uint32)
    tsz = sum((s // 16 + 7) // 8 * 8 for s in sizes)
    szw = sz.cpu().numpy().view(np.uint32).reshape(K // 128, tsz // 4, 16, 4)
    col0, ts0 = 0, 0
    for s_ in sizes:
        for (lt, nn) in ((0, 3), (s_ // 16 - 1, 15)):
            col = col0 + lt * 16 + nn
            ts = ts0 + lt
            wv = int(szw[g, ts // 4, nn, ts % 4])
            assert wv & 0xffff == int(L["scales"][g, col].view(np.uint16))
            assert np.array([wv >> 16], dtype=np.uint16).view(np.float16)[0] == int(z[g, col])
        col0 += s_
        ts0 += (s_ // 16 + 7) // 8 * 8


def _run_exchange_schedule(words, x, krot):
    """Host model of the register exchange network the GEMV executes (gemv_impl.hpp `stage` / `finish`) on
    one 128-channel group: words uint32 [768] from paro_pack_rotation, x float64 [128]."""
    def coef(w):                                              # Q << 16 | P, signed 16-bit units of 2^-14
        P = (w & 0xffff).astype(np.uint16).view(np.int16).astype(np.float64) / 16384.0
        Q = (w >> 16).astype(np.uint16).view(np.int16).astype(np.float64) / 16384.0
        return P, Q
    ch = words.reshape(3, 64, 4)
    A, B = x[0::2].copy(), x[1::2].copy()                     # lane l starts with channels 2l, 2l+1
    for t in range(krot):
        P, Q = coef(ch[t >> 2, :, t & 3])
        assert np.all(np.abs(P * P + Q * Q - 1) < 2e-4)
        src4 = (ch[2, :, t >> 2] >> (8 * (t & 3))) & 0xff
        assert not np.any(src4 & 3) and sorted((src4 >> 2).tolist()) == list(range(64))   # a permutation of the lanes
        keep, give = P * A + Q * B, P * B - Q * A
        A, B = keep, give[src4 >> 2]                          # ds_bpermute: pull give' of lane src
    P, Q = coef(ch[2, :, 2])
    f = ch[2, :, 3]
    sigma = np.where(f >> 31, -1.0, 1.0)
    out = np.full(128, np.nan)
    out[(f & 0xfe) >> 1] = P * A + Q * B
    out[((f >> 8) & 0xfe) >> 1] = sigma * (P * B - Q * A)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("krot", [1, 2, 5, 8])
def test_pack_rotation_schedule(dev, krot):
    """The packed exchange schedule, run on the host, IS the rotation of the checkpoint (float64 oracle):
    every lane fetches from a permutation of the lanes, every channel is written exactly once."""
    K, P = 512, 2
    rng = np.random.default_rng(100 + krot)
    pairs = np.stack([po.random_pairs(rng, krot, K) for _ in range(P)])
    theta = (rng.standard_normal((P, krot, K // 2)) * 0.7).astype(np.float16)
    theta[0, 0, :8] = 0                                          # dummy pairs (angle 0, optim/rotation.py:53)
    rot = torch.ops.paro.pack_rotation(_t(pairs, dev), _t(theta, dev)).cpu().numpy().view(np.uint32)
    rot = rot.reshape(P, K // 128, 768)
    x = rng.standard_normal(K)
    for pp in range(P):
        want = po.rotate(x[None], pairs[pp], theta[pp].astype(np.float64), None, 128, "ideal")[0]
        for g in range(K // 128):
            got = _run_exchange_schedule(rot[pp, g], x[g * 128:(g + 1) * 128], krot)
            assert not np.isnan(got).any()                        # every channel written exactly once
            np.testing.assert_allclose(got, want[g * 128:(g + 1) * 128], rtol=0, atol=4e-4)   # 16-bit coefficients: 1.5e-5 each
    # repeated pairs in consecutive stages (lane keeps both members: fetches from itself) and the identity
    pairs2 = np.repeat(pairs[:1, :1], krot, axis=1)
    rot2 = torch.ops.paro.pack_rotation(_t(pairs2, dev), _t(theta[:1], dev)).cpu().numpy().view(np.uint32)
    rot2 = rot2.reshape(1, K // 128, 768)
    want = po.rotate(x[None], pairs2[0], theta[0].astype(np.float64), None, 128, "ideal")[0]
    got = np.concatenate([_run_exchange_schedule(rot2[0, g], x[g * 128:(g + 1) * 128], krot) for g in range(K // 128)])
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-4)


@pytest.mark.gpu
def test_pack_rotation_rejects_illegal_pairs(dev):
    """A stage that is not a perfect matching raises, like the reference's converter ("illegal pair",
    optim/rotation.py:36-37)."""
    rng = np.random.default_rng(5)
    pairs = po.random_pairs(rng, 8, 256)[None]
    theta = torch.zeros(1, 8, 128, dtype=torch.float16, device=dev)
    for bad in ("dup", "self", "range"):
        q = pairs.copy()
        if bad == "dup":
            q[0, 3, 130] = q[0, 3, 140]
        elif bad == "self":
            q[0, 7, 1] = q[0, 7, 0]
        else:
            q[0, 0, 5] = 128
        with pytest.raises((RuntimeError, ValueError), match="illegal pair"):
            torch.ops.paro.pack_rotation(_t(q, dev), theta)


def test_repack_rejects_bad_shapes(dev):
    z = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
    h = lambda *s: torch.zeros(*s, dtype=torch.float16, device=dev)
    with pytest.raises(ValueError):
        torch.ops.paro.repack_awq(z(100, 8), z(1, 8), h(1, 64), [64])        # K % 128
    with pytest.raises(ValueError):
        torch.ops.paro.repack_awq(z(128, 1), z(1, 1), h(1, 8), [8])          # N % 16
    with pytest.raises(ValueError):
        torch.ops.paro.repack_awq(z(128, 8), z(1, 8), h(1, 64), [32, 16])    # sizes do not sum to N


# ---------------------------------------------------------------- fused GEMV (decode)

GEMV_SHAPES = [
    # K, partition sizes
    (4096, [4096]),                 # BASELINE config 1 / Llama-3-8B q_proj, o_proj
    (4096, [4096, 1024, 1024]),     # Llama-3-8B merged qkv (3 rotations)
    (1024, [2048, 1024, 1024]),     # Qwen3-0.6B merged qkv
    (2560, [9728, 9728]),           # Qwen3-4B merged gate_up (2 rotations, 20 groups)
    (9728, [2560]),                 # Qwen3-4B down_proj (76 groups)
    (256, [48, 16]),                # ragged tiny case: 3 + 1 tiles, 2 groups
    (4096, [2560]),                 # Qwen3-4B o_proj (t4 k4 w4 K-split path; the worst roofline shape)
    (2560, [4096, 1024, 1024]),     # Qwen3-4B merged qkv
    (2048, [1024]),                 # Qwen3-0.6B o_proj
    (1024, [3072, 3072]),           # Qwen3-0.6B merged gate_up
    (3072, [1024]),                 # Qwen3-0.6B down_proj
]


@pytest.mark.parametrize("K,sizes", GEMV_SHAPES)
@pytest.mark.parametrize("rows", [1, 3, 8, 16])
def test_gemv_matches_oracle(dev, K, sizes, rows):
    L = po.make_layer(K + sum(sizes) + rows, K, sizes, bias=(rows == 3))
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, K)).astype(np.float16)
    pk = _packed(L, dev, L.get("bias"))
    y = pk.apply(_t(x, dev))
    assert y.shape == (rows, sum(sizes)) and y.dtype == torch.float16
    ref = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                L["channel_scales"], sizes, L.get("bias"), act="f16")
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, L.get("bias"), ideal=True)
    got = _np(y)
    assert po.rel_err(got, ref) < REL_TOL
    assert po.rel_err(got, ideal) < TIGHT_F16
    assert np.isfinite(got).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,sizes", [(1536, [400, 112]), (2560, [4096, 1024, 1024]), (1024, [256])])
def test_gemv_on_caller_rotated_activations(dev, dtype, K, sizes):
    """mode 2 (ABI v10): x arrives already rotated, [n_parts][rows][K] -- here by rotation::rotate with each partition's
    parameters, as a producer kernel's epilogue would -- and the GEMV runs the pre-rotated kernels without a pre-pass: within
    tolerance of the float64 oracle.  The library's own pre-pass route (mode 1) rotates on the packed schedule since round 6
    (rotate.hip `prerot_sched_kernel`: the in-kernel rotation's arithmetic, x handed over in MFMA-fragment order), so on the same
    launch shape it returns the BITS of mode 0 (rotation inside every workgroup) -- and differs from the stage kernel's rotation
    (fp32 state, cos / sin from theta) only in the rotation's last-place rounding."""
    from paroquant_amd import ops
    L = po.make_layer(K + len(sizes), K, sizes, bias=True)
    pk = _packed(L, dev)
    bias = _t(L["bias"], dev, dtype)
    rng = np.random.default_rng(K)
    for rows in (1, 3, 8, 16):
        x = _t(rng.standard_normal((rows, K)).astype(np.float32), dev, dtype)
        xr = torch.stack([torch.ops.rotation.rotate(x, pk.pairs[p], pk.theta[p], pk.channel_scales.reshape(len(sizes), 1, K)[p])
                          for p in range(len(sizes))])
        y2 = ops.w4a16_gemv_tuned(xr, pk, 0, 0, 0, 2, bias)
        y1 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 1, bias)
        assert y2.shape == (rows, sum(sizes)) and y1.shape == y2.shape
        assert (y2.float() - y1.float()).abs().max().item() <= (4e-3 if dtype == torch.float16 else 3e-2) * y1.float().abs().max().item()
        for knobs in ((2, 1, 8), (4, 2, 4)):
            y0 = ops.w4a16_gemv_tuned(x, pk, *knobs, 0, bias)
            assert torch.equal(ops.w4a16_gemv_tuned(x, pk, *knobs, 1, bias), y0), (rows, knobs)
        ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                      L["channel_scales"], sizes, _np(bias), ideal=True)
        assert po.rel_err(_np(y2), ideal) < (TIGHT_F16 if dtype == torch.float16 else TIGHT_BF16)
    with pytest.raises(ValueError, match="pre-rotated"):
        ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 2, bias)              # [rows, K] is not the pre-rotated layout


@pytest.mark.parametrize("tpw", [1, 2, 4, 8])
@pytest.mark.parametrize("ksplit,waves,mode", [(1, 4, 0), (2, 4, 0), (3, 8, 0), (0, 0, 0), (1, 16, 0), (2, 16, 0),
                                               (1, 0, 1), (2, 4, 1)])
def test_gemv_launch_shapes_agree(dev, tpw, ksplit, waves, mode):
    """Every tiles-per-wave / K-split / waves-per-workgroup combination and the unfused (rotate pre-pass)
    route give the same answer (split-K combine included)."""
    from paroquant_amd import ops
    K, sizes = 1536, [400, 112]      # 12 groups; 25 + 7 tiles -> ragged column blocks for every tpw
    L = po.make_layer(99, K, sizes, bias=True)
    rng = np.random.default_rng(5)
    pk = _packed(L, dev, L["bias"])
    for rows in (1, 4, 6, 13):
        if (rows > 4 and waves == 16) or (rows > 8 and tpw == 8) or (tpw == 8 and waves == 16):
            continue   # combinations that are not built (see launch tables in gemv_impl.hpp)
        x = rng.standard_normal((rows, K)).astype(np.float16)
        y = ops.w4a16_gemv_tuned(_t(x, dev), pk, tpw, ksplit, waves, mode, pk.bias)
        ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                      L["channel_scales"], sizes, L["bias"], ideal=True)
        assert po.rel_err(_np(y), ideal) < TIGHT_F16
    # the K-split leaves nothing to re-arm: the counter area holds per-block epochs (each split launch advanced the words of
    # its column blocks by one), the status word behind them stays zero, and the next launch's tags differ from every granule
    torch.cuda.synchronize()
    words = pk.workspace[:16384].view(torch.int32)
    assert int(words[-1].item()) == 0 and int(words.min().item()) >= 0
    ops.check_workspace(pk.workspace)


def test_gemv_repeated_calls_are_deterministic(dev):
    L = po.make_layer(7, 4096, [4096, 1024, 1024])
    x = _t(np.random.default_rng(0).standard_normal((1, 4096)).astype(np.float16), dev)
    pk = _packed(L, dev)
    y0 = pk.apply(x)
    for _ in range(20):
        assert torch.equal(pk.apply(x), y0)


def test_gemv_bf16(dev):
    K, sizes = 2048, [1024, 512]
    L = po.make_layer(21, K, sizes)
    rng = np.random.default_rng(2)
    for rows in (1, 5):
        x = torch.from_numpy(rng.standard_normal((rows, K)).astype(np.float32)).to(dev).to(torch.bfloat16)
        pk = _packed(L, dev)
        y = pk.apply(x)
        assert y.dtype == torch.bfloat16
        ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                      L["channel_scales"], sizes, None, ideal=True)
        assert po.rel_err(_np(y), ideal) < TIGHT_BF16


@pytest.mark.parametrize("krot", [1, 3, 8, 12])
def test_gemv_other_krot(dev, krot):
    """krot < 8 pads the packed coefficients with identity stages; krot > 8 takes the unfused route."""
    rng = np.random.default_rng(8)
    K, N = 512, 256
    L = po.make_layer(31, K, [N], krot=krot)
    x = rng.standard_normal((2, K)).astype(np.float16)
    y = _packed(L, dev).apply(_t(x, dev))
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], [N], None, ideal=True)
    assert po.rel_err(_np(y), ideal) < TIGHT_F16


def test_gemv_hip_graph_capture(dev):
    """The fused op is capturable (no sync, no allocation by the library, current stream)."""
    L = po.make_layer(41, 2560, [4096, 1024, 1024])
    pk = _packed(L, dev)
    x = torch.randn(1, 2560, device=dev, dtype=torch.float16)
    y_eager = pk.apply(x).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            pk.apply(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = pk.apply(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_g, y_eager)
    x.copy_(torch.randn_like(x))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_g, pk.apply(x))


# ---------------------------------------------------------------- MFMA GEMM (prefill)

@pytest.mark.parametrize("K,sizes,rows", [
    (512, [256], 17), (1024, [2048, 1024, 1024], 130), (4096, [4096], 256), (2560, [9728, 9728], 64),
    (256, [48, 16], 300), (9728, [2560], 129),
    (256, [8192, 8192], 24), (128, [16384], 32), (2048, [512, 272], 40), (4096, [1024], 64), (1024, [48], 33),   # 17..64 rows: GEMV with 2 / 4 MFMA row tiles
])
def test_gemm_matches_oracle(dev, K, sizes, rows):
    L = po.make_layer(K + rows, K, sizes, bias=True)
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, K)).astype(np.float16)
    y = _packed(L, dev, L["bias"]).apply(_t(x, dev))
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, L["bias"], ideal=True)
    ref = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                L["channel_scales"], sizes, L["bias"], act="f16")
    assert po.rel_err(_np(y), ref) < REL_TOL
    assert po.rel_err(_np(y), ideal) < TIGHT_F16


@pytest.mark.parametrize("wq_order", [0, 1])
def test_gemm_and_gemv_agree(dev, wq_order):
    """Both tile orders of the packed weights ([tile][group] / [group][tile]) feed both kernels correctly."""
    from paroquant_amd import ops
    from paroquant_amd.linear import PackedParoWeights
    L = po.make_layer(77, 1024, [512, 256])
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev),
                           _t(L["pairs"], dev), _t(L["channel_scales"], dev), L["sizes"], wq_order=wq_order)
    W = ops.dequant_packed(pk.wq, pk.sz, 1024, L["sizes"], torch.float16, wq_order).cpu().numpy()
    assert np.array_equal(W.view(np.uint16), po.dequant_awq(L["qweight"], L["qzeros"], L["scales"]).view(np.uint16))
    x = torch.randn(16, 1024, device=dev, dtype=torch.float16)
    y1 = pk.apply(x)
    y2 = ops.w4a16_gemm_forced(x, pk)
    assert (y1.float() - y2.float()).abs().max().item() <= 2e-3 * y1.float().abs().max().item()


@pytest.mark.parametrize("rows", [17, 32, 48, 64])
def test_skinny_rows_bf16(dev, rows):
    """17..64 rows with bf16 activations take the pre-rotated GEMV with 2 / 4 MFMA row tiles (as fp16 does)."""
    K, sizes = 2048, [1024, 256]
    L = po.make_layer(rows, K, sizes)
    x = torch.randn(rows, K, device=dev).to(torch.bfloat16)
    y = _packed(L, dev).apply(x)
    assert y.dtype == torch.bfloat16
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, None, ideal=True)
    assert po.rel_err(_np(y), ideal) < TIGHT_BF16


def test_gemm_bf16(dev):
    K, sizes, rows = 1024, [512], 200
    L = po.make_layer(55, K, sizes)
    x = torch.randn(rows, K, device=dev).to(torch.bfloat16)
    y = _packed(L, dev).apply(x)
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, None, ideal=True)
    assert po.rel_err(_np(y), ideal) < TIGHT_BF16


# ---------------------------------------------------------------- the north star's shapes against the oracle, directly

@pytest.mark.parametrize("K,sizes", [(4096, [4096]), (4096, [1024]), (4096, [4096, 1024, 1024]), (4096, [14336, 14336]), (14336, [4096])])
def test_north_star_shapes_match_c_oracle(dev, K, sizes):
    """BASELINE.json: "batch-1 INT4 GEMV at Llama-3-8B q/k/v/o/mlp shapes ... numerics within 1e-2 of CPU reference": every row of
    BASELINE.md section 3 at one row (and three) against oracle/paro_cpu.c -- the C restatement of the reference algorithm
    (rotation.cuh:91-173 half path, per-stage rounding; (q - z) s dequant; fp32 accumulation), itself checked against the numpy oracle
    in tests/test_oracle_c.py -- through the per-call operator and, for the K-split shapes, the deferred-reduction route."""
    from oracle import paro_cpu as pc
    from paroquant_amd import ops
    L = po.make_layer(K + sum(sizes), K, sizes)
    pk = _packed(L, dev)
    rng = np.random.default_rng(K ^ sum(sizes))
    for rows in (1, 3):
        x = rng.standard_normal((rows, K)).astype(np.float16)
        ref = pc.linear_f16(x, dict(L, sizes=sizes)).astype(np.float32)
        got = _np(pk.apply(_t(x, dev)))
        assert got.shape == ref.shape and np.isfinite(got).all()
        assert po.rel_err(got, ref) < REL_TOL, (rows, po.rel_err(got, ref))
        assert np.allclose(got, ref, rtol=1e-2, atol=1e-2 * float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))   # SURVEY 8c's second form
    n = ops.gemv_parts_count(pk)
    if n >= 2:          # the route the decoder harness takes for this shape: partial sums completed by paro_parts_finish
        parts = torch.zeros(sum(sizes), 4, device=dev, dtype=torch.float32)
        ops.w4a16_gemv_fused(_t(x[:1], dev), pk, 0, parts_out=parts, parts_n=n)
        fin = ops.parts_finish(parts, out=torch.empty(sum(sizes), device=dev, dtype=torch.float16))
        assert po.rel_err(_np(fin)[None], pc.linear_f16(x[:1], dict(L, sizes=sizes)).astype(np.float32)) < REL_TOL


# ---------------------------------------------------------------- full-size properties (BASELINE shapes)

@pytest.mark.parametrize("K,sizes", [(4096, [14336, 14336]), (14336, [4096]), (8192, [8192]),
                                     (8192, [28672, 28672]), (28672, [8192]), (8192, [8192, 1024, 1024])])
def test_full_size_consistency_and_linearity(dev, K, sizes):
    """Size-independent properties at Llama-3-8B / 70B-class shapes (the north star's named shapes are ALSO compared with the
    oracle directly: test_north_star_shapes_match_c_oracle below): (i) fused == rotate-op -> dense matmul on the GPU-dequantised
    weights (each piece is separately oracle-checked above); (ii) linearity in x."""
    from paroquant_amd import ops
    rng = np.random.default_rng(K)
    N = sum(sizes)
    G = K // 128
    P = len(sizes)
    qweight = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int64, device=dev).to(torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int64, device=dev).to(torch.int32)
    scales = (torch.rand(G, N, device=dev) * 0.018 + 0.002).half()
    theta = (torch.randn(P, 8, K // 2, device=dev) * 0.1).half()
    pairs = _t(np.stack([po.random_pairs(rng, 8, K) for _ in range(P)]), dev)
    cs = (torch.rand(P, 1, K, device=dev) * 1.5 + 0.5).half()
    from paroquant_amd.linear import PackedParoWeights
    pk = PackedParoWeights(qweight, qzeros, scales, theta, pairs, cs, sizes)
    W = ops.dequant_packed(pk.wq, pk.sz, K, sizes, torch.float16, pk.wq_order).float()
    for rows in (1, 4, 48):
        x = torch.randn(rows, K, device=dev, dtype=torch.float16)
        y = pk.apply(x).float()
        col, parts = 0, []
        for p, n in enumerate(sizes):
            xr = torch.ops.rotation.rotate(x, pairs[p], theta[p], cs[p]).float()
            parts.append(xr @ W[:, col:col + n])
            col += n
        ref = torch.cat(parts, dim=-1)
        assert (y - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()
    x1 = torch.randn(1, K, device=dev, dtype=torch.float16)
    x2 = torch.randn(1, K, device=dev, dtype=torch.float16)
    ya, yb, yab = pk.apply(x1).float(), pk.apply(x2).float(), pk.apply((x1.float() * 0.5 + x2.float() * 0.25).half()).float()
    assert (yab - (0.5 * ya + 0.25 * yb)).abs().max().item() <= 6e-3 * yab.abs().max().item()


# ---------------------------------------------------------------- every GEMM variant, forced through the ABI knob

def _random_gpu_layer(dev, K, sizes, seed):
    """Random layer in checkpoint format generated ON the GPU (full-size shapes), plus its numpy view for the oracle."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    N, G, P = sum(sizes), K // 128, len(sizes)
    L = dict(
        qweight=torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32),
        qzeros=torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32),
        scales=(torch.rand(G, N, device=dev, generator=gen) * 0.018 + 0.002).half(),
        theta=(torch.randn(P, 8, K // 2, device=dev, generator=gen) * 0.1).half(),
        pairs=_t(np.stack([po.random_pairs(np.random.default_rng(seed + p), 8, K) for p in range(P)]), dev),
        channel_scales=(torch.rand(P, 1, K, device=dev, generator=gen) * 1.5 + 0.5).half(), sizes=list(sizes))
    return L


def _pack_gpu_layer(L, bias=None):
    from paroquant_amd.linear import PackedParoWeights
    return PackedParoWeights(L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"],
                             L["sizes"], bias)


def _oracle_rows(L, x_rows, bias=None):
    n = lambda t: t.detach().cpu().numpy()
    return po.paro_linear_merged(_np(x_rows), n(L["qweight"]), n(L["qzeros"]), n(L["scales"]), n(L["theta"]), n(L["pairs"]),
                                 n(L["channel_scales"]), L["sizes"], None if bias is None else _np(bias), ideal=True)


@pytest.mark.parametrize("variant", [1, 2, 4])
@pytest.mark.parametrize("K,sizes,rows", [
    (512, [256], 300),                      # one column block, ragged row tail
    (1024, [272, 48], 700),                 # ragged partitions: 17 + 3 tiles -> partial 256-column blocks
    (384, [512, 256, 256], 256),            # merged qkv-style, exactly one row block, 3 groups
    (256, [48, 16], 33),                    # tiny
])
def test_gemm_variants_forced(dev, variant, K, sizes, rows):
    """Every prefill kernel (ABI knob `variant`) computes the same function at sizes the oracle finishes in seconds."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows + variant, K, sizes, bias=True)
    pk = _packed(L, dev, L["bias"])
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    y = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, variant=variant)
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, L["bias"], ideal=True)
    ref = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                L["channel_scales"], sizes, L["bias"], act="f16")
    assert np.isfinite(_np(y)).all()
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    assert po.rel_err(_np(y), ref) < REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("gs", [128, 64])
@pytest.mark.parametrize("K,sizes,rows", [
    (2048, [512, 272], 33), (2048, [512, 272], 64),       # 64-row blocks, K-split 8 (ragged last column block)
    (1024, [1024, 256, 256], 65), (1024, [1024, 256, 256], 128), (4096, [768], 100),   # 96- / 128-row blocks, K-split
    (1024, [1024, 256, 256], 150), (2048, [512, 272], 192),                            # 160- / 192-row blocks
    (512, [8192, 8192], 48),                              # wide output: no K-split (64 column blocks already)
    (4096, [48], 40),                                     # a single partial column block
])
def test_gemm_row_tile_blocks(dev, gs, K, sizes, rows):
    """33..192 rows run GEMM variant 4 with 64- .. 192-row blocks and (narrow outputs) an fp32 K-split summed by a second
    kernel: the automatic route and the forced variant agree with the oracle, fp16 and bf16, group_size 128 and 64."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows + gs, K, sizes, group_size=gs, bias=True)
    pk = _packed(L, dev, L["bias"])
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes,
                                  L["bias"], group_size=gs, ideal=True)
    y_auto = pk.apply(_t(x, dev))
    y_v4 = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, variant=4)
    assert np.isfinite(_np(y_auto)).all()
    if rows > 64 or sum(sizes) // 16 >= 1024:
        assert torch.equal(y_auto, y_v4)          # the automatic route IS variant 4 here
    else:
        # round 6: up to 64 rows, outputs below 1024 tiles take the 4-row-tile GEMV behind the schedule pre-pass (abi.hip) -- the bits of
        # the in-kernel rotation, not the GEMM's; both against the oracle
        assert torch.equal(y_auto, ops.w4a16_gemv_tuned(_t(x, dev), pk, 0, 0, 0, 1, pk.bias))
        assert po.rel_err(_np(y_v4), ideal) < TIGHT_F16
    assert po.rel_err(_np(y_auto), ideal) < TIGHT_F16
    yb = pk.apply(_t(x, dev).to(torch.bfloat16), pk.bias.to(torch.bfloat16))
    assert po.rel_err(_np(yb), ideal) < 2e-2
    y1 = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, variant=1)              # an independent kernel on the same inputs
    assert po.rel_err(_np(y_auto), _np(y1).astype(np.float64)) < TIGHT_F16


@pytest.mark.parametrize("variant", [1, 4])
@pytest.mark.parametrize("K,sizes,rows", [(1024, [512, 272], 300), (256, [4096], 512)])
def test_gemm_bf16_native(dev, variant, K, sizes, rows):
    """bf16 activations run bf16 MFMAs on bf16-dequantised weights (no fp16 detour): variants 1 and 4."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows, K, sizes, bias=True)
    pk = _packed(L, dev)
    x = torch.randn(rows, K, device=dev).to(torch.bfloat16)
    bias = _t(L["bias"], dev).to(torch.bfloat16)
    y = ops.w4a16_gemm_forced(x, pk, bias, variant=variant)
    assert y.dtype == torch.bfloat16
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, _np(bias), ideal=True)
    assert po.rel_err(_np(y), ideal) < TIGHT_BF16
    # values far outside the fp16 range survive (the round-1 fp16 detour saturated them at 65504)
    xb = x.clone()
    xb[0, :] *= 3.0e4
    yb = ops.w4a16_gemm_forced(xb, pk, bias, variant=variant)
    assert torch.isfinite(yb.float()).all()
    ideal_b = po.paro_linear_merged(_np(xb[:1]), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                    L["channel_scales"], sizes, _np(bias), ideal=True)
    assert po.rel_err(_np(yb[:1]), ideal_b) < TIGHT_BF16


BASELINE_PREFILL = [
    ("qwen3-4b.gate_up", 2560, [9728, 9728]), ("qwen3-4b.down", 9728, [2560]),
    ("llama3-8b.o", 4096, [4096]), ("llama3-8b.qkv", 4096, [4096, 1024, 1024]),
    # Qwen3.5 (transformers default "9B style" text config): gated-delta-net in_proj_qkv | in_proj_z, gated full-attention q | k | v
    ("qwen3.5.in_proj_qkvz", 4096, [8192, 4096]), ("qwen3.5.qkv_gated", 4096, [8192, 1024, 1024]),
]


@pytest.mark.parametrize("rows", [65536, 3333])
@pytest.mark.parametrize("name,K,sizes", BASELINE_PREFILL)
def test_prefill_gemm_at_baseline_sizes(dev, name, K, sizes, rows):
    """BASELINE config 3 (batch 32 x seq 2048 = 65536 rows) and a ragged M, on the 256 x 256 prefill kernels
    (variant 4 = what auto picks):
      (i)  64 sampled rows x all N columns against the float64 oracle;
      (ii) the FULL tensor against rotation::rotate -> fp32 matmul on the GPU-dequantised weights
           (each piece is separately oracle-checked), in row chunks."""
    from paroquant_amd import ops
    L = _random_gpu_layer(dev, K, sizes, seed=K + rows)
    pk = _pack_gpu_layer(L).prepare_prefill(torch.float16)
    N = sum(sizes)
    gen = torch.Generator(device=dev)
    gen.manual_seed(rows)
    x = torch.randn(rows, K, device=dev, dtype=torch.float16, generator=gen)
    W = ops.dequant_packed(pk.wq, pk.sz, K, sizes, torch.float16, pk.wq_order).float()
    sample = torch.randperm(rows, device=dev, generator=gen)[:64].sort().values
    ideal = _oracle_rows(L, x[sample])
    assert pk.apply(x[:16]).shape == (16, N)
    for variant in (4,):
        y = ops.w4a16_gemm_forced(x, pk, variant=variant)
        assert y.shape == (rows, N)
        assert po.rel_err(_np(y[sample]), ideal) < TIGHT_F16, (name, variant)
        worst = 0.0
        for r0 in range(0, rows, 8192):
            xs = x[r0:r0 + 8192]
            col, parts = 0, []
            for p, n in enumerate(sizes):
                xr = torch.ops.rotation.rotate(xs, L["pairs"][p], L["theta"][p], L["channel_scales"][p]).float()
                parts.append(xr @ W[:, col:col + n])
                col += n
            ref = torch.cat(parts, dim=-1)
            worst = max(worst, ((y[r0:r0 + 8192].float() - ref).abs().max() / ref.abs().max()).item())
            del ref, parts
        assert worst <= 4e-3, (name, variant, worst)
        if variant == 4:
            y_auto = pk.apply(x)                      # the dispatcher's own choice at this size
            assert torch.equal(y_auto, y) or po.rel_err(_np(y_auto[sample]), ideal) < TIGHT_F16
            del y_auto
        del y
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- tensor parallelism, emulated on one GPU

@pytest.mark.parametrize("world", [2, 4])
def test_tp_sharded_hip_path_single_gpu(dev, world):
    """BASELINE config 5 (70B-class, TP = 2 / 4): every rank's shard goes through the HIP path
    (PackedParoWeights.apply on the sharded checkpoint tensors, rotation params narrowed exactly as
    vllm/plugin.py:33-50 does), the partial results are summed (row-parallel: what the RCCL all-reduce does) or
    concatenated (column-parallel) on the device, and compared with the unsharded HIP result and the oracle."""
    from paroquant_amd import tp
    from paroquant_amd.linear import PackedParoWeights
    cases = [("o_proj", 8192, [8192], "row"), ("qkv_proj", 8192, [8192, 1024, 1024], "col"),
             ("down_proj", 28672, [8192], "row"), ("gate_up_proj", 8192, [28672, 28672], "col")]
    for name, K, sizes, kind in cases:
        L = _random_gpu_layer(dev, K, sizes, seed=world * 1000 + K)
        full = _pack_gpu_layer(L)
        for rows in (1, 5):
            x = torch.randn(rows, K, device=dev, dtype=torch.float16)
            y_full = full.apply(x)
            if kind == "row":
                acc = torch.zeros(rows, sizes[0], device=dev, dtype=torch.float32)
                for r in range(world):
                    sh = tp.shard_row_parallel(L, r, world)
                    pk = PackedParoWeights(sh["qweight"], sh["qzeros"], sh["scales"], sh["theta"], sh["pairs"],
                                           sh["channel_scales"], sizes)
                    Kp = K // world
                    acc += pk.apply(x[:, r * Kp:(r + 1) * Kp].contiguous()).float()
                y_tp = acc
            else:
                outs = []
                for r in range(world):
                    sh = tp.shard_column_parallel(L, sizes, r, world)
                    pk = PackedParoWeights(sh["qweight"], sh["qzeros"], sh["scales"], sh["theta"], sh["pairs"],
                                           sh["channel_scales"], sh["sizes"])
                    outs.append(pk.apply(x).float().split(sh["sizes"], dim=-1))
                # rank-major per partition -> the unsharded column order
                y_tp = torch.cat([torch.cat([outs[r][p] for r in range(world)], dim=-1) for p in range(len(sizes))], dim=-1)
            ref = y_full.float()
            # row shards sum fp16-rounded partials; column shards compute the same dot products, but the narrower
            # shard resolves to a different launch shape (K-split / wave count), i.e. another fp32 summation order
            tol = 4e-3 if kind == "row" else 1e-3
            assert ((y_tp - ref).abs().max() / ref.abs().max()).item() <= tol, (name, world, rows)
            if name in ("o_proj", "qkv_proj") and rows == 1:
                ideal = _oracle_rows(L, x)
                assert po.rel_err(_np(y_tp), ideal) < TIGHT_F16
                assert po.rel_err(_np(y_full), ideal) < TIGHT_F16
        del full, L
        torch.cuda.empty_cache()


# ---------------------------------------------------------------- workspace contract (zero-filled, never uninitialised)

def test_workspace_never_uninitialised(dev):
    """17..64 rows take the K-split GEMV on {tag, partial} granules that must start at zero.  A layer whose shared
    decode workspace is too small for that must NOT fall back to uninitialised allocator memory: poison the
    caching allocator with 0x00000001 words (read as "tag = 1" by a reducer) and check the results."""
    from paroquant_amd import ops
    ops._workspaces.clear()
    L = po.make_layer(4242, 4096, [1024])             # narrow N, deep K: auto K-split at 17..64 rows
    pk = _packed(L, dev)
    x = np.random.default_rng(1).standard_normal((48, 4096)).astype(np.float16)
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], [1024], None, ideal=True)
    for _ in range(3):
        poison = torch.full((64 << 20,), 0x00000001, dtype=torch.int32, device=dev)
        del poison                                      # back to the caching allocator, contents intact
        y = pk.apply(_t(x, dev))
        assert po.rel_err(_np(y), ideal) < TIGHT_F16
    ops.check_workspace(ops.get_workspace(dev, 1))
    ops.check_workspace(pk.workspace)


def test_workspace_status_reports_poison(dev):
    """paro_workspace_status: healthy after K-split launches; a non-zero status word is reported."""
    from paroquant_amd import ops, _native
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    ops.check_workspace(ws)
    ws[_native.PARO_WS_STATUS_OFFSET:_native.PARO_WS_STATUS_OFFSET + 4] = torch.tensor([0xAD, 0xDE, 0, 0], dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="gave up"):
        ops.check_workspace(ws)


def test_ksplit_grid_must_be_resident(dev):
    """A K-split whose grid cannot be resident at once is refused on the host (the reducers spin on partials
    published by other workgroups of the same launch: forward progress must not depend on dispatch order)."""
    from paroquant_amd import ops
    L = po.make_layer(9, 2048, [16384 * 4])            # 4096 column tiles
    pk = _packed(L, dev)
    x = torch.randn(1, 2048, device=dev, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="resident"):
        ops.w4a16_gemv_tuned(x, pk, 2, 16, 4, 0)       # 2048 x 16 workgroups of 4 waves
    y = ops.w4a16_gemv_tuned(x, pk, 1, 1, 4, 0)
    assert torch.isfinite(y.float()).all()


# ---------------------------------------------------------------- operator API / plug-in surface

def test_rotate_quantized_linear_module(dev, golden_dir):
    """HF operator: state-dict load of a reference-exported layer (golden G6) + forward vs oracle."""
    from paroquant_amd import RotateQuantizedLinear
    g = np.load(os.path.join(golden_dir, "quantize_layer.npz"))
    K = g["weight"].shape[1]
    N = g["weight"].shape[0]
    m = RotateQuantizedLinear(K, N, bias=True)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("out_")}
    m.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, K, dtype=torch.float16))            # no CPU path, like the reference
    m = m.to(dev)
    x = torch.randn(2, 3, K, device=dev, dtype=torch.float16)
    y = m(x)
    assert y.shape == (2, 3, N)
    ref = po.paro_linear(_np(x), g["out_qweight"], g["out_qzeros"], g["out_scales"], g["out_theta"], g["out_pairs"],
                         g["out_channel_scales"], g["out_bias"], ideal=True)
    assert po.rel_err(_np(y), ref) < TIGHT_F16
    with pytest.raises(AssertionError, match="float16"):
        m(x.float())


@pytest.mark.parametrize("gs", [128, 64])
def test_vllm_linear_method_contract(dev, gs):
    """ParoQuantLinearMethod: create_weights -> shard-id loaders -> process_weights_after_loading -> apply,
    against the per-partition rotate + matmul + cat of the reference (plugin.py:281-311); group_size from the config."""
    from paroquant_amd.vllm_plugin import ParoQuantConfig, ParoQuantLinearMethod
    K, sizes = 1024, [512, 128, 128]
    L = po.make_layer(123, K, sizes, group_size=gs, bias=True)
    cfg = ParoQuantConfig.from_config({"bits": 4, "group_size": gs, "krot": 8})
    method = ParoQuantLinearMethod(cfg)
    layer = torch.nn.Module()
    method.create_weights(layer, K, sizes, K, sum(sizes), torch.float16)
    assert layer.theta.shape == (3, 8, K // 2) and layer.pairs.dtype == torch.int16 and layer.qzeros.shape[0] == K // gs
    layer.qweight.data.copy_(torch.from_numpy(L["qweight"]))
    layer.qzeros.data.copy_(torch.from_numpy(L["qzeros"]))
    layer.scales.data.copy_(torch.from_numpy(L["scales"]))
    for i, sid in enumerate(["q", "k", "v"]):
        layer.theta.weight_loader(layer.theta, torch.from_numpy(L["theta"][i]), sid)
        layer.pairs.weight_loader(layer.pairs, torch.from_numpy(L["pairs"][i]), sid)
        layer.channel_scales.weight_loader(layer.channel_scales, torch.from_numpy(L["channel_scales"][i]), sid)
    layer.to(dev)
    method.process_weights_after_loading(layer)
    assert not hasattr(layer, "qweight") and hasattr(layer, "rot_theta")
    x = torch.randn(5, K, device=dev, dtype=torch.float16)
    bias = _t(L["bias"], dev)
    y = method.apply(layer, x, bias)
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, L["bias"], group_size=gs, ideal=True)
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    xb = torch.randn(70, K, device=dev).to(torch.bfloat16)               # vLLM's bf16 activations, a prefill-sized batch
    yb = method.apply(layer, xb, bias.to(torch.bfloat16))
    ideal_b = po.paro_linear_merged(xb.float().cpu().numpy(), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                    L["channel_scales"], sizes, L["bias"], group_size=gs, ideal=True)
    assert yb.dtype == torch.bfloat16 and po.rel_err(yb.float().cpu().numpy(), ideal_b) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_prefill_prepass_variants_agree(dev, dtype):
    """Prefill pre-pass: dense per-group rotation on the matrix cores == the stage kernel (both vs oracle)."""
    from paroquant_amd import ops
    K, sizes, rows = 1024, [256, 128], 700          # 2 rotations, ragged row tail (700 = 2*256 + 188)
    L = po.make_layer(808, K, sizes)
    pk = _packed(L, dev)
    x = torch.randn(rows, K, device=dev).to(dtype)
    y_m = ops.w4a16_gemm_forced(x, pk, use_rmat=True)
    y_s = ops.w4a16_gemm_forced(x, pk, use_rmat=False)
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, None, ideal=True)
    tol = TIGHT_F16 if dtype == torch.float16 else TIGHT_BF16
    assert po.rel_err(_np(y_m), ideal) < tol and po.rel_err(_np(y_s), ideal) < tol
    assert po.rel_err(_np(y_m), _np(y_s)) < tol
    # the dense matrices are orthogonal up to the channel scales: R'_g^T R'_g = diag(cs^2)
    R = pk.rotation_matrices(torch.float32)          # [P, G, n, k]
    cs = pk.channel_scales.float().view(len(sizes), K // 128, 128)
    gram = torch.einsum("pgnk,pgnl->pgkl", R, R)
    eye = torch.diag_embed(cs * cs)
    assert (gram - eye).abs().max().item() < 5e-3


# ---------------------------------------------------------------- f1: the packer library in the product

def test_pack_library_gpu_matches_goldens(dev, golden_dir):
    """paroquant_amd.pack on GPU tensors (rotation through rotation::rotate in fp32): bit-exact against the
    reference-generated goldens G5 (_quantize_rotated_weight) and G6 (_quantize_layer)."""
    from paroquant_amd import pack
    g = np.load(os.path.join(golden_dir, "quantize_rotated.npz"))
    q, s2d, z2d = pack.quantize_rotated_weight(_t(g["weight"], dev), _t(g["pairs"], dev), _t(g["theta"], dev),
                                               _t(g["channel_scales"], dev), _t(g["scales_flat"], dev), _t(g["zp_flat"], dev))
    # the fp32 rotation on the GPU (v_sin / v_cos) and the oracle's libm rotation may round a weight that sits on
    # a quantisation boundary to the neighbouring level: allow |dq| <= 1 on < 0.1 % of the weights, exact elsewhere
    dq = (q.cpu().numpy().astype(np.int64) - g["quantized"].astype(np.int64))
    assert np.abs(dq).max() <= 1 and (dq != 0).mean() < 1e-3
    assert np.array_equal(z2d.cpu().numpy(), g["zeros_2d"].astype(np.int32))
    assert np.allclose(s2d.cpu().numpy(), g["scales_2d"], rtol=0, atol=0)
    g6 = np.load(os.path.join(golden_dir, "quantize_layer.npz"))
    sd = {"weight": torch.from_numpy(g6["weight"]), "n_bits": torch.tensor(int(g6["bits"])), "group_size": torch.tensor(int(g6["group_size"])),
          "pairs_grouped": torch.from_numpy(g6["pairs_in"]), "angles_grouped": torch.from_numpy(g6["theta_in"]),
          "channel_scales": torch.from_numpy(g6["channel_scales_opt"]), "quantizer.scale": torch.from_numpy(g6["scale"]),
          "quantizer.zero_point_float": torch.from_numpy(g6["zero_point_float"]), "bias": torch.from_numpy(g6["bias_in"])}
    out = pack.quantize_layer(sd, dev)
    for k in ("qzeros", "scales", "theta", "pairs", "channel_scales", "bias"):
        a, b = out[k].cpu().numpy(), g6["out_" + k]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), k
    dq = pack.unpack_awq(out["qweight"]).cpu().numpy().astype(np.int64) - po.unpack_awq(g6["out_qweight"]).astype(np.int64)
    assert np.abs(dq).max() <= 1 and (dq != 0).mean() < 1e-3


@pytest.mark.parametrize("gs", [128, 64])
def test_prepacked_roundtrip(dev, tmp_path, gs):
    """save_prepacked / load_prepacked: a linear stored in the CDNA4 kernel layout comes back bit-identical and
    computes the same outputs without any repack launch (the quantisation group is read off the packed tensors)."""
    from paroquant_amd import pack
    L = po.make_layer(606, 1024, [512, 256, 256], group_size=gs, bias=True)
    pk = _packed(L, dev, L["bias"])
    f = str(tmp_path / "qkv.prepacked.safetensors")
    pack.save_prepacked(pk, f)
    pk2 = pack.load_prepacked(f, dev)
    for name in ("wq", "sz", "rot"):
        assert torch.equal(getattr(pk, name), getattr(pk2, name))
    assert pk2.partition_sizes == [512, 256, 256] and pk2.K == 1024 and pk2.wq_order == pk.wq_order and pk2.group_size == gs
    for rows in (1, 40, 300):
        x = torch.randn(rows, 1024, device=dev, dtype=torch.float16)
        assert torch.equal(pk.apply(x), pk2.apply(x))


# ---------------------------------------------------------------- a8: HF from_pretrained end to end

@pytest.mark.parametrize("gs", [128, 64])
def test_hf_from_pretrained_end_to_end(dev, tmp_path, gs):
    """A synthetic 2-layer Llama-style PARO checkpoint (safetensors + config.json with quantization_config, tensors
    from the oracle's packer) loads through AutoModelForCausalLM.from_pretrained -> ParoQuantHfQuantizer
    (transformers/quantizer.py:88-115): every quantised nn.Linear is swapped for RotateQuantizedLinear, repacked,
    and one forward of each swapped linear matches the oracle on the activations it actually received."""
    import paroquant_amd.hf_quantizer  # noqa: F401  (registers the "paroquant" config + quantizer)
    from paroquant_amd import RotateQuantizedLinear
    from tests.hf_ckpt import write_tiny_paro_llama
    from transformers import AutoModelForCausalLM
    layers = write_tiny_paro_llama(str(tmp_path), group_size=gs)
    try:
        model = AutoModelForCausalLM.from_pretrained(str(tmp_path), dtype=torch.float16, device_map={"": "cuda:0"})
    except TypeError:
        model = AutoModelForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.float16, device_map={"": "cuda:0"})
    swapped = {k: m for k, m in model.named_modules() if isinstance(m, RotateQuantizedLinear)}
    assert set(swapped) == set(layers)
    assert all(m._packed is not None for m in swapped.values())          # repacked by the after-loading hook
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: seen.__setitem__(k, (inp[0].detach(), out.detach())))
             for k, m in swapped.items()]
    ids = torch.randint(0, 128, (1, 9), device=dev)
    with torch.no_grad():
        logits = model(input_ids=ids).logits
    for h in hooks:
        h.remove()
    assert logits.shape == (1, 9, 128) and torch.isfinite(logits.float()).all()
    assert set(seen) == set(layers)
    for k, (x, y) in seen.items():
        L = layers[k]
        K = x.shape[-1]
        ref = po.paro_linear(_np(x.reshape(-1, K)), L["qweight"], L["qzeros"], L["scales"], L["theta"][0], L["pairs"][0],
                             L["channel_scales"][0], None, gs, ideal=True)
        assert po.rel_err(_np(y.reshape(-1, y.shape[-1])), ref) < TIGHT_F16, k
    assert all(m.group_size == gs for m in swapped.values())
    # greedy decode runs (prefill rows > 1, then single-token steps through the GEMV)
    with torch.no_grad():
        gen = model.generate(ids, max_new_tokens=4, do_sample=False)
    assert gen.shape == (1, 13)


# ---------------------------------------------------------------- seeded random shapes through the automatic routes

def _fuzz_cases(n=120):
    rng = np.random.default_rng(20260927)
    row_choices = [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 24, 32, 33, 47, 64, 65, 100, 128, 129, 191, 192, 193, 255, 256, 257, 300, 513]
    out = []
    for i in range(n):
        K = int(rng.integers(1, 17)) * 128
        P = int(rng.integers(1, 5))
        sizes = [int(rng.integers(1, 49)) * 16 for _ in range(P)]
        if i % 6 == 0:
            sizes[0] = int(rng.integers(60, 130)) * 16          # some wide partitions (>= 1024 columns: other heuristics)
        rows = int(row_choices[int(rng.integers(0, len(row_choices)))])
        out.append((i, K, tuple(sizes), rows, 64 if i % 3 == 1 else 128, bool(i % 2), bool(i % 5 == 0)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case,K,sizes,rows,gs,use_bf16,with_bias", _fuzz_cases())
def test_random_shapes_automatic_routes(dev, case, K, sizes, rows, gs, use_bf16, with_bias):
    """Seeded random (K, merged partition sizes, rows, group_size, dtype, bias) through `PackedParoWeights.apply`, i.e.
    through whatever kernel the dispatchers pick (fused GEMV, pre-rotated GEMV, every GEMM variant and K-split): the
    row counts sit on every dispatch boundary (4/5, 8/9, 16/17, 32/33, 64/65, 128/129, 192/193, 255/256/257)."""
    sizes = list(sizes)
    L = po.make_layer(1000 + case, K, sizes, group_size=gs, bias=with_bias)
    pk = _packed(L, dev, L.get("bias"))
    x = np.random.default_rng(case).standard_normal((rows, K)).astype(np.float16)
    xt = _t(x, dev).to(torch.bfloat16) if use_bf16 else _t(x, dev)
    y = pk.apply(xt)
    ideal = po.paro_linear_merged(xt.float().cpu().numpy(), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, L.get("bias"), group_size=gs, ideal=True)
    got = y.float().cpu().numpy()
    assert got.shape == (rows, sum(sizes)) and np.isfinite(got).all()
    assert po.rel_err(got, ideal) < (2e-2 if use_bf16 else TIGHT_F16)


def _fused_fuzz_cases(n=60):
    rng = np.random.default_rng(927)
    out = []
    for i in range(n):
        K = int(rng.integers(1, 25)) * 128
        P = int(rng.integers(1, 4))
        sizes = [int(rng.integers(1, 40)) * 16 for _ in range(P)]
        if i % 5 == 0:
            sizes[0] = int(rng.integers(64, 160)) * 16
        out.append((i, K, tuple(sizes), int(rng.integers(1, 5)), i % 3, bool(rng.integers(0, 2)), 64 if i % 4 == 1 else 128, bool(i % 2)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case,K,sizes,rows,prologue,with_res,gs,use_bf16", _fused_fuzz_cases())
def test_random_shapes_fused_gemv(dev, case, K, sizes, rows, prologue, with_res, gs, use_bf16):
    """Seeded random fused decode linears: prologue none / RMSNorm / SiLU*mul x residual x rows 1..4 x group_size x dtype."""
    from paroquant_amd import ops, _native as nat
    sizes = list(sizes)
    N = sum(sizes)
    L = po.make_layer(3000 + case, K, sizes, group_size=gs)
    rng = np.random.default_rng(case)
    dt = torch.bfloat16 if use_bf16 else torch.float16
    pk = _packed(L, dev)
    res = _t(rng.standard_normal((rows, N)).astype(np.float32), dev).to(dt) if with_res else None
    if prologue == 2:      # SiLU * mul: x = [rows, 2 K] (gate then up)
        xin = _t(rng.standard_normal((rows, 2 * K)).astype(np.float32), dev).to(dt)
        xeff = po.silu_mul(xin.float().cpu().numpy(), K)
        y = ops.w4a16_gemv_fused(xin, pk, nat.PROLOGUE_SILU_MUL, residual=res)
    elif prologue == 1:    # RMSNorm: weight folded into the channel scales
        w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
        pk = pk.fold_norm_weight(_t(w, dev))
        xin = _t((rng.standard_normal((rows, K)) * 2.0).astype(np.float32), dev).to(dt)
        xeff = po.rmsnorm(xin.float().cpu().numpy(), w, 1e-6)
        y = ops.w4a16_gemv_fused(xin, pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=res)
    else:
        xin = _t(rng.standard_normal((rows, K)).astype(np.float32), dev).to(dt)
        xeff = xin.float().cpu().numpy()
        if res is None:
            res = _t(rng.standard_normal((rows, N)).astype(np.float32), dev).to(dt)     # prologue none needs the residual to be "fused"
        y = ops.w4a16_gemv_fused(xin, pk, 0, residual=res)
    ideal = po.paro_linear_merged(xeff, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes,
                                  None, group_size=gs, ideal=True)
    if res is not None:
        ideal = ideal + res.float().cpu().numpy().astype(np.float64)
    got = y.float().cpu().numpy()
    assert y.dtype == dt and np.isfinite(got).all()
    assert po.rel_err(got, ideal) < (2e-2 if use_bf16 else TIGHT_F16)


# ---------------------------------------------------------------- quantisation group_size 64 (rotation group stays 128)

GS64_SHAPES = [(256, [48, 16]), (1024, [3072, 3072]), (2560, [4096, 1024, 1024]), (4096, [2560])]


def _ideal(L, x, sizes, gs, bias=None):
    return po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes,
                                 bias, group_size=gs, ideal=True)


@pytest.mark.gpu
@pytest.mark.parametrize("K,sizes", [(128, [16]), (512, [208, 48, 16]), (4096, [1024])])
def test_repack_dequant_bit_exact_group64(dev, K, sizes):
    """group_size 64 (reference: the AWQ matmul gets group_size, the rotation does not -- transformers/modules.py:59-69):
    twice the scale/zero rows, same INT4 tiles; (q - z) * s bit-exact against the oracle."""
    from paroquant_amd import ops
    L = po.make_layer(K + 64, K, sizes, group_size=64)
    assert L["qzeros"].shape[0] == K // 64
    wq, sz = torch.ops.paro.repack_awq(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), sizes)
    tsz = sum((s // 16 + 7) // 8 * 8 for s in sizes)
    assert sz.numel() == (K // 64) * tsz * 16
    w = ops.dequant_packed(wq, sz, K, sizes, torch.float16).cpu().numpy()
    ref = po.dequant_awq(L["qweight"], L["qzeros"], L["scales"], 64, np.float16)
    assert np.array_equal(w.view(np.uint16), ref.view(np.uint16))
    wq128, _ = torch.ops.paro.repack_awq(_t(L["qweight"], dev), _t(L["qzeros"][::2].copy(), dev), _t(L["scales"][::2].copy(), dev), sizes)
    assert torch.equal(wq, wq128)   # the INT4 tiles do not depend on the quantisation group


@pytest.mark.gpu
@pytest.mark.parametrize("K,sizes", GS64_SHAPES)
@pytest.mark.parametrize("rows", [1, 3, 8, 16, 40])
def test_gemv_group64_matches_oracle(dev, K, sizes, rows):
    """Decode / small-batch path at group_size 64 (rows 40: the pre-rotated skinny path), f16 and bf16."""
    L = po.make_layer(K + rows + 64, K, sizes, group_size=64, bias=(rows == 3))
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    pk = _packed(L, dev, L.get("bias"))
    assert pk.group_size == 64
    y = pk.apply(_t(x, dev))
    ideal = _ideal(L, x, sizes, 64, L.get("bias"))
    assert np.isfinite(_np(y)).all()
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    yb = pk.apply(_t(x, dev).to(torch.bfloat16))
    assert po.rel_err(_np(yb), ideal) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("tpw,ksplit,waves,mode", [(1, 1, 8, 0), (2, 2, 4, 0), (4, 4, 8, 0), (8, 1, 8, 0), (0, 0, 0, 0),
                                                   (1, 1, 16, 0), (2, 1, 0, 1), (4, 2, 4, 1)])
def test_gemv_group64_launch_shapes(dev, tpw, ksplit, waves, mode):
    """Launch-shape knobs at group_size 64: unsupported combinations (16 waves) are resolved to a
    built one, never to a wrong answer."""
    from paroquant_amd import ops
    K, sizes = 1536, [400, 112]
    L = po.make_layer(164, K, sizes, group_size=64, bias=True)
    pk = _packed(L, dev, L["bias"])
    rng = np.random.default_rng(7)
    for rows in (1, 4, 7, 13):
        if rows > 8 and tpw == 8:
            continue
        x = rng.standard_normal((rows, K)).astype(np.float16)
        y = ops.w4a16_gemv_tuned(_t(x, dev), pk, tpw, ksplit, waves, mode, pk.bias)
        assert po.rel_err(_np(y), _ideal(L, x, sizes, 64, L["bias"])) < TIGHT_F16


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 4])
@pytest.mark.parametrize("K,sizes,rows", [(512, [256], 300), (1024, [272, 48], 700), (384, [512, 256, 256], 256), (256, [4096], 512)])
def test_gemm_group64(dev, variant, K, sizes, rows):
    """Prefill kernels at group_size 64: variant 1 (128 x 128) and variant 4 (256 x 256, k-steps 0..3 / 4..7 of a slab
    dequantised with different scale / zero words); variant 2 refuses."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows + 64, K, sizes, group_size=64, bias=True)
    pk = _packed(L, dev, L["bias"])
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    y = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, variant=variant) if variant else pk.apply(_t(x, dev))
    ideal = _ideal(L, x, sizes, 64, L["bias"])
    assert np.isfinite(_np(y)).all()
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    if variant == 4:
        yb = ops.w4a16_gemm_forced(_t(x, dev).to(torch.bfloat16), pk, pk.bias.to(torch.bfloat16), variant=4)
        assert po.rel_err(_np(yb), ideal) < 2e-2
        with pytest.raises(RuntimeError, match="group_size 64"):
            ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, variant=2)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 4])
def test_fused_prologues_group64(dev, rows):
    """RMSNorm / SiLU*mul prologues and the residual epilogue at group_size 64."""
    from paroquant_amd import ops, _native as nat
    K, sizes = 1024, [3072, 3072]
    L = po.make_layer(K + rows, K, sizes, group_size=64)
    rng = np.random.default_rng(rows)
    w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
    x = (rng.standard_normal((rows, K)) * 3.0).astype(np.float16)
    res = rng.standard_normal((rows, sum(sizes))).astype(np.float16)
    pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
    y = ops.w4a16_gemv_fused(_t(x, dev), pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=_t(res, dev))
    assert po.rel_err(_np(y), _ideal(L, po.rmsnorm(x, w, 1e-6), sizes, 64) + res.astype(np.float64)) < TIGHT_F16
    Ld = po.make_layer(K + rows + 1, 3072, [1024], group_size=64)
    gu = rng.standard_normal((rows, 2 * 3072)).astype(np.float16)
    yd = ops.w4a16_gemv_fused(_t(gu, dev), _packed(Ld, dev), nat.PROLOGUE_SILU_MUL)
    assert po.rel_err(_np(yd), _ideal(Ld, po.silu_mul(gu, 3072), [1024], 64)) < TIGHT_F16


@pytest.mark.gpu
def test_rotate_quantized_linear_module_group64(dev):
    """The HF-style module with group_size=64 buffers (n_groups = K / 64, transformers/modules.py:40): state-dict load,
    decode and prefill rows."""
    from paroquant_amd import RotateQuantizedLinear
    K, N = 512, 384
    L = po.make_layer(64, K, [N], group_size=64, bias=True)
    m = RotateQuantizedLinear(K, N, bias=True, group_size=64, bits=4, krot=8)
    assert tuple(m.qzeros.shape) == (K // 64, N // 8) and tuple(m.scales.shape) == (K // 64, N)
    m.load_state_dict({"theta": torch.from_numpy(L["theta"][0]), "pairs": torch.from_numpy(L["pairs"][0]),
                       "channel_scales": torch.from_numpy(L["channel_scales"][0].reshape(1, K)), "qweight": torch.from_numpy(L["qweight"]),
                       "qzeros": torch.from_numpy(L["qzeros"]), "scales": torch.from_numpy(L["scales"]), "bias": torch.from_numpy(L["bias"])})
    m = m.to(dev)
    for rows in (1, 5, 300):
        x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
        y = m(_t(x, dev))
        assert po.rel_err(_np(y), _ideal(L, x, [N], 64, L["bias"])) < TIGHT_F16


# ---------------------------------------------------------------- f3: fused prologue / epilogue of the decode GEMV

@pytest.mark.parametrize("K,sizes", [(2560, [4096, 1024, 1024]), (2560, [9728, 9728]), (1024, [3072, 3072]), (256, [48, 16])])
@pytest.mark.parametrize("rows", [1, 3])
def test_fused_rmsnorm_prologue(dev, K, sizes, rows):
    """y = linear(rmsnorm(x) * w): the norm weight folded into the channel scales, the rsqrt scalar applied in the
    kernel from the sum(x^2) it gathers while seeding the rotation -- against the oracle on the explicitly normalised x."""
    from paroquant_amd import ops, _native as nat
    L = po.make_layer(K + rows, K, sizes)
    rng = np.random.default_rng(K)
    w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
    x = (rng.standard_normal((rows, K)) * 3.0).astype(np.float16)
    pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
    res = rng.standard_normal((rows, sum(sizes))).astype(np.float16)
    y = ops.w4a16_gemv_fused(_t(x, dev), pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=_t(res, dev))
    xn = po.rmsnorm(x, w, 1e-6)
    ideal = po.paro_linear_merged(xn, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], sizes, None, ideal=True) + res.astype(np.float64)
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    # without the residual, and a call through a strided view of a wider buffer
    wide = torch.zeros(rows, K + 64, device=dev, dtype=torch.float16)
    wide[:, :K] = _t(x, dev)
    y2 = ops.w4a16_gemv_fused(wide[:, :K], pk, nat.PROLOGUE_RMSNORM, 1e-6)
    assert po.rel_err(_np(y2), ideal - res.astype(np.float64)) < TIGHT_F16


@pytest.mark.parametrize("rows", [1, 2, 3, 4])
def test_fused_gemv_repeated_calls_are_deterministic(dev, rows):
    """The same fused launches 40 times: bit-identical outputs every time (a build whose results vary from run to run passes a
    tolerance test most of the time -- this is the test that catches it; tools/stress_fused.py is the long version)."""
    from paroquant_amd import ops, _native as nat
    for K, sizes in [(2560, [4096, 1024, 1024]), (1024, [3072, 3072])]:
        L = po.make_layer(K + rows, K, sizes)
        rng = np.random.default_rng(K + rows)
        w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
        pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
        x = _t((rng.standard_normal((rows, K)) * 3.0).astype(np.float16), dev)
        res = _t(rng.standard_normal((rows, sum(sizes))).astype(np.float16), dev)
        wide = torch.zeros(rows, K + 64, device=dev, dtype=torch.float16)
        wide[:, :K] = x
        first = None
        for it in range(40):
            y = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=res)
            y2 = ops.w4a16_gemv_fused(wide[:, :K], pk, nat.PROLOGUE_RMSNORM, 1e-6)
            y3 = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_NONE, residual=res)
            torch.cuda.synchronize()
            if first is None:
                first = (y.clone(), y2.clone(), y3.clone())
            else:
                assert torch.equal(y, first[0]) and torch.equal(y2, first[1]) and torch.equal(y3, first[2]), (K, it)


@pytest.mark.parametrize("K,N", [(9728, 2560), (3072, 1024), (14336, 4096), (512, 48)])
@pytest.mark.parametrize("rows", [1, 4])
def test_fused_silu_mul_prologue_and_residual(dev, K, N, rows):
    """down_proj with the SiLU(gate) * up prologue on the merged gate_up output and the residual epilogue (K-split
    launch shapes included: 9728 -> 2560 and 14336 -> 4096 resolve to t4 k4 w8)."""
    from paroquant_amd import ops, _native as nat
    L = po.make_layer(K + N + rows, K, [N])
    rng = np.random.default_rng(N)
    gu = rng.standard_normal((rows, 2 * K)).astype(np.float16)
    res = rng.standard_normal((rows, N)).astype(np.float16)
    pk = _packed(L, dev)
    y = ops.w4a16_gemv_fused(_t(gu, dev), pk, nat.PROLOGUE_SILU_MUL, residual=_t(res, dev))
    act = po.silu_mul(gu, K)
    ideal = po.paro_linear_merged(act, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                  L["channel_scales"], [N], None, ideal=True) + res.astype(np.float64)
    assert po.rel_err(_np(y), ideal) < TIGHT_F16
    # residual only (no prologue) == plain linear + residual
    x = rng.standard_normal((rows, K)).astype(np.float16)
    y3 = ops.w4a16_gemv_fused(_t(x, dev), pk, nat.PROLOGUE_NONE, residual=_t(res, dev))
    assert torch.equal(y3, (pk.apply(_t(x, dev)).float() + _t(res, dev).float()).half()) or \
        po.rel_err(_np(y3), _np(pk.apply(_t(x, dev))) + res.astype(np.float64)) < 1e-3
    torch.cuda.synchronize()
    ops.check_workspace(pk.workspace)
    with pytest.raises(RuntimeError, match="at most 4 rows"):
        ops.w4a16_gemv_fused(torch.zeros(5, 2 * K, device=dev, dtype=torch.float16), pk, nat.PROLOGUE_SILU_MUL)


@pytest.mark.parametrize("hd,Hq,Hkv,qk_norm", [(128, 8, 2, True), (64, 4, 2, False), (128, 32, 8, True), (128, 4, 4, False)])
@pytest.mark.parametrize("pos", [0, 5, 300, 1023])
def test_attn_decode_matches_oracle(dev, hd, Hq, Hkv, qk_norm, pos):
    """Decode attention (q/k norm + RoPE + KV append + GQA softmax) against the float64 oracle."""
    from paroquant_amd import ops
    rng = np.random.default_rng(hd + Hq + pos)
    T = 1024
    qkv = rng.standard_normal((Hq + 2 * Hkv) * hd).astype(np.float16)
    kc = (rng.standard_normal((Hkv, T, hd))).astype(np.float16)
    vc = (rng.standard_normal((Hkv, T, hd))).astype(np.float16)
    qw = (1 + 0.2 * rng.standard_normal(hd)).astype(np.float16) if qk_norm else None
    kw = (1 + 0.2 * rng.standard_normal(hd)).astype(np.float16) if qk_norm else None
    cos, sin = po.rope_tables(hd, T, 1e4)
    rope = torch.from_numpy(np.concatenate([cos, sin], axis=-1).astype(np.float32)).to(dev)
    kct, vct = _t(kc, dev), _t(np.ascontiguousarray(vc.transpose(0, 2, 1)), dev)     # V cache: [head][dim][position]
    out = ops.attn_decode(_t(qkv, dev), kct, vct, torch.tensor([pos], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd,
                          None if qw is None else _t(qw, dev), None if kw is None else _t(kw, dev), 1e-6)
    ref, k_new, v_new = po.attention_decode(qkv, kc, vc, pos, Hq, Hkv, hd, cos, sin, qw, kw, 1e-6)
    assert po.rel_err(_np(out), ref) < 4e-3
    assert po.rel_err(_np(kct[:, pos]), k_new) < 2e-3 and po.rel_err(_np(vct[:, :, pos]), v_new) < 1e-6
    if pos > 0:   # the rest of the cache is untouched
        vt = _t(np.ascontiguousarray(vc.transpose(0, 2, 1)), dev)
        assert torch.equal(kct[:, :pos], _t(kc, dev)[:, :pos]) and torch.equal(vct[:, :, pos + 1:], vt[:, :, pos + 1:]) \
            and torch.equal(vct[:, :, :pos], vt[:, :, :pos])


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 3])
def test_fused_prologues_and_attention_bf16(dev, rows):
    """bf16 activations through the decode-layer fusions (RMSNorm prologue + residual, SiLU*mul prologue) and the decode
    attention kernel (bf16 K / V caches, probabilities rounded to bf16 for the P V product as HF does)."""
    from paroquant_amd import ops, _native as nat
    bf = torch.bfloat16
    K, sizes = 1024, [3072, 3072]
    L = po.make_layer(K + rows + 7, K, sizes)
    rng = np.random.default_rng(rows + 70)
    w = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float16)
    x = _t((rng.standard_normal((rows, K)) * 3.0).astype(np.float32), dev).to(bf)
    res = _t(rng.standard_normal((rows, sum(sizes))).astype(np.float32), dev).to(bf)
    pk = _packed(L, dev).fold_norm_weight(_t(w, dev))
    y = ops.w4a16_gemv_fused(x, pk, nat.PROLOGUE_RMSNORM, 1e-6, residual=res)
    assert y.dtype == bf
    xn = po.rmsnorm(x.float().cpu().numpy(), w, 1e-6)
    ideal = po.paro_linear_merged(xn, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes,
                                  None, ideal=True) + res.float().cpu().numpy().astype(np.float64)
    assert po.rel_err(y.float().cpu().numpy(), ideal) < 2e-2
    Ld = po.make_layer(K + rows + 8, 3072, [1024])
    gu = _t(rng.standard_normal((rows, 2 * 3072)).astype(np.float32), dev).to(bf)
    yd = ops.w4a16_gemv_fused(gu, _packed(Ld, dev), nat.PROLOGUE_SILU_MUL)
    ideal_d = po.paro_linear_merged(po.silu_mul(gu.float().cpu().numpy(), 3072), Ld["qweight"], Ld["qzeros"], Ld["scales"], Ld["theta"],
                                    Ld["pairs"], Ld["channel_scales"], [1024], None, ideal=True)
    assert po.rel_err(yd.float().cpu().numpy(), ideal_d) < 2e-2
    # attention, bf16
    hd, Hq, Hkv, T, pos = 128, 8, 2, 512, 300 + rows
    qkv = _t(rng.standard_normal((Hq + 2 * Hkv) * hd).astype(np.float32), dev).to(bf)
    kc = _t(rng.standard_normal((Hkv, T, hd)).astype(np.float32), dev).to(bf)
    vc = _t(rng.standard_normal((Hkv, T, hd)).astype(np.float32), dev).to(bf)
    qw = _t((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32), dev).to(bf)
    kw = _t((1 + 0.2 * rng.standard_normal(hd)).astype(np.float32), dev).to(bf)
    cos, sin = po.rope_tables(hd, T, 1e4)
    rope = torch.from_numpy(np.concatenate([cos, sin], axis=-1).astype(np.float32)).to(dev)
    kct, vct = kc.clone(), vc.transpose(1, 2).contiguous()
    out = ops.attn_decode(qkv, kct, vct, torch.tensor([pos], dtype=torch.int32, device=dev), rope, Hq, Hkv, hd, qw, kw, 1e-6)
    f = lambda t: t.float().cpu().numpy()
    ref, k_new, v_new = po.attention_decode(f(qkv), f(kc), f(vc), pos, Hq, Hkv, hd, cos, sin, f(qw), f(kw), 1e-6)
    assert out.dtype == bf and po.rel_err(f(out), ref) < 3e-2
    assert po.rel_err(f(kct[:, pos]), k_new) < 2e-2 and po.rel_err(f(vct[:, :, pos]), v_new) < 1e-6


# ---------------------------------------------------------------- f2: the decode harness against HF on the same checkpoint

@pytest.mark.parametrize("model_type,head_dim", [("llama", 64), ("qwen3", 128)])
def test_decoder_harness_matches_hf(dev, tmp_path, model_type, head_dim):
    """ParoDecoderLM (5 fused launches per layer, HIP-graph decode) against HF's own modelling code running our
    RotateQuantizedLinear modules on the SAME synthetic PARO checkpoint: prefill logits, then greedy decode steps
    (graph replays) against HF logits of the growing sequence."""
    import paroquant_amd.hf_quantizer  # noqa: F401
    from paroquant_amd.decoder import ParoDecoderLM
    from tests.hf_ckpt import write_tiny_paro_llama
    from transformers import AutoModelForCausalLM
    # hidden 512 so that the fused tail (final norm + lm_head GEMV + argmax kernels) is on the tested path
    write_tiny_paro_llama(str(tmp_path), hidden=512, inter=1024, heads=8 if head_dim == 64 else 4, kv_heads=2, layers=2, vocab=200,
                          model_type=model_type, head_dim=head_dim, seed=3)
    try:
        hf = AutoModelForCausalLM.from_pretrained(str(tmp_path), dtype=torch.float16, device_map={"": "cuda:0"})
    except TypeError:
        hf = AutoModelForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.float16, device_map={"": "cuda:0"})
    hf.eval()
    lm = ParoDecoderLM.from_checkpoint(str(tmp_path), dev, max_positions=64)
    assert lm.cfg.qk_norm == (model_type == "qwen3") and lm.fused_tail
    ids = torch.randint(0, 200, (11,), device=dev)
    logits = lm.prefill(ids)
    with torch.no_grad():
        ref = hf(input_ids=ids[None]).logits[0, -1].float()
    scale = ref.abs().max().item()
    assert (logits[0].float() - ref).abs().max().item() < 2e-2 * scale
    lm.capture()
    seq = ids.clone()
    for step in range(6):
        tok = lm.tok.clone()
        seq = torch.cat([seq, tok])
        lm._graph.replay()
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = hf(input_ids=seq[None]).logits[0, -1].float()
        got = lm.logits[0].float()
        assert (got - ref).abs().max().item() < 3e-2 * ref.abs().max().item(), step
        # greedy token agrees unless HF's own top-2 logits are closer than the tolerance
        top2 = ref.topk(2).values
        if (top2[0] - top2[1]).item() > 6e-2 * ref.abs().max().item():
            assert int(lm.tok.item()) == int(ref.argmax().item())
    toks, stats = ParoDecoderLM.from_checkpoint(str(tmp_path), dev, max_positions=64).generate(ids, 8)
    assert toks.shape == (19,) and torch.equal(toks[:11], ids) and stats["new_tokens"] == 8


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decoder_harness_deferred_matches_reducer(dev, dtype, monkeypatch):
    """One GPU: o / down leave their K-split partial sums to the RMSNorm-prologue launch behind them (decoder._layers_deferred) --
    logits bit for bit and tokens one for one what the in-launch reducer route (PARO_DEFERRED_KSPLIT=0) gives, eager and graph."""
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    # o: 3072 -> 512, down: 3072 -> 512 -- 24 groups each: the automatic per-call shape K-splits them 2-way like the deferred route does
    # (below 24 groups the per-call route no longer splits at all -- round-4 re-sweep, gemv.hip -- and the two routes then differ by the
    # summation order, i.e. at rounding level)
    cfg = lambda: DecoderConfig(512, 3072, 24, 4, 128, 3, 640, 1e-6, 10000.0, True, 64)
    ids = torch.tensor([3, 17, 101, 7, 250, 9, 33], device=dev)
    monkeypatch.setenv("PARO_DEFERRED_QKV", "0")           # (the 2-way qkv changes qkv's own summation order: its own test below)
    monkeypatch.setenv("PARO_SPLIT_ATTN", "0")             # (the split attention merges 128-position slots: another rounding, its own test below)
    lm_d = ParoDecoderLM.random(cfg(), dev, seed=9, dtype=dtype)
    assert lm_d.deferred and not lm_d.deferred_qkv and not lm_d.split_attn
    monkeypatch.setenv("PARO_DEFERRED_KSPLIT", "0")
    lm_r = ParoDecoderLM.random(cfg(), dev, seed=9, dtype=dtype)
    assert not lm_r.deferred
    for use_graph in (False, True):
        td, _ = lm_d.generate(ids, 10, use_graph=use_graph)
        tr, _ = lm_r.generate(ids, 10, use_graph=use_graph)
        assert torch.equal(td, tr)
        assert torch.equal(lm_d.logits, lm_r.logits)
    from paroquant_amd import ops
    ops.check_workspace(lm_d.layers[0].o.workspace)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decoder_harness_split_attention(dev, dtype, monkeypatch):
    """The attention's merge over position chunks completed by o_proj (ABI v14, decoder.split_attn) against the in-launch merge: the
    logits of every teacher-forced step agree to rounding (another partition of the positions, one more fp32 rescale), eager and graph;
    positions cross the 128-position slot boundary."""
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    cfg = lambda: DecoderConfig(512, 2048, 16, 2, 128, 3, 640, 1e-6, 10000.0, True, 320)
    ids = torch.randint(0, 640, (150,), device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    lm_s = ParoDecoderLM.random(cfg(), dev, seed=9, dtype=dtype)
    assert lm_s.deferred and lm_s.split_attn
    monkeypatch.setenv("PARO_SPLIT_ATTN", "0")
    lm_m = ParoDecoderLM.random(cfg(), dev, seed=9, dtype=dtype)
    assert lm_m.deferred and not lm_m.split_attn
    tol = 2e-2 if dtype == torch.float16 else 8e-2
    for use_graph in (False, True):
        for lm in (lm_s, lm_m):
            lm.prefill(ids[:120])
            if use_graph:
                lm.capture()
        for i in range(120, 150):                    # teacher-forced: both models see the same tokens at positions 120..149
            for lm in (lm_s, lm_m):
                lm.tok.copy_(ids[i:i + 1])
                lm._graph.replay() if use_graph else lm.decode_step()
            a, b = lm_s.logits.float(), lm_m.logits.float()
            assert (a - b).abs().max().item() < tol * b.abs().max().item(), (use_graph, i)
    from paroquant_amd import ops
    ops.check_workspace(lm_s.layers[0].o.workspace)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decoder_harness_deferred_qkv(dev, dtype, monkeypatch):
    """qkv as a 2-way K-split producer whose partial sums (and sums of squares: its RMSNorm prologue) the attention kernel completes:
    teacher-forced decode steps against the same model with qkv reduced in its own launch -- equal within rounding (the fp32 summation
    order of qkv differs), KV caches included."""
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    cfg = lambda: DecoderConfig(2048, 2048, 16, 2, 128, 2, 640, 1e-6, 10000.0, True, 64)     # qkv 2048 -> 2560: splits 2-way when nobody polls
    ids = torch.tensor([3, 17, 101, 7, 250, 9, 33], device=dev)
    lm_q = ParoDecoderLM.random(cfg(), dev, seed=4, dtype=dtype)
    assert lm_q.deferred and lm_q.deferred_qkv
    monkeypatch.setenv("PARO_DEFERRED_QKV", "0")
    lm_r = ParoDecoderLM.random(cfg(), dev, seed=4, dtype=dtype)
    assert lm_r.deferred and not lm_r.deferred_qkv
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-3
    lq, lr = lm_q.prefill(ids), lm_r.prefill(ids)
    assert torch.equal(lq, lr)                               # the prefill path is the same code
    for step in range(4):
        lm_r.tok.copy_(lm_q.tok)                             # teacher forcing: both consume the same token
        lm_q.decode_step(); lm_r.decode_step()
        torch.cuda.synchronize()
        a, b = lm_q.logits.float(), lm_r.logits.float()
        assert torch.isfinite(a).all() and ((a - b).abs().max() / b.abs().max()).item() < tol, step
    for Lq, Lr in zip(lm_q.layers, lm_r.layers):
        n = int(lm_q.pos.item())
        assert ((Lq.kcache[:, :n].float() - Lr.kcache[:, :n].float()).abs().max() / Lr.kcache[:, :n].float().abs().max()).item() < tol
        assert ((Lq.vcache[:, :, :n].float() - Lr.vcache[:, :, :n].float()).abs().max() / Lr.vcache[:, :, :n].float().abs().max()).item() < tol
    tq, _ = lm_q.generate(ids, 6, use_graph=True)            # and the graph replay runs
    assert tq.numel() == ids.numel() + 6


@pytest.mark.gpu
def test_decoder_harness_bf16_and_graph_consistency(dev):
    """The decode harness in bf16: the HIP-graph replay path reproduces the eager launch sequence token for token, and the
    bf16 model's first decode steps agree with the fp16 model built from the same seed (same synthetic weights)."""
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    cfg = lambda: DecoderConfig(256, 512, 4, 2, 64, 2, 512, 1e-6, 10000.0, True, 128)
    ids = torch.tensor([3, 17, 101, 7, 250, 9, 33], device=dev)
    outs = {}
    for dt in (torch.float16, torch.bfloat16):
        lm_g = ParoDecoderLM.random(cfg(), dev, seed=5, dtype=dt)
        lm_e = ParoDecoderLM.random(cfg(), dev, seed=5, dtype=dt)
        tg, _ = lm_g.generate(ids, 12, use_graph=True)
        te, _ = lm_e.generate(ids, 12, use_graph=False)
        assert torch.equal(tg, te)                        # graph replay == eager, token for token
        lm_p = ParoDecoderLM.random(cfg(), dev, seed=5, dtype=dt)
        outs[dt] = lm_p.prefill(ids).float()              # logits of the last prompt position
        assert torch.isfinite(outs[dt]).all()
    a, b = outs[torch.float16].flatten(), outs[torch.bfloat16].flatten()
    assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.99


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("V,H", [(1000, 512), (151936, 2560), (4099, 4096)])
def test_lm_head_and_argmax(dev, dtype, V, H):
    """Final RMSNorm + lm_head GEMV + greedy argmax kernels against torch on the same tensors."""
    from paroquant_amd import ops
    gen = torch.Generator(device=dev); gen.manual_seed(V + H)
    W = (torch.randn(V, H, device=dev, generator=gen) * H ** -0.5).to(dtype)
    x = (torch.randn(1, H, device=dev, generator=gen) * 3).to(dtype)
    nw = (1 + 0.1 * torch.randn(H, device=dev, generator=gen)).to(dtype)
    logits = torch.empty(1, V, device=dev, dtype=dtype)
    ws = ops.lm_head_workspace(dev, V)
    ops.lm_head(x, nw, W, logits, 1e-6, ws)
    xf = x.float()
    xn = ((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype) * nw)
    ref = (xn.double() @ W.double().t())
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    assert ((logits.double() - ref).abs().max() / ref.abs().max()).item() < tol
    tok = torch.tensor([7], dtype=torch.int64, device=dev)
    pos = torch.tensor([3], dtype=torch.int32, device=dev)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    ops.argmax_advance(ws, V, tok, pos, out)
    assert int(tok.item()) == int(torch.argmax(logits, dim=-1).item()) and int(pos.item()) == 4 and int(out[3].item()) == 7


# ---------------------------------------------------------------- f4: mixture-of-experts (shared rotation, routed experts)

@pytest.mark.parametrize("E,H,I,T,k", [(8, 512, 256, 1, 2), (16, 2048, 768, 1, 8), (8, 512, 256, 3, 4), (4, 256, 128, 40, 2)])
def test_moe_experts_match_oracle(dev, E, H, I, T, k):
    """Routed experts with one shared rotation per projection (cli/convert.py:280-379, mlx/modules.py:159-212): the
    slot kernels (decode: T * k <= 64, two launches for all slots) and the grouped prefill path against the oracle."""
    from paroquant_amd.moe import ParoMoEExperts
    experts, rot = po.make_moe(E * 1000 + H + T, E, H, I)
    tensors = {}
    for proj, d in experts.items():
        for name, stack in d.items():
            for e in range(E):
                tensors[f"{e}.{proj}.{name}"] = torch.from_numpy(stack[e])
    for name, v in rot.items():
        tensors[name] = torch.from_numpy(v)
    moe = ParoMoEExperts(tensors, E, dev)
    rng = np.random.default_rng(T + k)
    x = rng.standard_normal((T, H)).astype(np.float16)
    idx = np.stack([rng.choice(E, size=k, replace=False) for _ in range(T)]).astype(np.int64)
    y = moe(_t(x, dev), _t(idx, dev))
    assert y.shape == (T, k, H) and y.dtype == torch.float16
    ref = po.moe_experts_forward(x, idx, experts, rot)
    assert po.rel_err(_np(y), ref) < 4e-3
    # deterministic, and capturable at decode size (expert ids come from device memory)
    if T * k <= 64:
        assert torch.equal(moe(_t(x, dev), _t(idx, dev)), y)
        xs, ids = _t(x, dev), _t(idx, dev)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            moe(xs, ids)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            yg = moe(xs, ids)
        ids.copy_(torch.flip(ids, dims=[1]))        # other experts per slot, same graph
        g.replay()
        torch.cuda.synchronize()
        assert po.rel_err(_np(yg), ref[:, ::-1]) < 4e-3


# ---------------------------------------------------------------- e: the tensor-parallel bench path on real kernels

@pytest.mark.parametrize("world,workload,layers", [(2, "llama3-70b-tp", 2), (4, "qwen3-32b-tp", 2),
                                                   (4, "qwen3.5-27b-class-tp", 4)])     # BASELINE config 5: 3 delta-net layers + 1 full-attention layer
def test_tp_bench_path_on_one_gpu(dev, world, workload, layers):
    """`bench.py --gpus N --workload <70B-class>-tp` with N processes sharing this one GPU (gloo instead of RCCL, which
    refuses two ranks per device): every rank builds its Megatron shard, runs it through the HIP kernels, all-reduces
    after o / down, and rank 0 prints the contract line with n_gpus = N, scaling strong and the TP = 1 reference."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--workload", workload, "--layers", str(layers),
                          "--tp-backend", "gloo", "--same-device", "--steps", "3", "--warmup", "1"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == world and r["scaling"] == "strong" and r["config"]["parallelism"] == f"tp{world}"
    assert r["value"] > 0 and r["config"]["tp1_reference"]["tokens_per_s"] > 0
    assert r["roofline"]["launches_per_step"] == 4 * layers
    # the collective is the one-shot kernel (self-tested against gloo inside the bench), so the TP step is one HIP graph
    assert r["config"]["allreduce"] == "oneshot" and r["config"]["hip_graph"] is True
    # ... and the library collective was timed on the same shards (A/B, the faster healthy leg is the headline)
    ab = r["config"]["allreduce_ab"]
    assert set(ab) == {"oneshot", "gloo"} and ab["oneshot"]["ms_per_step"] > 0 and ab["gloo"].get("ms_per_step", 0) > 0, ab
    out2 = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--workload", workload, "--layers", str(layers), "--no-oneshot",
                           "--tp-backend", "gloo", "--same-device", "--steps", "3", "--warmup", "1"],
                          cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-3000:]
    r2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith("{")][0])
    assert r2["config"]["allreduce"] == "gloo" and r2["config"]["hip_graph"] is False


def test_tp_bench_end_to_end_leg(dev):
    """At full depth the TP bench also reports the whole model tensor-parallel on the fused harness (`end_to_end`, tp2)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "qwen3-0.6b-tp", "--tp-backend", "gloo", "--same-device",
                          "--steps", "3", "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    e = r["end_to_end"]
    assert "error" not in e, e
    assert e["value"] > 0 and e["parallelism"].startswith("tp2") and e["launches_per_token"] == 5 * 28 + 3


# ---------------------------------------------------------------- e: one-shot all-reduce of the row-parallel outputs

@pytest.mark.parametrize("world", [2, 4])
def test_oneshot_allreduce_ranks_share_one_gpu(dev, world):
    """paro_allreduce_oneshot through CUDA-IPC-mapped peer buffers, `world` processes on this one GPU (gloo as the control
    channel): against gloo's all-reduce, bit-identical across ranks, fp16 / bf16, and replayed from a HIP graph."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29600 + world), os.path.join(root, "tests", "_oneshot_worker.py")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ONESHOT_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@pytest.mark.parametrize("world", [2, 4])
def test_tp_decoder_ranks_share_one_gpu(dev, world):
    """The end-to-end decode harness tensor-parallel (heads and MLP columns sharded, one-shot all-reduce with the residual
    added in its summation after o / down): logits and greedy tokens of the unsharded model, eager and from a HIP graph."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29620 + world), os.path.join(root, "tests", "_tp_decoder_worker.py")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "TP_DECODER_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_oneshot_allreduce_absent_peer_gives_up_once(dev):
    """A peer that never arrives: the call gives up after its bounded wait (sticky status word, reported by
    paro_allreduce_status), and every LATER call polls once instead of waiting again -- an out-of-step rank costs seconds
    once, not seconds per all-reduce."""
    import ctypes
    import time
    from paroquant_amd import _native as nat
    lib = nat.load()
    n = 4096
    nbytes = lib.paro_allreduce_buffer_bytes(2, n)
    bufs = []
    for _ in range(2):                      # both "ranks'" buffers live in this process; rank 1 never calls
        p, h = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        nat.check(lib.paro_allreduce_buffer_create(nbytes, ctypes.byref(p), h))
        bufs.append(p.value)
    try:
        peers = (ctypes.c_void_p * 2)(*bufs)                    # host array of the two buffers
        x = torch.randn(n, device=dev, dtype=torch.float16)
        y = torch.empty_like(x)
        st = nat.current_stream_ptr(dev)
        call = lambda: nat.check(lib.paro_allreduce_oneshot(x.data_ptr(), None, y.data_ptr(), n, nat.dtype_code(x.dtype), ctypes.cast(peers, ctypes.c_void_p), 2, 0, n, st))
        assert lib.paro_allreduce_status(bufs[0], st) == 0
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize(dev)
        first = time.perf_counter() - t0
        assert lib.paro_allreduce_status(bufs[0], st) != 0          # gave up, and says so
        assert 0.05 < first < 60.0, first                           # a bounded wait of seconds
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        torch.cuda.synchronize(dev)
        later = time.perf_counter() - t0
        assert later < 0.05 * max(first, 1.0), (first, later)       # no second wait
        assert lib.paro_allreduce_status(bufs[0], st) != 0          # sticky
    finally:
        torch.cuda.synchronize(dev)
        for b in bufs:
            lib.paro_allreduce_buffer_destroy(b)


def test_prefill_merged_projection_in_a_graph(dev):
    """The prefill path of a merged projection (csrc/gemm.hip: pre-pass + GEMM inside ONE call) gives the same bits eagerly, from a
    captured HIP graph and over repeated calls on one stream; reference semantics: per-partition rotate then GEMM,
    vllm/plugin.py:288-306.  (Round 5's two-stream per-partition variant of this call was measured slower and removed in round 6:
    profiles/NOTES.md 5.4.)"""
    K, sizes, rows = 2048, [2048, 512, 512], 4096
    L = _random_gpu_layer(dev, K, sizes, seed=77)
    pk = _pack_gpu_layer(L).prepare_prefill(torch.float16)
    x = torch.randn(rows, K, device=dev, dtype=torch.float16)
    sample = torch.arange(0, rows, rows // 32, device=dev)
    ideal = _oracle_rows(L, x[sample])
    y_single = pk.apply(x).clone()
    _overlap_checks(dev, pk, x, sample, ideal, y_single)


def _overlap_checks(dev, pk, x, sample, ideal, y_single):
    y_eager = pk.apply(x).clone()
    assert torch.equal(y_eager, y_single)
    assert po.rel_err(_np(y_eager[sample]), ideal) < TIGHT_F16
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        pk.apply(x)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = pk.apply(x)
    for _ in range(3):
        y_g.zero_()
        g.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(y_g, y_eager)
    # three calls back to back on one stream (the events are reused), then the result of the last
    for _ in range(3):
        y2 = pk.apply(x)
    assert torch.equal(y2, y_eager)


# ---------------------------------------------------------------- prompt pass of the decode harness (ABI v19, csrc/prompt.hip)

@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("T,nh,nkv,hd,qk_norm,pos0", [(1, 4, 2, 128, True, 0), (37, 8, 2, 128, True, 5), (128, 6, 3, 64, False, 0), (9, 5, 1, 32, True, 100)])
def test_prompt_pass_kernels_match_framework_expressions(dev, dtype, tol, T, nh, nkv, hd, qk_norm, pos0):
    """paro_prompt_row_rms / paro_prompt_qkv_post / paro_prompt_silu_mul against the framework expressions they replace in
    `ParoDecoderLM.prefill` (themselves HF's prompt pass with the RMSNorm weights folded away, transformers/generator.py:37-67): the row
    scale, q / k head RMSNorm, rotary embedding, the attention's inputs and BOTH decode caches (K position-major, V position-contiguous);
    SiLU * up.  Tolerance: two roundings of the activation type."""
    from paroquant_amd import ops
    gen = torch.Generator(device=dev); gen.manual_seed(T * 131 + hd)
    hidden, inter, T_max, eps = 256, 384, 160, 1e-6
    h = torch.randn(T, hidden, device=dev, generator=gen).to(dtype)
    rs_ref = torch.rsqrt(h.float().pow(2).mean(-1, keepdim=True) + eps)
    rs = ops.prompt_row_rms(h, eps)
    assert rs.shape == (T,) and torch.allclose(rs, rs_ref[:, 0], rtol=1e-5, atol=0)
    raw = torch.randn(T, (nh + 2 * nkv) * hd, device=dev, generator=gen).to(dtype)
    qn = (1.0 + 0.05 * torch.randn(hd, device=dev, generator=gen)).to(dtype) if qk_norm else None
    kn = (1.0 + 0.05 * torch.randn(hd, device=dev, generator=gen)).to(dtype) if qk_norm else None
    ang = torch.arange(T_max, dtype=torch.float32, device=dev)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, device=dev).float() / hd))[None, :]
    rope = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()
    kc = torch.zeros(nkv, T_max, hd, dtype=dtype, device=dev)
    vc = torch.zeros(nkv, hd, T_max, dtype=dtype, device=dev)
    q, k, v = ops.prompt_qkv_post(raw, rs, rope, kc, vc, nh, nkv, hd, qn, kn, eps, pos0=pos0)
    # the framework expressions (decoder.py, PARO_PROMPT_TORCH=1)
    half = hd // 2
    cos, sin = rope[pos0:pos0 + T, :half].to(dtype)[:, None, :], rope[pos0:pos0 + T, half:].to(dtype)[:, None, :]
    rsx = lambda x: torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)
    rot = lambda x: torch.cat([x[..., :half] * cos - x[..., half:] * sin, x[..., half:] * cos + x[..., :half] * sin], dim=-1)
    hn = lambda x, w: x if w is None else ((x.float() * rsx(x)).to(dtype) * w)
    qkv = (raw.float() * rs_ref).to(dtype)
    qr, kr, vr = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    qr, kr, vr = rot(hn(qr.view(T, nh, hd), qn)), rot(hn(kr.view(T, nkv, hd), kn)), vr.view(T, nkv, hd)
    close = lambda a, b: (a.float() - b.float()).abs().max().item() <= tol * max(b.float().abs().max().item(), 1e-6)
    assert close(q, qr) and close(k, kr) and close(v, vr)
    assert torch.equal(kc[:, pos0:pos0 + T], k.transpose(0, 1)) and torch.equal(vc[:, :, pos0:pos0 + T], v.permute(1, 2, 0))
    assert not kc[:, :pos0].any() and not kc[:, pos0 + T:].any() and not vc[:, :, :pos0].any() and not vc[:, :, pos0 + T:].any()
    gu = torch.randn(T, 2 * inter, device=dev, generator=gen).to(dtype)
    act = ops.prompt_silu_mul(gu, rs)
    gus = (gu.float() * rs_ref).to(dtype)
    assert act.shape == (T, inter) and close(act, torch.nn.functional.silu(gus[:, :inter]) * gus[:, inter:])
    with pytest.raises(RuntimeError, match="head_dim"):
        ops.prompt_qkv_post(torch.zeros(1, 4 * 256, dtype=dtype, device=dev), None, torch.zeros(8, 256, device=dev), torch.zeros(1, 8, 256, dtype=dtype, device=dev),
                            torch.zeros(1, 256, 8, dtype=dtype, device=dev), 2, 1, 256)


@pytest.mark.gpu
def test_captured_graph_survives_workspace_growth(dev):
    """A HIP graph holds the address of the workspace it was captured on; a later, larger call grows the shared workspace
    (`ops.get_workspace`).  The old buffer must stay allocated: replays keep writing their K-split granules / rotated rows there
    (round 6: it was dropped, and a replay wrote through a dangling pointer -- found by tools/sweep_gemm4.py as a GPU memory fault)."""
    from paroquant_amd import ops
    K, sizes = 4096, [2560]
    L = po.make_layer(4242, K, sizes)
    pk = _packed(L, dev)
    x = torch.randn(24, K, device=dev, dtype=torch.float16)        # 17..32 rows: pre-pass into the workspace + K-split granules
    y0 = pk.apply(x).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = pk.apply(x)
    before = ops.get_workspace(dev, 1).data_ptr()
    big = torch.randn(3000, K, device=dev, dtype=torch.float16)
    ops.get_workspace(dev, ops.get_workspace(dev, 1).numel() * 2 + (64 << 20))     # force the growth, then use the new buffer
    yb = pk.apply(big)
    assert ops.get_workspace(dev, 1).data_ptr() != before and torch.isfinite(yb.float()).all()
    junk = [torch.full((8 << 20,), 7, dtype=torch.int32, device=dev) for _ in range(4)]   # would land in a freed workspace
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and all(int(j[0]) == 7 and int(j[-1]) == 7 for j in junk)


@pytest.mark.gpu
@pytest.mark.parametrize("K,sizes,rows", [(1024, [512, 272], 300), (512, [256], 256), (2560, [1024, 256, 256], 700), (384, [4096], 1000)])
def test_fused_rotation_gemm_experiment_matches_two_launches(dev, K, sizes, rows):
    """GEMM variant 44 -- the north star's fused form (the rotation applied to the LDS-staged slab inside the GEMM, one launch, no rotated
    copy of x; gemm3.hip DIAG 4) -- is an experiment that LOSES by 2.5 .. 3.4x (profiles/r06_fused_rot_gemm_*.jsonl, NOTES 6.11) and is never
    selected; it stays correct: the bits of the rotation pre-pass + variant 4, within tolerance of the oracle (plugin.py:288-306 semantics:
    one rotation per merged partition)."""
    from paroquant_amd import ops
    L = po.make_layer(K + rows, K, sizes, bias=True)
    pk = _packed(L, dev, L["bias"])
    x = np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16)
    y44 = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, True, 44)
    y4 = ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, True, 4)
    assert torch.equal(y44, y4)
    ideal = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes, L["bias"], ideal=True)
    assert po.rel_err(_np(y44), ideal) < TIGHT_F16
    with pytest.raises(RuntimeError, match="rmat|rotation matrices"):
        ops.w4a16_gemm_forced(_t(x, dev), pk, pk.bias, False, 44)

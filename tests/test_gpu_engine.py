"""GPU tests of the persistent decode engines (EXPERIMENTAL build only -- ``paro_engine_*``, csrc/experimental/engine.hip, and the loader / consumer build ``paro_engine2_*``,
csrc/engine2.hip; ``paroquant_amd.engine.DecodeEngine(version=1 | 2)``): a chain of ParoQuant linears at batch 1 in ONE launch.  Checked against the CPU oracle applied linear by linear on the same seeded inputs (the
north star's 1e-2 gate, and the tight tolerance of tests/test_gpu_parity.py per stage), against the per-call kernels on the same chain,
and for run-to-run bit identity (eager and HIP-graph replay): the engine's hand-offs are placement-independent by construction, the
determinism loop is what would catch a protocol error that a tolerance test passes most of the time."""
import numpy as np
import pytest
import torch

from oracle import paro_oracle as po
from tests.test_gpu_parity import REL_TOL, TIGHT_BF16, TIGHT_F16, _np, _packed, _t, dev  # noqa: F401

from tests.conftest import needs_experimental

pytestmark = [pytest.mark.gpu, needs_experimental]   # experimental build only: the engines are not in the default library


def _oracle_chain(x, layers, in_col0, act):
    """The reference's chain, linear by linear, in float64 with ONE rounding to the activation type per linear (what a linear stores
    and the next one reads: transformers/modules.py:57-71); linear i + 1 reads columns in_col0 .. in_col0 + K of linear i's output."""
    cur = np.asarray(x, dtype=np.float64)
    for L, c0 in zip(layers, in_col0):
        K = L["K"]
        y = po.paro_linear_merged(cur[:, c0:c0 + K], L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"],
                                  L["sizes"], L.get("bias"), ideal=True)
        cur = po.round_to(y, act).astype(np.float64)
    return cur


# (K, partition sizes) chains: single linears of every kind, then real edges
SINGLE = [(256, [48, 16]), (1024, [2048, 1024, 1024]), (2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560]),
          (4096, [4096]), (512, [16]), (128, [272])]


VERSIONS = [1, 2]


@pytest.mark.parametrize("version", VERSIONS)
@pytest.mark.parametrize("K,sizes", SINGLE)
@pytest.mark.parametrize("dtype,act,tol", [(torch.float16, "f16", TIGHT_F16), (torch.bfloat16, "bf16", TIGHT_BF16)])
def test_engine_single_linear_matches_oracle(dev, K, sizes, dtype, act, tol, version):
    from paroquant_amd.engine import DecodeEngine
    L = po.make_layer(K + sum(sizes), K, sizes, bias=(K == 4096))
    pk = _packed(L, dev, L.get("bias"))
    eng = DecodeEngine([pk], dtype=dtype, version=version)
    rng = np.random.default_rng(K)
    x = _t(rng.standard_normal((1, K)).astype(np.float32), dev, dtype)
    y = eng(x).clone()
    torch.cuda.synchronize()
    assert eng.status_ok()
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes,
                                  L.get("bias"), ideal=True)
    assert np.isfinite(_np(y)).all() and po.rel_err(_np(y), ideal) < tol
    # the per-call kernel on the same input: two implementations of one function
    y2 = pk.apply(x)
    assert po.rel_err(_np(y), _np(y2)) < tol
    # launch after launch on the same workspace: epochs advance, nothing is re-armed, same bits
    for _ in range(5):
        assert torch.equal(eng(x), y)
    assert eng.status_ok()


CHAINS = [
    # Qwen3-4B decoder layer as bench.py chains it (every linear reads the first K columns of its predecessor), then the next qkv
    ([(2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560]), (2560, [4096, 1024, 1024])], [0, 0, 0, 0, 0]),
    # Qwen3-0.6B layer twice (tiny launches: the hand-offs dominate)
    ([(1024, [2048, 1024, 1024]), (2048, [1024]), (1024, [3072, 3072]), (3072, [1024])] * 2, [0] * 8),
    # a consumer that reads a window of its predecessor (in_col0 != 0), ragged partitions, a bias in the middle
    ([(512, [400, 112]), (256, [384]), (384, [128, 64])], [0, 144, 0]),
    # Llama-3-8B o -> gate_up -> down (the widest edge: 28672 partial-sum columns)
    ([(4096, [4096]), (4096, [14336, 14336]), (14336, [4096])], [0, 0, 0]),
    # a very deep consumer behind a wide producer (Llama-3-70B-class down_proj: 224 groups -- several rotation batches per wave in
    # engine 2 -- and a phase whose tiles exceed the loader's ring several times over)
    ([(1024, [28672]), (28672, [512])], [0, 0]),
]


@pytest.mark.parametrize("version", VERSIONS)
@pytest.mark.parametrize("ci", range(len(CHAINS)))
@pytest.mark.parametrize("dtype,act,tol", [(torch.float16, "f16", TIGHT_F16), (torch.bfloat16, "bf16", TIGHT_BF16)])
def test_engine_chain_matches_oracle_and_per_call_kernels(dev, ci, dtype, act, tol, version):
    from paroquant_amd.engine import DecodeEngine
    shapes, in_col0 = CHAINS[ci]
    layers = []
    for i, (K, sizes) in enumerate(shapes):
        L = po.make_layer(1000 * ci + i, K, sizes, bias=(ci == 2 and i == 1))
        # unit gain so that a chain of eight linears stays in range: scale the scales (a checkpoint-format change, same for every route)
        gain = 1.0 / (6.52 * np.sqrt(K) * np.sqrt(1.75) * np.sqrt(13.0 / 12.0)) / 0.011
        L["scales"] = (L["scales"].astype(np.float32) * gain).astype(np.float16)
        layers.append(L)
    pks = [_packed(L, dev, L.get("bias")) for L in layers]
    eng = DecodeEngine(pks, in_col0=in_col0, dtype=dtype, version=version)
    rng = np.random.default_rng(ci)
    x = _t(rng.standard_normal((1, shapes[0][0])).astype(np.float32), dev, dtype)
    y = eng(x).clone()
    torch.cuda.synchronize()
    assert eng.status_ok() and np.isfinite(_np(y)).all()
    # the per-call kernels over the same chain
    cur = x
    for pk, c0 in zip(pks, in_col0):
        cur = pk.apply(cur[:, c0:c0 + pk.K].contiguous())
    ref = _oracle_chain(_np(x), layers, in_col0, act)
    n = len(shapes)
    # rounding differences are amplified along a chain: the gate is the north star's 1e-2 for the whole chain, and the per-stage
    # tolerance times the depth against the oracle
    assert po.rel_err(_np(y), ref) < min(REL_TOL, tol * n), po.rel_err(_np(y), ref)
    assert po.rel_err(_np(y), _np(cur)) < min(REL_TOL, tol * n)
    for _ in range(10):
        assert torch.equal(eng(x), y)
    assert eng.status_ok()
    # other inputs through the same engine (the epochs keep counting), then the first one again
    x2 = _t(rng.standard_normal((1, shapes[0][0])).astype(np.float32), dev, dtype)
    y2 = eng(x2).clone()
    assert not torch.equal(y2, y) and torch.equal(eng(x), y)


@pytest.mark.parametrize("version", VERSIONS)
def test_engine_graph_replay_and_determinism(dev, version):
    """One captured launch replayed 200 times on changing inputs: every replay equals the eager result of the same input bit for bit
    (no host work between replays: the epoch word advances on the device)."""
    from paroquant_amd.engine import DecodeEngine
    shapes = [(2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560])] * 3
    pks = []
    for i, (K, sizes) in enumerate(shapes):
        L = po.make_layer(77 + i, K, sizes)
        gain = 1.0 / (6.52 * np.sqrt(K) * np.sqrt(1.75) * np.sqrt(13.0 / 12.0)) / 0.011
        L["scales"] = (L["scales"].astype(np.float32) * gain).astype(np.float16)
        pks.append(_packed(L, dev))
    eng = DecodeEngine(pks, version=version)
    xs = [torch.randn(1, 2560, device=dev, dtype=torch.float16) for _ in range(4)]
    eager = [eng(x).clone() for x in xs]
    xbuf = xs[0].clone()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        eng(xbuf)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = eng(xbuf)
    for it in range(200):
        xbuf.copy_(xs[it % 4])
        g.replay()
        assert torch.equal(yg, eager[it % 4]), it
    assert eng.status_ok()
    assert all(s_ >= 1 and mx >= mn for s_, mx, mn in eng.describe())


@pytest.mark.parametrize("version", VERSIONS)
def test_engine_argument_errors(dev, version):
    from paroquant_amd.engine import DecodeEngine as _DE
    DecodeEngine = lambda *a_, **k_: _DE(*a_, version=version, **k_)
    a = _packed(po.make_layer(1, 256, [128]), dev)
    b = _packed(po.make_layer(2, 256, [64]), dev)
    with pytest.raises(RuntimeError, match="reads columns"):
        DecodeEngine([a, b])                      # b needs 256 input channels, a has 128 outputs
    with pytest.raises(RuntimeError, match="reads columns"):
        DecodeEngine([b, _packed(po.make_layer(3, 128, [64]), dev)], in_col0=[0, 1])   # odd window start (and it overruns)
    eng = DecodeEngine([a])
    with pytest.raises(ValueError, match="one contiguous row"):
        eng(torch.zeros(2, 256, device=dev, dtype=torch.float16))
    g64 = po.make_layer(4, 256, [64], group_size=64)
    with pytest.raises(RuntimeError, match="group_size 128"):
        DecodeEngine([_packed(g64, dev)])

"""GPU tests of the measured launch-shape selection (paroquant_amd/autotune.py, ``paro_linear_t.launch_hint``, ABI v16):
the one-time counterpart of the reference's ``process_weights_after_loading`` hook (vllm/plugin.py:251-279).  On every BASELINE decode
shape (Llama-3-8B, Qwen3-4B) and on three shapes no sweep of this repo has seen, the tuned choice is re-measured against the rule tree's
shape in a second, independent measurement; a tuned layer still matches the oracle; two tuning runs agree."""
import numpy as np
import pytest
import torch

from oracle import paro_oracle as po
from tests.test_gpu_parity import TIGHT_F16, _np, _packed, _t, dev  # noqa: F401

pytestmark = pytest.mark.gpu

BASELINE_SHAPES = [
    (4096, [4096]), (4096, [1024]), (4096, [4096, 1024, 1024]), (4096, [14336, 14336]), (14336, [4096]),       # Llama-3-8B
    (2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [9728, 9728]), (9728, [2560]),                           # Qwen3-4B
]
UNSEEN_SHAPES = [(3584, [18944]), (5120, [27648]), (6144, [4096])]


def _synth(K, sizes, dev_, seed=0):
    import bench
    gen = torch.Generator(device=dev_).manual_seed(seed)
    return bench.synth_packed(K, sizes, dev_, gen)


@pytest.mark.parametrize("K,sizes", BASELINE_SHAPES + UNSEEN_SHAPES)
def test_autotuned_shape_is_not_slower_than_the_rule_tree(dev, K, sizes):
    from paroquant_amd import autotune
    pk = _synth(K, sizes, dev)
    rep = pk.autotune(force=True)
    default, choice = tuple(rep["default"]), tuple(rep["choice"])
    assert "%d,%d,%d" % default in rep["candidates"] and len(rep["candidates"]) >= 4
    assert rep["choice_us"] <= rep["default_us"]                      # by construction: the rule tree's shape unless one is >= 2 % ahead
    # an independent second measurement of both shapes (other copies of the weights, other launches)
    again = autotune.measure(pk, sorted({default, choice}), launches=80, reps=5)
    assert again[choice] <= again[default] * 1.03, (rep, again)
    # the hint reaches the drop-in call: auto knobs resolve to the chosen shape (host query), and the layer still computes the same linear
    import ctypes
    from paroquant_amd import _native as nat, ops
    d = ops.pk_desc(pk, torch.float16)
    a, b, c, m = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(-1)
    nat.check(nat.load().paro_gemv_launch_shape(ctypes.byref(d), 1, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(m)))
    assert (a.value, b.value, c.value) == choice
    x = torch.randn(1, K, device=dev, dtype=torch.float16)
    y_tuned = pk.apply(x)
    hint = pk.launch_hint
    pk.launch_hint = 0
    y_rule = pk.apply(x)
    pk.launch_hint = hint
    scale = float(y_rule.float().abs().max())
    assert float((y_tuned.float() - y_rule.float()).abs().max()) <= 4e-3 * scale          # another K partition: fp32 summation order only


def test_autotune_runs_agree_and_cache(dev):
    from paroquant_amd import autotune
    for K, sizes in [(4096, [4096]), (3584, [18944]), (9728, [2560])]:
        pk = _synth(K, sizes, dev)
        r1 = dict(pk.autotune(force=True, launches=80, reps=5))
        r2 = dict(pk.autotune(force=True, launches=80, reps=5))
        if r1["choice"] != r2["choice"]:
            # two shapes inside the noise of this box: the selection rule may then land on either side of its 2 % threshold -- they must
            # really be that close (and the rule tree's shape is within the same margin)
            t = r2["candidates"]
            k1, k2 = "%d,%d,%d" % tuple(r1["choice"]), "%d,%d,%d" % tuple(r2["choice"])
            assert abs(t[k1] - t[k2]) <= 0.04 * min(t[k1], t[k2]), (r1, r2)
        twin = _synth(K, sizes, dev, seed=5)
        r3 = twin.autotune()                                           # same shape: from the cache, no launches
        assert r3["choice"] == r2["choice"] and twin.launch_hint == pk.launch_hint


def test_tuned_layer_matches_oracle_and_bad_hints_are_harmless(dev):
    from paroquant_amd import autotune
    L = po.make_layer(21, 1024, [2048, 512])
    pk = _packed(L, dev)
    x = _t(np.random.default_rng(3).standard_normal((1, 1024)).astype(np.float32), dev, torch.float16)
    ideal = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], [2048, 512], None, ideal=True)
    default, shapes = autotune.candidates(pk)
    assert default in shapes
    for s in shapes:                                                   # every candidate shape computes the same linear
        pk.launch_hint = autotune.launch_hint(*s)
        assert po.rel_err(_np(pk.apply(x)), ideal) < TIGHT_F16, s
    for bad in (0x7fffffff, autotune.launch_hint(3, 200, 5), autotune.launch_hint(8, 16, 16)):
        pk.launch_hint = bad                                           # illegal fields fall back to the rules, legal ones are clamped
        assert po.rel_err(_np(pk.apply(x)), ideal) < TIGHT_F16
    pk.launch_hint = 0
    rep = pk.autotune(force=True)
    assert po.rel_err(_np(pk.apply(x)), ideal) < TIGHT_F16 and rep["choice_us"] > 0

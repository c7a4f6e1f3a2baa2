"""The C restatement (oracle/paro_cpu.c, the CPU-baseline "port") agrees with the numpy oracle."""
import numpy as np
import pytest

from oracle import paro_cpu as pc
from oracle import paro_oracle as po


@pytest.fixture(scope="module", autouse=True)
def _built():
    try:
        pc.load()
    except RuntimeError:
        import subprocess, os
        subprocess.run(["make", "-C", os.path.dirname(pc.__file__)], check=True)
        pc.load()


@pytest.mark.parametrize("rows,K,gs,krot", [(1, 256, 128, 8), (5, 512, 128, 8), (3, 256, 64, 1)])
def test_c_rotate_matches_numpy(rows, K, gs, krot):
    rng = np.random.default_rng(rows + K)
    x = rng.standard_normal((rows, K)).astype(np.float16)
    idx = np.stack([np.concatenate([rng.permutation(gs) for _ in range(K // gs)]) for _ in range(krot)]).astype(np.int16)
    th = (rng.standard_normal((krot, K // 2)) * 0.3).astype(np.float16)
    sc = rng.uniform(0.5, 2, K).astype(np.float16)
    got = pc.rotate_f16(x, idx, th, sc, gs).astype(np.float64)
    ref = po.rotate(x, idx, th, sc, gs, mode="f16")
    # same algorithm, libm sinf/cosf vs numpy: at most an fp16 ulp apart on a handful of elements
    assert po.rel_err(got, ref) < 2e-3
    assert np.mean(got != ref) < 0.02


@pytest.mark.parametrize("rows", [1, 4, 19])
def test_c_linear_matches_numpy(rows):
    L = po.make_layer(17, 512, [128, 64, 64], bias=True)
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, 512)).astype(np.float16)
    got = pc.linear_f16(x, L, L["bias"]).astype(np.float64)
    ref = po.paro_linear_merged(x, L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                L["channel_scales"], L["sizes"], L["bias"], act="f16")
    assert po.rel_err(got, ref) < 2e-3
    assert pc.threads() >= 1

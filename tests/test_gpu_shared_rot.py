"""GPU tests of the GEMV's mode 3 -- the rotation SHARED inside the launch (csrc/gemv_impl.hpp, FUSED | 32; round 6): every (partition,
group, pair of rows) of x is rotated ONCE per launch by a producer workgroup in front of the grid and handed to the workgroups that
multiply by it as {two channels, launch tag} granules; the tag is the hardware's dispatch id of the launch mixed with the queue's address.
What `RotateQuantizedLinear.forward` / `ParoQuantLinearMethod.apply` (transformers/modules.py:57-71, vllm/plugin.py:281-311) reach
from 5 rows on.  Checked: the CPU oracle on the same seeded inputs (the north star's gate and the tight tolerance), BIT identity with
the replicated rotation (mode 0: same arithmetic, same one rounding, same launch shape), fresh inputs on ONE workspace launch after
launch (a stale granule would show), graph replays, two streams taking turns on one workspace (equal dispatch ids on two queues must
not match), two layers of different shapes sharing the workspace, and the fall-back when the grid cannot be resident at once."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import paro_oracle as po
from tests.test_gpu_parity import REL_TOL, TIGHT_BF16, TIGHT_F16, _np, _packed, _t, dev  # noqa: F401

pytestmark = pytest.mark.gpu


def _ideal(L, x):
    return po.paro_linear_merged(np.asarray(x, dtype=np.float64), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"],
                                 L["channel_scales"], L["sizes"], L.get("bias"), ideal=True)


# (K, partition sizes): fewer groups than waves, ragged partitions, merged projections, K-split shapes (narrow N, deep K), a wide one
SHAPES = [(256, [48, 16]), (512, [1024, 256, 256]), (1024, [272]), (2560, [4096, 1024, 1024]), (4096, [2560]), (2560, [2432, 2432]),
          (9728, [640]), (1536, [512]), (128, [16])]


@pytest.mark.parametrize("K,sizes", SHAPES)
@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8, 9, 13, 16])
@pytest.mark.parametrize("dtype,act,tol", [(torch.float16, "f16", TIGHT_F16), (torch.bfloat16, "bf16", TIGHT_BF16)])
def test_shared_rotation_matches_oracle_and_mode0_bits(dev, K, sizes, rows, dtype, act, tol):
    from paroquant_amd import ops
    L = po.make_layer(K * 7 + rows, K, sizes)
    pk = _packed(L, dev)
    rng = np.random.default_rng(K + rows)
    x = _t(rng.standard_normal((rows, K)).astype(np.float32), dev, dtype)
    y3 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 3)
    y0 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0)
    ref = _ideal(L, _np(x))
    assert po.rel_err(_np(y3), ref) < min(REL_TOL, tol)
    assert torch.equal(y3, y0), "mode 3 must return the replicated rotation's bits"
    ops.check_workspace(pk.workspace)


@pytest.mark.parametrize("gs", [64])
@pytest.mark.parametrize("rows", [5, 8, 16])
def test_shared_rotation_group_size_64(dev, gs, rows):
    from paroquant_amd import ops
    K, sizes = 1024, [512, 256]
    from paroquant_amd.linear import PackedParoWeights
    L = po.make_layer(99 + rows, K, sizes, group_size=gs)
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev), _t(L["pairs"], dev),
                           _t(L["channel_scales"], dev), sizes, None, gs)
    x = _t(np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16), dev)
    y3 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 3)
    y0 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0)
    assert torch.equal(y3, y0)
    ref = po.paro_linear_merged(_np(x), L["qweight"], L["qzeros"], L["scales"], L["theta"], L["pairs"], L["channel_scales"], sizes, None,
                                ideal=True, group_size=gs)
    assert po.rel_err(_np(y3), ref) < TIGHT_F16


@pytest.mark.parametrize("krot", [1, 3])
@pytest.mark.parametrize("rows", [6, 16])
def test_shared_rotation_short_schedules_and_bias(dev, krot, rows):
    """Checkpoints with fewer than eight rotation stages (identity-padded schedules: the producers skip the padded stages like the
    replicated form does) and a bias (added once, by whoever writes y: transformers/modules.py:69-70)."""
    from paroquant_amd import ops
    K, sizes = 1024, [768, 256]
    L = po.make_layer(500 + krot + rows, K, sizes, krot=krot, bias=True)
    pk = _packed(L, dev, L["bias"])
    x = _t(np.random.default_rng(krot * 100 + rows).standard_normal((rows, K)).astype(np.float16), dev)
    y3 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 3, bias=pk.bias)      # (the tuning entry takes the bias explicitly)
    y0 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0, bias=pk.bias)
    assert torch.equal(y3, y0)
    assert po.rel_err(_np(y3), _ideal(L, _np(x))) < TIGHT_F16
    ya = pk.apply(x)                     # the automatic route at these row counts
    assert torch.equal(ya, y0)


def test_shared_rotation_wide_output_at_16_rows_runs_eight_tile_blocks(dev):
    """9..16 rows on a wide merged projection: the automatic route is mode 3 on 8-tile column blocks (4-tile blocks would not be resident
    at once; 16 rows x 8 tiles is built for this mode only) and returns the replicated rotation's bits (which runs 4-tile blocks: the
    column tiling does not enter the K summation order)."""
    from paroquant_amd import _native as nat, ops
    lib = nat.load()
    K, sizes = 2560, [9728, 9728]
    L = po.make_layer(77, K, sizes)
    pk = _packed(L, dev)
    d = ops.pk_desc(pk, torch.float16)
    out = [ctypes.c_int(v) for v in (0, 0, 0, -1)]
    nat.check(lib.paro_gemv_launch_shape(ctypes.byref(d), 16, *[ctypes.byref(o) for o in out]))
    assert (out[0].value, out[3].value) == (8, 3)
    x = _t(np.random.default_rng(4).standard_normal((16, K)).astype(np.float16), dev)
    y = pk.apply(x)
    assert torch.equal(y, ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0))
    rng = np.random.default_rng(8)
    cols = rng.choice(sum(sizes), size=96, replace=False)
    ref = _ideal(L, _np(x))[:, cols] if K * sum(sizes) <= (1 << 26) else None
    if ref is not None:
        assert po.rel_err(_np(y)[:, cols], ref) < TIGHT_F16


def test_shared_rotation_hybrid_forced_in_a_child_process(dev):
    """PARO_SHR_SELF=1 (read once per process) forces the hybrid form wherever mode 3 runs with at most 8 rows: a child process sweeps shapes
    x rows x types, comparing explicit mode 3 with mode 0 bit for bit and the status word; PARO_SHR_SELF=0 forces the pure form the same way."""
    import subprocess, sys, os, json
    code = r'''
import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import paro_oracle as po
from tests.test_gpu_parity import _packed, _t, _np
from paroquant_amd import ops
from paroquant_amd.linear import PackedParoWeights
dev = torch.device("cuda:0")
bad = []
n = 0
for K, sizes in [(2560, [4096, 1024, 1024]), (2560, [2432, 2432]), (9728, [640]), (4096, [2560]), (1536, [512]), (256, [48, 16]), (1024, [272])]:
    gs = 64 if K == 1024 else 128           # (one group_size-64 layer: two quantisation groups per rotation span)
    L = po.make_layer(K + 3, K, sizes, group_size=gs)
    pk = _packed(L, dev) if gs == 128 else PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev),
                                                             _t(L["pairs"], dev), _t(L["channel_scales"], dev), sizes, None, gs)
    for rows in (1, 2, 3, 4, 5, 7, 8):
        for dt in (torch.float16, torch.bfloat16):
            x = _t(np.random.default_rng(K + rows).standard_normal((rows, K)).astype(np.float32), dev, dt)
            y3 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 3)
            y0 = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0)
            n += 1
            if not torch.equal(y3, y0):
                bad.append((K, sizes, rows, str(dt)))
    ops.check_workspace(pk.workspace)
print(json.dumps({"cases": n, "bad": bad}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for force in ("1", "0"):
        env = dict(os.environ, PARO_SHR_SELF=force)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads(out.stdout.strip().splitlines()[-1])
        assert res["cases"] == 98 and not res["bad"], (force, res["bad"][:5])


def test_shared_rotation_is_what_apply_reaches_from_five_rows(dev):
    """The automatic route of the boundary: `PackedParoWeights.apply` (= RotateQuantizedLinear.forward / ParoQuantLinearMethod.apply) takes
    mode 3 from 5 rows on for the BASELINE decode shapes (from 3 rows in its hybrid form on mid-width / wide outputs), the replicated
    rotation below, and matches the oracle either way."""
    from paroquant_amd import _native as nat
    lib = nat.load()
    K, sizes = 2560, [4096, 1024, 1024]
    L = po.make_layer(5, K, sizes)
    pk = _packed(L, dev)
    from paroquant_amd import ops
    d = ops.pk_desc(pk, torch.float16)
    # (3..4 rows: mode 3 in its HYBRID form on this mid-width merged projection -- 384 tiles, 20 groups on 16 waves: four groups are nobody's first)
    for rows, want in ((1, 0), (2, 0), (4, 3), (5, 3), (8, 3), (16, 3)):
        out = [ctypes.c_int(v) for v in (0, 0, 0, -1)]
        nat.check(lib.paro_gemv_launch_shape(ctypes.byref(d), rows, *[ctypes.byref(o) for o in out]))
        assert out[3].value == want, (rows, out[3].value)
        x = _t(np.random.default_rng(rows).standard_normal((rows, K)).astype(np.float16), dev)
        y = pk.apply(x)
        assert po.rel_err(_np(y), _ideal(L, _np(x))) < TIGHT_F16
        # bit identity with the replicated rotation AT THE SAME LAUNCH SHAPE (mode 3 may pick another K-split than mode 0's rule tree -- at
        # 9..16 rows this projection runs unsplit 2-tile blocks -- and another K-split is another fp32 summation order)
        assert torch.equal(y, ops.w4a16_gemv_tuned(x, pk, out[0].value, out[1].value, out[2].value, 0))
    ops.check_workspace(pk.workspace)


def test_shared_rotation_fresh_inputs_launch_after_launch(dev):
    """ONE workspace, two layers of different shapes taking turns, a new x for every launch: a granule left by an earlier launch (another
    tag) must never be consumed -- eagerly and from a captured graph replayed with changing inputs."""
    from paroquant_amd import ops
    La, Lb = po.make_layer(1, 2560, [4096, 1024, 1024]), po.make_layer(2, 4096, [2560])
    pa, pb = _packed(La, dev), _packed(Lb, dev)
    pb.workspace = pa.workspace            # shared on purpose
    rows = 8
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    for it in range(40):
        xa = torch.randn(rows, 2560, device=dev, dtype=torch.float16, generator=gen)
        xb = torch.randn(rows, 4096, device=dev, dtype=torch.float16, generator=gen)
        ya, yb = ops.w4a16_gemv_tuned(xa, pa, 0, 0, 0, 3), ops.w4a16_gemv_tuned(xb, pb, 0, 0, 0, 3)
        assert torch.equal(ya, ops.w4a16_gemv_tuned(xa, pa, 0, 0, 0, 0)), it
        assert torch.equal(yb, ops.w4a16_gemv_tuned(xb, pb, 0, 0, 0, 0)), it
    # captured: the inputs are rewritten between replays
    xa = torch.randn(rows, 2560, device=dev, dtype=torch.float16, generator=gen)
    xb = torch.randn(rows, 4096, device=dev, dtype=torch.float16, generator=gen)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        ops.w4a16_gemv_tuned(xa, pa, 0, 0, 0, 3); ops.w4a16_gemv_tuned(xb, pb, 0, 0, 0, 3)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = []
        for _ in range(6):
            outs.append((ops.w4a16_gemv_tuned(xa, pa, 0, 0, 0, 3), ops.w4a16_gemv_tuned(xb, pb, 0, 0, 0, 3)))
    for it in range(10):
        xa.copy_(torch.randn(rows, 2560, device=dev, dtype=torch.float16, generator=gen))
        xb.copy_(torch.randn(rows, 4096, device=dev, dtype=torch.float16, generator=gen))
        g.replay()
        torch.cuda.synchronize(dev)
        ra, rb = ops.w4a16_gemv_tuned(xa, pa, 0, 0, 0, 0), ops.w4a16_gemv_tuned(xb, pb, 0, 0, 0, 0)
        for ya, yb in outs:
            assert torch.equal(ya, ra) and torch.equal(yb, rb), it
    ops.check_workspace(pa.workspace)


def test_shared_rotation_two_streams_take_turns_on_one_workspace(dev):
    """Two streams = two hardware queues whose dispatch ids run side by side: each launch's tag also carries its queue's address, so the
    granules one queue left are never taken for the other's launch with the same id.  The streams alternate (synchronised in between, as
    the workspace contract asks), every launch on a new input."""
    from paroquant_amd import ops
    L = po.make_layer(11, 1536, [768, 256])
    pk = _packed(L, dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    xs = [torch.randn(6, 1536, device=dev, dtype=torch.float16, generator=gen) for _ in range(60)]
    refs = [ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 0) for x in xs]
    torch.cuda.synchronize(dev)
    for i, x in enumerate(xs):
        st = s1 if i % 2 == 0 else s2
        with torch.cuda.stream(st):
            y = ops.w4a16_gemv_tuned(x, pk, 0, 0, 0, 3)
        st.synchronize()
        assert torch.equal(y, refs[i]), i
    ops.check_workspace(pk.workspace)


def test_shared_rotation_falls_back_when_the_grid_cannot_be_resident(dev):
    """Nobody may wait for a producer that is not running: a grid that does not fit the chip at once takes the replicated rotation (same
    bits), silently -- explicit mode 3 on a wide output with one tile per wave."""
    from paroquant_amd import ops
    K, sizes = 512, [16384]
    L = po.make_layer(21, K, sizes)
    pk = _packed(L, dev)
    x = _t(np.random.default_rng(2).standard_normal((8, K)).astype(np.float16), dev)
    y3 = ops.w4a16_gemv_tuned(x, pk, 1, 1, 8, 3)          # 1024 column blocks of 8 waves: not resident at once
    y0 = ops.w4a16_gemv_tuned(x, pk, 1, 1, 8, 0)
    assert torch.equal(y3, y0)
    assert po.rel_err(_np(y3), _ideal(L, _np(x))) < TIGHT_F16


def test_shared_rotation_refuses_fusions(dev):
    from paroquant_amd import _native as nat
    lib = nat.load()
    L = po.make_layer(31, 256, [64])
    pk = _packed(L, dev)
    from paroquant_amd import ops
    d = ops.pk_desc(pk, torch.float16)
    x = torch.zeros(2, 256, device=dev, dtype=torch.float16)
    y = torch.zeros(2, 64, device=dev, dtype=torch.float16)
    ws = pk.workspace
    # (the fused entry has no mode argument: it always runs mode 0; mode 3 with more than 16 rows becomes the pre-pass like mode 0 does)
    rc = lib.paro_w4a16_gemv(ctypes.byref(d), x.data_ptr(), y.data_ptr(), 2, ws.data_ptr(), ws.numel(), 0, 0, 0, 4, None)
    assert rc == -1 and b"mode must be" in lib.paro_last_error()

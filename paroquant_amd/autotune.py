"""Measured launch-shape selection for the batch-1 GEMV, once per distinct layer shape at load time.

The reference leaves the INT4 matmul's launch configuration to the third-party kernel it calls (AutoAWQ / Marlin behind
``ParoQuantLinearMethod.apply``, vllm/plugin.py:281-311); its one-time hook is ``process_weights_after_loading``
(plugin.py:251-279).  This library's GEMV has three launch knobs -- tiles per wave, K-slices, waves per workgroup -- and a rule
tree (csrc/gemv.hip ``gemv_autotune``) calibrated by sweeps over named models' shapes.  A shape no sweep has seen falls through to
whichever branch catches it, so the same one-time hook can MEASURE instead:

    pk.autotune()                      # PackedParoWeights; or PARO_AUTOTUNE=1 for every layer the plug-ins prepare

For each legal (tiles_per_wave, ksplit, waves) the rules can resolve to, ``launches`` graph-replayed launches over rotating copies
of the packed weights (more bytes than the Infinity Cache holds: the shape is chosen for weights that stream from HBM, as they do
inside a decoder step) are timed; the choice goes into ``paro_linear_t.launch_hint`` (include/paro_abi.h, v16) and one-row calls with
auto knobs use it.  The rule tree stays the cold default, and stays the choice unless a candidate beats it by ``min_gain``;
near-ties are broken by a fixed order, so that runs agree unless two shapes really are within the noise.  Results are cached per
(device, K, partition sizes, group size, tile order, dtype): a model's layers of one shape are measured once."""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _native as nat
from . import ops

_CACHE: Dict[tuple, dict] = {}

TPW = (1, 2, 4, 8)
KSPLIT = (1, 2, 3, 4, 6, 8)
WAVES = (4, 8, 16)


def launch_hint(tpw: int, ksplit: int, waves: int) -> int:
    """PARO_LAUNCH_HINT of include/paro_abi.h."""
    return (tpw & 0xff) | ((ksplit & 0xff) << 8) | ((waves & 0xff) << 16)


def enabled() -> bool:
    return os.environ.get("PARO_AUTOTUNE", "0") not in ("", "0")


def candidates(pk, dtype: torch.dtype = torch.float16) -> Tuple[Tuple[int, int, int], List[Tuple[int, int, int]]]:
    """(the rule tree's shape, every distinct shape the knobs resolve to) for one row -- host only, no launches."""
    lib = nat.load()
    d = ops.pk_desc(pk, dtype)
    d.launch_hint = 0

    def resolve(t, k, w):
        a, b, c, m = ctypes.c_int(t), ctypes.c_int(k), ctypes.c_int(w), ctypes.c_int(-1)
        if lib.paro_gemv_launch_shape(ctypes.byref(d), 1, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(m)) != 0:
            return None
        return (a.value, b.value, c.value)

    default = resolve(0, 0, 0)
    if default is None:
        nat.check(-1)                                      # (raises with the library's message)
    seen = {default}
    for t in TPW:
        for k in KSPLIT:
            for w in WAVES:
                r = resolve(t, k, w)
                if r is not None and r[0] == t and r[2] == w:       # (a clamped request is another candidate's shape)
                    seen.add(r)
    return default, sorted(seen, key=lambda s: (s[1], s[0], s[2]))


def _key(pk, dtype) -> tuple:
    dev = pk.wq.device
    name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else str(dev)
    return (name, pk.K, tuple(pk.partition_sizes), int(getattr(pk, "group_size", 128)), int(pk.wq_order), str(dtype))


def choose(default: Tuple[int, int, int], shapes: List[Tuple[int, int, int]], times: Dict[Tuple[int, int, int], float],
           min_gain: float = 0.02, tie: float = 0.01) -> Tuple[int, int, int]:
    """The selection rule on measured times: the rule tree's shape unless the best candidate is more than ``min_gain`` ahead of it; then
    the FIRST shape in the fixed candidate order within ``tie`` of the best (two runs agree unless shapes really are within the noise)."""
    t_best = min(times.values())
    if times[default] <= t_best * (1.0 + min_gain):
        return default
    return next(s for s in shapes if s in times and times[s] <= t_best * (1.0 + tie))


@torch.no_grad()
def measure(pk, shapes: List[Tuple[int, int, int]], dtype: torch.dtype = torch.float16, launches: int = 60, reps: int = 3,
            rotate_bytes: int = 320 << 20) -> Dict[Tuple[int, int, int], float]:
    """Microseconds per launch of every shape: a HIP graph of ``launches`` dependent one-row launches over rotating copies of the weights."""
    lib = nat.load()
    dev = pk.wq.device
    wbytes = pk.wq.numel() * pk.wq.element_size() + pk.sz.numel() * pk.sz.element_size()
    ncopy = int(max(2, min(48, math.ceil(rotate_bytes / max(wbytes, 1)))))
    wqs = [pk.wq] + [pk.wq.clone() for _ in range(ncopy - 1)]
    szs = [pk.sz] + [pk.sz.clone() for _ in range(ncopy - 1)]
    descs = []
    for wq, sz in zip(wqs, szs):
        d = ops.make_desc(pk.K, pk.partition_sizes, int(pk.pairs.size(1)), dtype, wq, sz, pk.rot, pk.pairs, pk.theta, pk.channel_scales, None,
                          pk.wq_order, group_size=int(getattr(pk, "group_size", 0)))
        descs.append(d)
    x = torch.randn(1, pk.K, device=dev, dtype=dtype)
    y = torch.empty(1, pk.N, device=dev, dtype=dtype)
    ws = pk.workspace
    need = max(lib.paro_linear_workspace_bytes(ctypes.byref(descs[0]), 1), 0)
    if ws.numel() * ws.element_size() < need:
        ws = ops.get_workspace(dev, need)
    out: Dict[Tuple[int, int, int], float] = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.device(dev):
        for shape in shapes:
            t, k, w = shape

            def burst():
                sp = nat.current_stream_ptr(dev)
                for i in range(launches):
                    nat.check(lib.paro_w4a16_gemv(ctypes.byref(descs[i % ncopy]), x.data_ptr(), y.data_ptr(), 1, ws.data_ptr(),
                                                  ws.numel() * ws.element_size(), t, k, w, -1, sp))
            try:
                s = torch.cuda.Stream(dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    burst()
                torch.cuda.current_stream(dev).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    burst()
                g.replay()
                torch.cuda.synchronize(dev)
                best = float("inf")
                for _ in range(reps):
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize(dev)
                    best = min(best, e0.elapsed_time(e1) * 1e3 / launches)
                out[shape] = best
            except RuntimeError:
                continue                                   # (a shape the library refuses at launch is not a candidate)
    return out


def autotune_packed(pk, dtype: torch.dtype = torch.float16, min_gain: float = 0.02, tie: float = 0.01, force: bool = False, **kw) -> dict:
    """Measure (or take from the cache) and set ``pk.launch_hint``.  Returns the report: the rule tree's shape and time, the choice
    and its time, every candidate's time."""
    key = _key(pk, dtype)
    rep = None if force else _CACHE.get(key)
    if rep is None:
        default, shapes = candidates(pk, dtype)
        times = measure(pk, shapes, dtype, **kw)
        if default not in times:
            raise RuntimeError("autotune: the rule tree's own launch shape failed to run")
        choice = choose(default, shapes, times, min_gain, tie)
        rep = {"default": list(default), "default_us": round(times[default], 3), "choice": list(choice), "choice_us": round(times[choice], 3),
               "candidates": {"%d,%d,%d" % s: round(t, 3) for s, t in sorted(times.items())}}
        _CACHE[key] = rep
    ch = tuple(rep["choice"])
    pk.launch_hint = 0 if ch == tuple(rep["default"]) else launch_hint(*ch)
    pk.autotune_report = rep
    return rep

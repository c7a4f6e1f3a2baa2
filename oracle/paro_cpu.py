"""ctypes wrapper of oracle/paro_cpu.c (TEST INFRASTRUCTURE / CPU BASELINE ONLY).

Loads ``oracle/_build/libparo_cpu.so`` (AVX2+FMA+F16C build) when the host CPU supports it, else the
generic build.  Used by tests (cross-check against the numpy oracle) and by bench.py's
``cpu_baseline`` leg; never by the product path."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _cpu_flags() -> set:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def load():
    global _lib
    if _lib is not None:
        return _lib
    flags = _cpu_flags()
    name = "libparo_cpu.so" if {"avx2", "fma", "f16c"} <= flags else "libparo_cpu_generic.so"
    path = os.path.join(_HERE, "_build", name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built; run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.paro_cpu_threads.restype = i32
    lib.paro_cpu_set_threads.restype = None
    lib.paro_cpu_set_threads.argtypes = [i32]
    lib.paro_cpu_has_f16c.restype = i32
    lib.paro_cpu_rotate_f16.restype = None
    lib.paro_cpu_rotate_f16.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32]
    lib.paro_cpu_linear_f16.restype = None
    lib.paro_cpu_linear_f16.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32]
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def threads() -> int:
    return int(load().paro_cpu_threads())


def set_threads(n: int) -> None:
    """Cap the OpenMP team of the C port (the CPU-baseline leg of bench.py)."""
    load().paro_cpu_set_threads(int(n))


def rotate_f16(x, idx_ij, theta, scales=None, group_size: int = 128) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float16)
    idx_ij = np.ascontiguousarray(idx_ij, dtype=np.int16)
    theta = np.ascontiguousarray(theta, dtype=np.float16)
    sc = None if scales is None else np.ascontiguousarray(np.asarray(scales).reshape(-1), dtype=np.float16)
    H = x.shape[-1]
    out = np.empty_like(x)
    load().paro_cpu_rotate_f16(_p(x), _p(out), _p(idx_ij), _p(theta), _p(sc), x.size // H, H, idx_ij.shape[0],
                               group_size)
    return out


def linear_f16(x, L, bias=None) -> np.ndarray:
    """``L`` = layer dict in checkpoint format (see paro_oracle.make_layer)."""
    x = np.ascontiguousarray(x, dtype=np.float16)
    K = x.shape[-1]
    rows = x.size // K
    sizes = np.ascontiguousarray(L["sizes"], dtype=np.int32)
    N = int(sizes.sum())
    qw = np.ascontiguousarray(L["qweight"], dtype=np.int32)
    qz = np.ascontiguousarray(L["qzeros"], dtype=np.int32)
    sc = np.ascontiguousarray(L["scales"], dtype=np.float16)
    pairs = np.ascontiguousarray(L["pairs"], dtype=np.int16)
    theta = np.ascontiguousarray(L["theta"], dtype=np.float16)
    cs = np.ascontiguousarray(np.asarray(L["channel_scales"]).reshape(len(sizes), K), dtype=np.float16)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float16)
    y = np.empty((rows, N), dtype=np.float16)
    load().paro_cpu_linear_f16(_p(x), _p(y), rows, K, N, _p(qw), _p(qz), _p(sc), len(sizes), _p(sizes), _p(pairs),
                               _p(theta), _p(cs), _p(b), pairs.shape[1])
    return y.reshape(*x.shape[:-1], N)

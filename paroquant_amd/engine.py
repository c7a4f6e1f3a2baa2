"""``DecodeEngine`` -- a chain of ParoQuant linears at batch 1 in ONE persistent launch (``paro_engine_*``, csrc/experimental/engine.hip).

EXPERIMENTAL (round 6): both engine builds are parity-green and measured slower than one launch per linear on every shape this repo
times (profiles/NOTES.md 4.2, 5.1), so they left the default library -- this class needs ``make -C paroquant_amd/csrc EXPERIMENTAL=1``
(include/paro_abi_experimental.h) and raises otherwise.  Nothing in the product path constructs it.

The reference runs ``rotate -> INT4 GEMM`` per linear (``transformers/modules.py:57-71``, ``vllm/plugin.py:281-311``); an HF MLP block is
three such linears plus the activation (``down(act(gate(x)) * up(x))``).  At one row those are dependent launches of a few microseconds
each; the engine keeps one resident grid that requests the next linear's INT4 tiles while the current linear's outputs are handed over,
and rotates every 128-channel group once per partition.

    eng = DecodeEngine([pk_a, pk_b, ...], in_col0=[0, c1, ...])      # linear i + 1 reads columns c .. c + K of linear i's output
    y = eng(x)                                                        # x [1, K_0] fp16 / bf16  ->  y [1, N_last]

The packed weights (``PackedParoWeights``) must stay alive while the engine is (the plan holds their device pointers)."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native as nat
from . import ops


class DecodeEngine:
    def __init__(self, layers: Sequence, in_col0: Optional[Sequence[int]] = None, dtype: torch.dtype = torch.float16, n_cus: int = 0,
                 version: int = 0, split: Optional[Sequence[int]] = None):
        """``version`` 1: the sixteen-wave build (csrc/engine.hip, ``paro_engine_*``) -- the default: it is the faster one on every measured
        model (profiles/NOTES.md, round 5); 2: the loader / consumer build on the LDS-DMA ring (csrc/engine2.hip, ``paro_engine2_*``);
        0: ``PARO_ENGINE_VERSION`` or the default.  ``split`` (version 2): K-chunks per linear, 0 = the planner's."""
        import os
        lib = nat.load()
        if not nat.has_experimental():
            raise RuntimeError("DecodeEngine needs a library built with `make -C paroquant_amd/csrc EXPERIMENTAL=1` "
                               "(the persistent engines are not part of the default libparo_mi355x.so: include/paro_abi_experimental.h)")
        self.version = int(version) or int(os.environ.get("PARO_ENGINE_VERSION", "1"))
        if self.version not in (1, 2):
            raise ValueError("engine version must be 1 or 2")
        pre = "paro_engine_" if self.version == 1 else "paro_engine2_"
        self._fn = {k: getattr(lib, pre + k) for k in ("plan", "build", "describe", "run", "trace")}
        self._trace_words = 32 if self.version == 1 else 64
        if not layers:
            raise ValueError("DecodeEngine needs at least one linear")
        self.layers = list(layers)
        dev = self.layers[0].wq.device
        if dev.type != "cuda":
            raise RuntimeError("DecodeEngine needs the packed weights on a GPU (there is no CPU path)")
        self.device, self.dtype = dev, dtype
        n = len(self.layers)
        in_col0 = [0] * n if in_col0 is None else [int(c) for c in in_col0]
        if len(in_col0) != n:
            raise ValueError("in_col0 needs one entry per linear")
        self._descs = [ops.make_desc(pk.K, pk.partition_sizes, int(pk.pairs.size(1)), dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                                     pk.channel_scales, (pk.bias.to(dtype) if pk.bias is not None else None), pk.wq_order,
                                     group_size=pk.group_size) for pk in self.layers]
        self._bias = [pk.bias.to(dtype) if pk.bias is not None else None for pk in self.layers]   # (keeps converted biases alive)
        for d, b in zip(self._descs, self._bias):
            d.bias = b.data_ptr() if b is not None else None
        self._phases = (nat.ParoEnginePhase * n)()
        for i, d in enumerate(self._descs):
            self._phases[i].L = ctypes.pointer(d)
            self._phases[i].in_col0 = in_col0[i]
            self._phases[i].flags = int(split[i]) if (split is not None and self.version == 2) else 0
        self._e = nat.ParoEngine()
        with torch.cuda.device(dev):
            nat.check(self._fn["plan"](self._phases, n, int(n_cus), ctypes.byref(self._e)))
            host = np.zeros(int(self._e.plan_bytes), dtype=np.uint8)
            nat.check(self._fn["build"](self._phases, ctypes.byref(self._e), host.ctypes.data_as(ctypes.c_void_p)))
        self.plan = torch.from_numpy(host).to(dev)
        self.workspace = torch.zeros(int(self._e.workspace_bytes), dtype=torch.uint8, device=dev)     # zero-filled ONCE
        self.K, self.N = int(self._e.in_features), int(self._e.out_features)
        self.y = torch.empty(1, self.N, dtype=dtype, device=dev)

    def describe(self):
        """[(K-chunks, most tiles per CU, fewest tiles per CU)] per phase: what the planner chose."""
        lib, out = nat.load(), []
        for i in range(len(self.layers)):
            s, mx, mn = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            nat.check(self._fn["describe"](self._phases, ctypes.byref(self._e), i, ctypes.byref(s), ctypes.byref(mx), ctypes.byref(mn)))
            out.append((s.value, mx.value, mn.value))
        return out

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dtype != self.dtype or x.numel() != self.K or not x.is_contiguous() or x.device != self.device:
            raise ValueError(f"DecodeEngine expects one contiguous row of {self.K} {self.dtype} activations on {self.device}")
        y = self.y if out is None else out
        if y.dtype != self.dtype or y.numel() != self.N or not y.is_contiguous():
            raise ValueError(f"out must be a contiguous row of {self.N} {self.dtype} values")
        with torch.cuda.device(self.device):
            nat.check(self._fn["run"](ctypes.byref(self._e), self.plan.data_ptr(), x.data_ptr(), y.data_ptr(),
                                                 self.workspace.data_ptr(), self.workspace.numel(), nat.current_stream_ptr(self.device)))
        return y

    @torch.no_grad()
    def trace(self, x: torch.Tensor) -> torch.Tensor:
        """One launch of the diagnostic twin (``paro_engine_trace``): int64 [n_phases, n_cus, 32] stamps (16 events: 100 MHz counter, then the shader clock at the same events) of the 100 MHz counter."""
        tr = torch.zeros(len(self.layers), int(self._e.n_cus), self._trace_words, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._fn["trace"](ctypes.byref(self._e), self.plan.data_ptr(), x.data_ptr(), self.y.data_ptr(),
                                                  self.workspace.data_ptr(), self.workspace.numel(), tr.data_ptr(),
                                                  nat.current_stream_ptr(self.device)))
        torch.cuda.synchronize(self.device)
        return tr

    def status_ok(self) -> bool:
        """False once a hand-off inside a launch gave up waiting (outputs of that launch are NaN).  Synchronises."""
        return int(self.workspace[4:8].view(torch.int32).item()) == 0

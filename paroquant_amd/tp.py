"""Tensor-parallel sharding of a ParoQuant linear (reference
``vllm/plugin.py:33-50,196-198``).

One process per GPU; the only collective on the path is the all-reduce(SUM) after a row-parallel
linear, issued through ``torch.distributed`` (backend ``nccl`` == RCCL over xGMI on MI355X, ``gloo``
in the CPU tests).  Column-parallel layers shard N and replicate the rotation parameters;
row-parallel layers shard K in multiples of 128 (the rotation is block-diagonal per 128-channel
group, so a rank rotates only its own K-slice) and narrow the rotation parameters along the input
dim by ``tp_rank`` exactly as ``_maybe_shard_input`` does.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import ctypes
import torch
import torch.distributed as dist


def shard_column_parallel(layer: Dict[str, torch.Tensor], sizes: Sequence[int], rank: int, world: int):
    """Shard every merged partition's columns; rotation params are replicated (full K)."""
    qw_parts, qz_parts, sc_parts, new_sizes = [], [], [], []
    col = 0
    for n in sizes:
        if n % (world * 16) != 0:
            raise ValueError(f"partition of {n} columns cannot be split {world}-way in multiples of 16")
        per = n // world
        lo = col + rank * per
        qw_parts.append(layer["qweight"][:, lo // 8:(lo + per) // 8])
        qz_parts.append(layer["qzeros"][:, lo // 8:(lo + per) // 8])
        sc_parts.append(layer["scales"][:, lo:lo + per])
        new_sizes.append(per)
        col += n
    out = dict(layer)
    out["qweight"] = torch.cat(qw_parts, dim=1).contiguous()
    out["qzeros"] = torch.cat(qz_parts, dim=1).contiguous()
    out["scales"] = torch.cat(sc_parts, dim=1).contiguous()
    out["sizes"] = new_sizes
    if layer.get("bias") is not None:
        bs, col = [], 0
        for n in sizes:
            per = n // world
            bs.append(layer["bias"][col + rank * per: col + (rank + 1) * per])
            col += n
        out["bias"] = torch.cat(bs).contiguous()
    return out


def shard_row_parallel(layer: Dict[str, torch.Tensor], rank: int, world: int, group_size: int = 128):
    """Shard K (rows of qweight, group-rows of scales/qzeros) and narrow the rotation params
    (``loaded_weight.narrow(-1, tp_rank * shard, shard)``, plugin.py:47-50)."""
    K = layer["qweight"].shape[0]
    group_size = K // layer["qzeros"].shape[0] if layer["qzeros"].shape[0] else group_size   # what the tensors say
    # a shard must hold whole ROTATION groups (128 channels at inference whatever the quantisation group is)
    if K % (world * max(group_size, 128)) != 0:
        raise ValueError(f"in_features {K} cannot be split {world}-way in multiples of {max(group_size, 128)}")
    Kp = K // world
    g0, g1 = rank * Kp // group_size, (rank + 1) * Kp // group_size
    out = dict(layer)
    out["qweight"] = layer["qweight"][rank * Kp:(rank + 1) * Kp].contiguous()
    out["qzeros"] = layer["qzeros"][g0:g1].contiguous()
    out["scales"] = layer["scales"][g0:g1].contiguous()
    out["pairs"] = layer["pairs"].narrow(-1, rank * Kp, Kp).contiguous()
    out["theta"] = layer["theta"].narrow(-1, rank * Kp // 2, Kp // 2).contiguous()
    out["channel_scales"] = layer["channel_scales"].narrow(-1, rank * Kp, Kp).contiguous()
    # the bias is added once, after the reduction (rank 0 keeps it)
    if layer.get("bias") is not None and rank != 0:
        out["bias"] = None
    return out


def row_parallel_forward(apply_fn, x_full: torch.Tensor, rank: int, world: int, group=None) -> torch.Tensor:
    """``y = all_reduce_sum(apply_fn(x[..., rank-th K slice]))`` -- RowParallelLinear semantics."""
    K = x_full.shape[-1]
    Kp = K // world
    y = apply_fn(x_full[..., rank * Kp:(rank + 1) * Kp].contiguous())
    if world > 1:
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    return y


class OneShotAllReduce:
    """All-reduce(SUM) of the row-parallel linears' small decode outputs in ONE kernel launch per rank
    (``paro_allreduce_oneshot``, csrc/allreduce.hip): every rank stores its vector -- pairs of activations tagged with
    the call's epoch, 8 bytes at a time -- straight into a slot of every peer's buffer over xGMI, polls the granules of
    its own buffer until their tags read the epoch, and sums in rank order.  No host work per call, so a tensor-parallel
    decode step stays one HIP graph whatever the collective library can or cannot capture.  The same object also feeds
    the row-parallel GEMV's all-reduce EPILOGUE (``ops.w4a16_gemv_fused(..., allreduce=self)``, :meth:`fusion_args`):
    the exchange then runs inside the GEMV launch and the separate kernel disappears.

    The per-rank buffers are FINE-GRAINED device memory allocated by the library (peers write into them and the owner
    polls them inside one kernel: ordinary device memory is coherent across GPUs only at kernel boundaries) and exchanged
    once as HIP IPC handles over ``group`` (any backend that can ``all_gather_object``: nccl or gloo).  ``self_test``
    compares a few calls against ``dist.all_reduce``; :func:`make_allreduce` falls back to the library collective when the
    exchange or the test fails."""

    def __init__(self, device, max_elems: int, group=None):
        import ctypes
        from . import _native as nat
        self.lib = nat.load()
        self.nat = nat
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        self.max_elems = int(max_elems)
        self._own, self._opened = None, []
        nbytes = self.lib.paro_allreduce_buffer_bytes(self.world, self.max_elems)
        if nbytes < 0:
            raise ValueError(f"one-shot all-reduce supports up to 16 ranks and >= 8 elements (world {self.world}, {max_elems} elements)")
        handle, err = None, None
        with torch.cuda.device(self.device):
            try:
                own = ctypes.c_void_p()
                hbuf = ctypes.create_string_buffer(64)
                nat.check(self.lib.paro_allreduce_buffer_create(nbytes, ctypes.byref(own), hbuf))
                self._own, handle = own.value, hbuf.raw
            except Exception as e:
                err = e
            # every rank runs the same collectives whatever happened locally: exchange, agree, then raise together
            gathered = [None] * self.world
            dist.all_gather_object(gathered, handle, group=group)
            ptrs = []
            if err is None and all(h is not None for h in gathered):
                try:
                    for r in range(self.world):
                        if r == self.rank:
                            ptrs.append(self._own)
                        else:
                            p = ctypes.c_void_p()
                            nat.check(self.lib.paro_allreduce_buffer_open(gathered[r], ctypes.byref(p)))
                            self._opened.append(p.value)
                            ptrs.append(p.value)
                    self.peers = (ctypes.c_void_p * self.world)(*ptrs)     # HOST array: the library copies it into the launch arguments
                    # launch state of the GEMV's all-reduce epilogue (give-up flag, one epoch per 16-column tile): ordinary cached memory
                    self._gemv_state = torch.zeros(16 + (self.max_elems + 15) // 16, dtype=torch.int32, device=self.device)
                    torch.cuda.synchronize(self.device)
                except Exception as e:                        # e.g. IPC mapping refused on this rank
                    err = e
            elif err is None:
                err = RuntimeError("a peer could not allocate its buffer")
        if not self._agree(err is None):
            self.close()
            raise RuntimeError(f"peer buffers could not be set up on every rank (rank {self.rank}: {err})")
        dist.barrier(group=group)                             # every buffer is zeroed and mapped before the first store lands

    def _agree(self, ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def close(self) -> None:
        """Unmap the peers' buffers and free the own one (after a barrier of the caller's: peers may still be storing)."""
        with torch.cuda.device(self.device):
            for p in self._opened:
                self.lib.paro_allreduce_buffer_close(p)
            self._opened = []
            if self._own:
                self.lib.paro_allreduce_buffer_destroy(self._own)
                self._own = None

    def __call__(self, y: torch.Tensor, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``out <- sum over ranks of y (+ residual)``; ``out`` defaults to ``y`` (in place, like ``dist.all_reduce``)."""
        n = y.numel()
        out = y if out is None else out
        if not y.is_contiguous() or y.dtype not in (torch.float16, torch.bfloat16) or n % 8 or n > self.max_elems:
            raise ValueError(f"one-shot all-reduce takes contiguous fp16 / bf16 vectors of a multiple of 8 up to {self.max_elems} elements")
        if out.numel() != n or out.dtype != y.dtype or not out.is_contiguous() or \
                (residual is not None and (residual.numel() != n or residual.dtype != y.dtype or not residual.is_contiguous())):
            raise ValueError("residual / out must match y (contiguous, same dtype and size)")
        with torch.cuda.device(y.device):
            self.nat.check(self.lib.paro_allreduce_oneshot(y.data_ptr(), None if residual is None else residual.data_ptr(), out.data_ptr(), n,
                                                           self.nat.dtype_code(y.dtype), ctypes.cast(self.peers, ctypes.c_void_p), self.world, self.rank,
                                                           self.max_elems, self.nat.current_stream_ptr(y.device)))
        return out

    def fusion_args(self):
        """(peers, own, state, world, rank, max_elems) for ``paro_fusion_t``'s all-reduce epilogue (``ops.w4a16_gemv_fused(...,
        allreduce=self)``): the row-parallel GEMV exchanges its partial outputs itself, no separate launch."""
        return ctypes.cast(self.peers, ctypes.c_void_p), self._own, self._gemv_state.data_ptr(), self.world, self.rank, self.max_elems

    def gave_up(self) -> bool:
        """True when a call timed out waiting for a peer (sticky status word; synchronises the current stream)."""
        with torch.cuda.device(self.device):
            return self.lib.paro_allreduce_status(self._own, self.nat.current_stream_ptr(self.device)) != 0

    def self_test(self, iters: int = 4) -> bool:
        """A few calls on seeded per-rank data against the library collective; every rank returns the same verdict."""
        ok = True
        n = min(self.max_elems, 4096) // 8 * 8
        backend = dist.get_backend(self.group)
        for i in range(iters):
            g = torch.Generator(device="cpu").manual_seed(1234 + 17 * i + self.rank)
            x = (torch.randn(n, generator=g) * 0.5).to(torch.float16)
            ref = x.float().to(self.device) if backend == "nccl" else x.float()
            dist.all_reduce(ref, group=self.group)           # the same collective sequence on every rank, whatever fails locally
            try:
                got = self(x.to(self.device).clone()).float().cpu()
                ok = ok and bool(torch.isfinite(got).all()) and float((got - ref.cpu()).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
            except Exception:
                ok = False
        try:
            ok = ok and not self.gave_up()
        except Exception:
            ok = False
        return self._agree(ok)


def make_allreduce(device, max_elems: int, group=None, prefer_oneshot: bool = True):
    """``(fn, name)``: the in-place all-reduce a tensor-parallel decode step should use -- the one-shot kernel when it can
    be set up and passes its self-test on every rank, else ``dist.all_reduce`` (RCCL under the nccl backend).
    ``PARO_ONESHOT=0`` in the environment forces the library collective (operator override; set it on every rank)."""
    import os
    if os.environ.get("PARO_ONESHOT", "1") == "0":
        prefer_oneshot = False

    def _library(y, residual=None, out=None):
        dist.all_reduce(y, group=group)
        if residual is not None:
            y = torch.add(y, residual, out=out if out is not None else y)
        elif out is not None and out is not y:
            out.copy_(y)
            y = out
        return y
    _library.group = group          # ParoDecoderLM reads it for the prefill's [T, hidden] all-reduce (same as OneShotAllReduce.group)
    fallback = _library, dist.get_backend(group)
    if not prefer_oneshot or dist.get_world_size(group) == 1:
        return fallback
    try:
        ar = OneShotAllReduce(device, max_elems, group)      # raises on EVERY rank or on none (the ranks agree inside)
    except Exception as e:
        if dist.get_rank(group) == 0:
            print(f"[paroquant_amd.tp] one-shot all-reduce unavailable ({type(e).__name__}: {e}); using {fallback[1]}", flush=True)
        return fallback
    if ar.self_test():
        return ar, "oneshot"
    if dist.get_rank(group) == 0:
        print(f"[paroquant_amd.tp] one-shot all-reduce failed its self-test; using {fallback[1]}", flush=True)
    try:
        dist.barrier(group=group)   # no peer may still be storing into a buffer that is about to be unmapped
        ar.close()                  # the IPC-mapped fine-grained buffers would otherwise leak on every rank
    except Exception:
        pass
    return fallback

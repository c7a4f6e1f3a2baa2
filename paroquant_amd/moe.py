"""Mixture-of-experts experts on the fused kernels.

The reference exports MoE experts with ONE rotation per projection shared by all experts
(``cli/convert.py:280-379``: ``{base}.{e}.{gate,up,down}_proj.{qweight,qzeros,scales}`` per expert plus
``{base}.gate_up_weight_{theta,pairs,channel_scales}`` / ``{base}.down_weight_*``) and runs them in its MLX back-end as
``RotateSwitchGLU`` (``mlx/modules.py:159-212``): rotate x once, gate / up of the routed experts, ``silu(gate) * up``,
rotate the activation with the down rotation, down of the same experts -- returning the per-(token, expert) outputs;
the router's weights are applied by the surrounding MoE block.

Here, for decode-sized inputs (few tokens) the routed experts of all tokens are TWO launches: the merged gate|up
projection of every (token, expert) slot, then the down projection with the SiLU*mul prologue -- the fused GEMV with
``blockIdx.z`` = slot and the slot's expert id read from device memory (``paro_w4a16_gemv_experts``), the shared
rotation packed once.  Larger token counts rotate the tokens once, sort the (token, expert) pairs by expert on the device
and run ONE grouped W4A16 GEMM per projection over the expert segments (``paro_w4a16_gemm_grouped``), no host round trip.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _native as nat
from . import ops
from .linear import PackedParoWeights

_DECODE_SLOTS = 64      # (tokens x experts-per-token) up to which the slot kernels are used
_DEBUG_CHECKS = os.environ.get("PARO_DEBUG_CHECKS", "0") == "1"


class ParoMoEExperts:
    """The routed experts of one MoE layer.  ``tensors``: the checkpoint tensors under the experts' base prefix --
    ``"{e}.gate_proj.qweight"`` ..., ``"gate_up_weight_theta"``, ``"down_weight_pairs"`` ... (cli/convert.py:381-405)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], num_experts: int, device):
        dev = torch.device(device)
        g = lambda k: tensors[k].to(dev)
        self.E = int(num_experts)
        self.H = int(tensors["0.gate_proj.qweight"].shape[0])
        self.I = int(tensors["0.down_proj.qweight"].shape[0])
        self.device = dev
        self.gate_up, self.down = [], []
        gu_rot = (g("gate_up_weight_theta"), g("gate_up_weight_pairs"), g("gate_up_weight_channel_scales"))
        dn_rot = (g("down_weight_theta"), g("down_weight_pairs"), g("down_weight_channel_scales"))
        for e in range(self.E):
            qw = torch.cat([g(f"{e}.gate_proj.qweight"), g(f"{e}.up_proj.qweight")], dim=1)
            qz = torch.cat([g(f"{e}.gate_proj.qzeros"), g(f"{e}.up_proj.qzeros")], dim=1)
            sc = torch.cat([g(f"{e}.gate_proj.scales"), g(f"{e}.up_proj.scales")], dim=1)
            # gate|up is ONE rotation partition of 2 I columns (both halves see the same rotated x)
            self.gate_up.append(PackedParoWeights(qw, qz, sc, *gu_rot, [2 * self.I], wq_order=0))
            self.down.append(PackedParoWeights(g(f"{e}.down_proj.qweight"), g(f"{e}.down_proj.qzeros"), g(f"{e}.down_proj.scales"),
                                               *dn_rot, [self.H], wq_order=0))
        # the slot kernels index experts with a uniform byte stride: stack the packed buffers
        self.gu_wq = torch.stack([p.wq for p in self.gate_up]).contiguous()
        self.gu_sz = torch.stack([p.sz for p in self.gate_up]).contiguous()
        self.dn_wq = torch.stack([p.wq for p in self.down]).contiguous()
        self.dn_sz = torch.stack([p.sz for p in self.down]).contiguous()
        for e in range(self.E):     # the per-expert views now alias the stacks (one copy of the weights)
            self.gate_up[e].wq, self.gate_up[e].sz = self.gu_wq[e], self.gu_sz[e]
            self.down[e].wq, self.down[e].sz = self.dn_wq[e], self.dn_sz[e]

    # ------------------------------------------------------------------ decode: two launches for all slots
    def _slots(self, pk0: PackedParoWeights, wq: torch.Tensor, sz: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
               idx: torch.Tensor, x_div: int, prologue: int) -> None:
        lib = nat.load()
        d = ops.make_desc(pk0.K, pk0.partition_sizes, int(pk0.pairs.size(1)), x.dtype, wq, sz, pk0.rot, pk0.pairs, pk0.theta,
                          pk0.channel_scales, None, 0, group_size=pk0.group_size)
        f = nat.ParoFusion()
        f.prologue, f.eps, f.x_stride, f.residual = int(prologue), 0.0, 0, None
        e = nat.ParoExperts()
        e.expert_idx, e.n_slots, e.x_slot_div = idx.data_ptr(), int(idx.numel()), int(x_div)
        e.wq_stride_bytes, e.sz_stride_bytes = wq.stride(0) * 4, sz.stride(0) * 4
        e.x_slot_stride, e.y_slot_stride = x.stride(0), y.stride(0)
        e.n_experts = self.E
        ws = pk0.workspace
        with torch.cuda.device(x.device):
            nat.check(lib.paro_w4a16_gemv_experts(ctypes.byref(d), x.data_ptr(), y.data_ptr(), 1, ws.data_ptr(),
                                                  ws.numel(), ctypes.byref(f), ctypes.byref(e), nat.current_stream_ptr(x.device)))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        """x [T, H] (fp16 / bf16), indices [T, k] expert ids -> [T, k, H] per-slot expert outputs."""
        T, k = indices.shape
        x = x.reshape(T, self.H).contiguous()
        out = torch.empty(T, k, self.H, dtype=x.dtype, device=x.device)
        # Expert ids are validated ON THE DEVICE (paro_experts_t.n_experts, paro_w4a16_gemm_grouped's n_experts: an id outside [0, E) never
        # reads out of bounds and its outputs are NaN) -- no device->host sync per MoE block, and the same guarantee under HIP-graph replay,
        # where a host check cannot run.  PARO_DEBUG_CHECKS=1 adds the eager host check (IndexError) for debugging.
        if _DEBUG_CHECKS and not torch.cuda.is_current_stream_capturing() and bool((indices.min() < 0) | (indices.max() >= self.E)):
            raise IndexError(f"expert indices must lie in [0, {self.E})")
        if T * k <= _DECODE_SLOTS:
            idx = indices.reshape(-1).to(torch.int32).contiguous()
            gu = torch.empty(T * k, 2 * self.I, dtype=x.dtype, device=x.device)
            self._slots(self.gate_up[0], self.gu_wq, self.gu_sz, x, gu, idx, k, nat.PROLOGUE_NONE)
            self._slots(self.down[0], self.dn_wq, self.dn_sz, gu, out.view(T * k, self.H), idx, 1, nat.PROLOGUE_SILU_MUL)
            return out
        return self._grouped_prefill(x, indices, out)

    def _grouped_prefill(self, x: torch.Tensor, indices: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """Prefill: rotate the tokens ONCE (the experts share the rotation), sort the (token, expert) pairs by expert on the
        device, one grouped W4A16 GEMM over the expert segments (``paro_w4a16_gemm_grouped``), SiLU*mul and the down rotation
        once over all rows, a second grouped GEMM, un-sort.  Nothing is read back to the host (shapes depend on T, k, E
        only), so the whole sequence is HIP-graph capturable."""
        lib = nat.load()
        T, k = indices.shape
        S, E, dev, dt = T * k, self.E, x.device, x.dtype
        BM = 64 if S // E < 96 else 128                              # row block of the grouped GEMM (static: shapes only)
        max_rows = (S + (E + 1) * (BM - 1) + BM - 1) // BM * BM      # every expert's segment (+ the bucket of invalid ids) padded up to a block
        raw = indices.reshape(-1).to(torch.int64)
        bad = (raw < 0) | (raw >= E)
        flat = torch.where(bad, torch.full_like(raw, E), raw)        # ids outside [0, E): bucket E -- the kernel skips its blocks (n_experts), NaN below
        counts = torch.zeros(E + 1, dtype=torch.int64, device=dev).scatter_add_(0, flat, torch.ones_like(flat))
        padded = (counts + (BM - 1)) // BM * BM
        pend = torch.cumsum(padded, 0)
        pstart, ustart = pend - padded, torch.cumsum(counts, 0) - counts
        order = torch.argsort(flat, stable=True)
        es = flat[order]
        dest = pstart[es] + (torch.arange(S, device=dev) - ustart[es])      # row of sorted pair j in the padded buffer
        blocks = torch.arange(max_rows // BM, device=dev, dtype=torch.int64) * BM
        block_expert = torch.where(blocks < pend[-1], torch.searchsorted(pend, blocks, right=True), torch.full_like(blocks, -1)).to(torch.int32)
        gu0, dn0 = self.gate_up[0], self.down[0]
        xr = torch.ops.rotation.rotate(x, gu0.pairs[0], gu0.theta[0], gu0.channel_scales[0], 128)       # [T, H], once
        xs = torch.zeros(max_rows, self.H, dtype=dt, device=dev)
        xs.index_copy_(0, dest, xr.index_select(0, order // k))
        gu = torch.empty(max_rows, 2 * self.I, dtype=dt, device=dev)

        def grouped(pk0, wq, sz, xin, yout):
            d = ops.make_desc(pk0.K, pk0.partition_sizes, int(pk0.pairs.size(1)), dt, wq, sz, pk0.rot, pk0.pairs, pk0.theta,
                              pk0.channel_scales, None, 0, group_size=pk0.group_size)
            with torch.cuda.device(dev):
                nat.check(lib.paro_w4a16_gemm_grouped(ctypes.byref(d), xin.data_ptr(), yout.data_ptr(), max_rows, BM, block_expert.data_ptr(),
                                                      wq.stride(0) * 4, sz.stride(0) * 4, E, nat.current_stream_ptr(dev)))
        grouped(gu0, self.gu_wq, self.gu_sz, xs, gu)
        act = torch.nn.functional.silu(gu[:, :self.I]) * gu[:, self.I:]
        ar = torch.ops.rotation.rotate(act.contiguous(), dn0.pairs[0], dn0.theta[0], dn0.channel_scales[0], 128)
        yd = torch.empty(max_rows, self.H, dtype=dt, device=dev)
        grouped(dn0, self.dn_wq, self.dn_sz, ar, yd)
        out.view(S, self.H).index_copy_(0, order, yd.index_select(0, dest))
        out.view(S, self.H).masked_fill_(bad.unsqueeze(1), float("nan"))      # invalid ids: loud, like the decode slots
        return out

    def per_expert_prefill(self, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        """The round-2 prefill route, kept for A/B timing (tools/bench_moe.py): a host loop over the experts (one
        device->host read of the counts, E x (rotate pre-pass + 2 GEMMs))."""
        T, k = indices.shape
        x = x.reshape(T, self.H).contiguous()
        out = torch.empty(T, k, self.H, dtype=x.dtype, device=x.device)
        flat = indices.reshape(-1)
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=self.E).tolist()
        tok = (order // k)
        outf = out.view(T * k, self.H)
        start = 0
        for e, n in enumerate(counts):
            if n == 0:
                continue
            sel = order[start:start + n]
            xe = x[tok[start:start + n]]
            gu = self.gate_up[e].apply(xe)
            act = torch.nn.functional.silu(gu[:, :self.I]) * gu[:, self.I:]
            outf[sel] = self.down[e].apply(act.contiguous())
            start += n
        return out

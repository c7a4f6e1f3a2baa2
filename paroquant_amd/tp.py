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

from typing import Dict, Sequence

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

"""Checkpoint packer / repacker as a library -- the inference-side counterpart of the
reference's exporter, on GPU tensors:

    pack_awq / unpack_awq            cli/convert.py:19,149-155   (nibble order 0,2,4,6,1,3,5,7)
    to_awq_buffers                   cli/convert.py:194-203      ([N, K] quantised -> qweight / qzeros / scales)
    quantize_rotated_weight          cli/convert.py:158-191      (rotate W * cs in fp32, quantise with learned scale / zp)
    quantize_layer                   cli/convert.py:239-277      (optimiser state dict -> checkpoint tensors of one linear)

plus what only this build has: `save_prepacked` / `load_prepacked` write and read a linear already in the CDNA4
kernel layout (tiles in MFMA fragment order, packed scale/zero words, the rotation exchange schedule), so that
load-time `paro_repack_awq` / `paro_pack_rotation` can be skipped.  The rotation goes through
`torch.ops.rotation.rotate` (HIP); everything else is integer / elementwise torch on the same device.
Bit-exactness against the reference-generated goldens G1 / G2 / G5 / G6 is tested in tests/test_gpu_parity.py.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import ops  # noqa: F401  (registers rotation::rotate)
from .linear import PackedParoWeights

AWQ_REORDER = (0, 2, 4, 6, 1, 3, 5, 7)          # cli/convert.py:19
AWQ_INV_REORDER = (0, 4, 1, 5, 2, 6, 3, 7)      # mlx/load.py:18


def pack_awq(values: torch.Tensor) -> torch.Tensor:
    """int [rows, cols] in 0..15 -> int32 [rows, cols / 8]; nibble p of a word = column 8c + AWQ_REORDER[p]."""
    rows, cols = values.shape
    if cols % 8:
        raise ValueError(f"cols must be a multiple of 8, got {cols}")
    v = values.to(torch.int64).view(rows, cols // 8, 8)
    out = torch.zeros(rows, cols // 8, dtype=torch.int64, device=values.device)
    for p, src in enumerate(AWQ_REORDER):
        out |= (v[:, :, src] & 0xF) << (4 * p)
    return out.to(torch.int32)            # wraps modulo 2^32 like the reference's int32 arithmetic


def unpack_awq(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_awq`: int32 [rows, c] -> uint8 [rows, 8 c]."""
    w = packed.to(torch.int64) & 0xFFFFFFFF
    rows, c = packed.shape
    out = torch.empty(rows, c, 8, dtype=torch.uint8, device=packed.device)
    for p, src in enumerate(AWQ_REORDER):
        out[:, :, src] = ((w >> (4 * p)) & 0xF).to(torch.uint8)
    return out.view(rows, c * 8)


def to_awq_buffers(quantized: torch.Tensor, scales_2d: torch.Tensor, zeros_2d: torch.Tensor) -> Dict[str, torch.Tensor]:
    """[N, K] integer weights + [N, K/gs] scales / zero points -> checkpoint tensors (cli/convert.py:194-203)."""
    return {
        "qweight": pack_awq(quantized.t().contiguous().to(torch.int32)),
        "qzeros": pack_awq(zeros_2d.t().contiguous().to(torch.int32)),
        "scales": scales_2d.t().contiguous().to(torch.float16),
    }


def quantize_rotated_weight(weight: torch.Tensor, pairs: torch.Tensor, theta: torch.Tensor, channel_scales: torch.Tensor,
                            scales_flat: torch.Tensor, zp_flat: torch.Tensor, bits: int = 4, group_size: int = 128):
    """cli/convert.py:158-191 on the GPU: W' = rotate(W * cs) in fp32 (rotation::rotate), then
    q = clamp(round(W' / s) + zp, 0, 2^bits - 1) with zp = clamp(-round(zp_float), 0, qmax)
    (optim/quantizer.py:87-117).  Returns (quantized int32 [N, K], scales [N, K/gs], zeros int32 [N, K/gs])."""
    N, K = weight.shape
    qmax = (1 << bits) - 1
    w = weight.float() * channel_scales.float().view(1, K)
    w = torch.ops.rotation.rotate(w.contiguous(), pairs.contiguous(), theta.float().contiguous(), None, group_size)
    scale = scales_flat.float().view(-1, 1)
    zp = (-torch.round(zp_flat.float().view(-1, 1))).clamp(0, qmax)
    wg = w.reshape(-1, group_size)
    q = (torch.round(wg / scale) + zp).clamp(0, qmax)
    return (q.reshape(N, K).to(torch.int32), scale.reshape(N, K // group_size), zp.reshape(N, K // group_size).to(torch.int32))


def state_value(sd: Dict, *keys: str) -> int:
    """First of `keys` present in an optimiser state dict, as a Python int (0-d tensors are unwrapped) -- the optimiser
    writes hyper-parameters either flat ("n_bits") or under the quantiser ("quantizer.n_bits"), cli/convert.py:127-132."""
    for key in keys:
        if key in sd:
            val = sd[key]
            return int(val.item()) if isinstance(val, torch.Tensor) else int(val)
    raise KeyError(f"None of {keys} found")


def stack_if_numbered(sd: Dict, key: str) -> torch.Tensor:
    """`sd[key]`, or the stack of `sd["key.0"], sd["key.1"], ...` (a ParameterList in the optimiser), cli/convert.py:135-146."""
    if key in sd:
        return sd[key]
    parts, i = [], 0
    while f"{key}.{i}" in sd:
        parts.append(sd[f"{key}.{i}"])
        i += 1
    if parts:
        return torch.stack(parts)
    raise KeyError(key)


def quantize_layer(sd: Dict[str, torch.Tensor], device=None) -> Dict[str, torch.Tensor]:
    """Optimiser state dict of one linear (keys as cli/convert.py:239-262, both hyper-parameter spellings and numbered
    `pairs_grouped.N` / `angles_grouped.N` lists accepted) -> its checkpoint tensors (qweight, qzeros, scales, theta, pairs,
    channel_scales[, bias]); channel_scales are stored inverted (:264)."""
    dev = torch.device(device) if device is not None else sd["weight"].device
    g = lambda k: sd[k].to(dev)
    bits, gs = state_value(sd, "n_bits", "quantizer.n_bits"), state_value(sd, "group_size", "quantizer.group_size")
    pairs = stack_if_numbered(sd, "pairs_grouped").to(dev, torch.int16)
    theta = stack_if_numbered(sd, "angles_grouped").to(dev, torch.float32)
    q, s2d, z2d = quantize_rotated_weight(g("weight"), pairs, theta, g("channel_scales"),
                                          g("quantizer.scale"), g("quantizer.zero_point_float"), bits, gs)
    out = to_awq_buffers(q, s2d, z2d)
    out["theta"] = theta.to(torch.float16)
    out["pairs"] = pairs
    out["channel_scales"] = (1.0 / g("channel_scales").float()).to(torch.float16).view(1, -1)
    if sd.get("bias") is not None:
        out["bias"] = g("bias").to(torch.float16)
    return out


def quantize_moe(sd: Dict[str, torch.Tensor], device=None):
    """Optimiser state dict of one MoE expert block -> (per-projection AWQ buffers stacked over experts, the shared
    rotation buffers), cli/convert.py:280-379: gate_up [E, 2 I, H] and down [E, H, I] are each quantised with ONE rotation
    shared by all experts; rows [:I] of gate_up are gate_proj, rows [I:] up_proj; rotation buffers are named
    `gate_up_weight_{theta,pairs,channel_scales}` / `down_weight_*` with channel_scales stored inverted."""
    dev = torch.device(device) if device is not None else sd["gate_up_weight"].device
    g = lambda k: sd[k].to(dev)
    bits, gs = state_value(sd, "n_bits", "quantizer.n_bits"), state_value(sd, "group_size", "quantizer.group_size")
    gate_up, down = g("gate_up_weight").float(), g("down_weight").float()
    E, two_i, H = gate_up.shape
    _, H2, I = down.shape
    if H2 != H:
        raise ValueError(f"Unexpected MoE shapes: gate_up={tuple(gate_up.shape)} down={tuple(down.shape)}")   # convert.py:292-293
    rot, q = {}, {}
    for name, w, K in (("gate_up", gate_up.reshape(-1, H), H), ("down", down.reshape(-1, I), I)):
        pairs = stack_if_numbered(sd, f"{name}_pairs_grouped").to(dev, torch.int16)
        theta = stack_if_numbered(sd, f"{name}_angles_grouped").to(dev, torch.float32)
        cs = g(f"{name}_channel_scales").float()
        q[name] = quantize_rotated_weight(w, pairs, theta, cs, g(f"{name}_quantizer.scale"), g(f"{name}_quantizer.zero_point_float"), bits, gs)
        rot[f"{name}_weight_theta"] = theta.to(torch.float16)
        rot[f"{name}_weight_pairs"] = pairs
        rot[f"{name}_weight_channel_scales"] = (1.0 / cs).to(torch.float16).view(1, -1)
    half = two_i // 2
    gq, gsc, gz = (t.reshape(E, two_i, -1) for t in q["gate_up"])
    dq, dsc, dz = (t.reshape(E, H, -1) for t in q["down"])
    out = {}
    for proj, qq, sc, zp in (("gate_proj", gq[:, :half], gsc[:, :half], gz[:, :half]), ("up_proj", gq[:, half:], gsc[:, half:], gz[:, half:]),
                             ("down_proj", dq, dsc, dz)):
        bufs = [to_awq_buffers(qq[e], sc[e], zp[e]) for e in range(E)]
        out[proj] = {k: torch.stack([b[k] for b in bufs]) for k in ("qweight", "qzeros", "scales")}
    return out, rot


# ----------------------------------------------------------------------------- pre-packed (CDNA4 layout) files
_PREPACKED_VERSION = 1


def save_prepacked(pk: PackedParoWeights, path: str) -> None:
    """Write one linear in the kernel layout (safetensors): skips repack_awq / pack_rotation at the next load."""
    from safetensors.torch import save_file
    tensors = {"wq": pk.wq, "sz": pk.sz, "rot": pk.rot, "theta": pk.theta, "pairs": pk.pairs,
               "channel_scales": pk.channel_scales}
    if pk.bias is not None:
        tensors["bias"] = pk.bias
    meta = {"format": "paroquant_amd.prepacked", "version": str(_PREPACKED_VERSION), "K": str(pk.K), "N": str(pk.N),
            "partition_sizes": ",".join(str(s) for s in pk.partition_sizes), "wq_order": str(pk.wq_order)}
    save_file({k: v.contiguous().cpu() for k, v in tensors.items()}, path, metadata=meta)


def load_prepacked(path: str, device) -> PackedParoWeights:
    """Read a :func:`save_prepacked` file straight into a :class:`PackedParoWeights` (no device-side repack)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
        if meta.get("format") != "paroquant_amd.prepacked" or int(meta.get("version", "0")) != _PREPACKED_VERSION:
            raise ValueError(f"{path} is not a paroquant_amd pre-packed file of version {_PREPACKED_VERSION}")
        t = {k: f.get_tensor(k).to(device) for k in f.keys()}
    return PackedParoWeights.from_packed(t["wq"], t["sz"], t["rot"], t["theta"], t["pairs"], t["channel_scales"],
                                         [int(s) for s in meta["partition_sizes"].split(",")], int(meta["K"]),
                                         int(meta["wq_order"]), t.get("bias"))

"""``RotateQuantizedLinear`` -- the HF-side operator of the reference, backed by the fused
MI355X kernels.

Same constructor, same flat buffer names / dtypes / shapes and the same ``forward`` contract as
``paroquant/inference/backends/transformers/modules.py:16-71`` so existing ``*-PARO`` checkpoints
load unchanged through ``from_pretrained`` (state-dict keys ``...q_proj.qweight``,
``...q_proj.theta`` ... match directly, modules.py:19-22).  The difference is underneath:
``rotate -> AutoAWQ WQLinearMMFunction`` (two launches + a third-party GEMM) becomes ONE call of
``torch.ops.paro.w4a16_linear`` on a one-time CDNA4 repack of the AWQ buffers.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops


def pad_partitions(qweight, qzeros, scales, sizes, pack: int = 8, multiple: int = 16):
    """Pad every merged partition's column count up to ``multiple`` with zero weights / zero scales
    (the reference pads N to the Marlin tile the same way, vllm/plugin.py:210-217).
    Returns ``(qweight, qzeros, scales, padded_sizes)``; a no-op when nothing needs padding."""
    sizes = [int(s) for s in sizes]
    padded = [((s + multiple - 1) // multiple) * multiple for s in sizes]
    if padded == sizes:
        return qweight, qzeros, scales, padded
    F = torch.nn.functional
    qws = qweight.split([s // pack for s in sizes], dim=1)
    qzs = qzeros.split([s // pack for s in sizes], dim=1)
    scs = scales.split(sizes, dim=1)
    qweight = torch.cat([F.pad(t, (0, (p - s) // pack)) for t, s, p in zip(qws, sizes, padded)], dim=1)
    qzeros = torch.cat([F.pad(t, (0, (p - s) // pack)) for t, s, p in zip(qzs, sizes, padded)], dim=1)
    scales = torch.cat([F.pad(t, (0, p - s)) for t, s, p in zip(scs, sizes, padded)], dim=1)
    return qweight.contiguous(), qzeros.contiguous(), scales.contiguous(), padded


def coalesce_partitions(theta, pairs, channel_scales, sizes):
    """Merge ADJACENT partitions whose rotation parameters are identical into one kernel partition.

    vLLM loads one checkpoint rotation into several slots when a fused projection spans more than one output partition
    (tuple shard ids, reference ``vllm/plugin.py:60-76``: Qwen3.5's ``in_proj_qkvz`` = ``in_proj_qkv`` -> slots (0, 1, 2) +
    ``in_proj_z`` -> slot 3).  The reference then rotates x once per slot (``plugin.py:288-306``); the fused kernel's cost per
    128-channel group grows with the number of rotations it runs, so slots that hold the SAME rotation over neighbouring columns
    are one partition here (the outputs are identical: same rotated x, same columns).  Returns
    ``(theta, pairs, channel_scales, merged_sizes, slot_of_partition)``."""
    P = len(sizes)
    cs = channel_scales.reshape(P, -1)
    keep, merged = [0], [int(sizes[0])]
    for p in range(1, P):
        q = keep[-1]
        if torch.equal(theta[p], theta[q]) and torch.equal(pairs[p], pairs[q]) and torch.equal(cs[p], cs[q]):
            merged[-1] += int(sizes[p])
        else:
            keep.append(p)
            merged.append(int(sizes[p]))
    if len(keep) == P:
        return theta, pairs, channel_scales, [int(s) for s in sizes], keep
    idx = torch.tensor(keep, device=theta.device)
    return theta.index_select(0, idx), pairs.index_select(0, idx), cs.index_select(0, idx).reshape(len(keep), 1, -1), merged, keep


class PackedParoWeights:
    """Kernel-ready parameters of one (possibly merged) ParoQuant linear.

    Built once from checkpoint-format tensors (``cli/convert.py:268-277``):
      qweight int32 [K, N/8], qzeros int32 [K/gs, N/8], scales f16 [K/gs, N]  (gs = group_size, 128 or 64),
      theta f16 [P, krot, K/2], pairs int16 [P, krot, K], channel_scales f16 [P, 1, K].
    Takes the place of the Marlin-repacked tensors the reference stashes on the layer in
    ``process_weights_after_loading`` (vllm/plugin.py:251-279).
    """

    def __init__(self, qweight, qzeros, scales, theta, pairs, channel_scales, partition_sizes: Sequence[int],
                 bias: Optional[torch.Tensor] = None, group_size: Optional[int] = None, bits: int = 4,
                 wq_order: Optional[int] = None):
        if bits != 4:
            raise ValueError(f"Unsupported bits={bits}. Supported: [4]")          # plugin.py:84-85
        if group_size is None:    # what the checkpoint tensors say: rows of qzeros = K / group_size
            group_size = qweight.shape[0] // qzeros.shape[0] if qzeros.dim() == 2 and qzeros.shape[0] else 0
        if group_size not in (64, 128):
            raise ValueError(f"Unsupported group_size={group_size}; the fused kernels take a quantisation group of 64 or "
                             "128 (the rotation group is 128 at inference either way, modules.py:59-60)")
        self.partition_sizes = [int(s) for s in partition_sizes]
        self.group_size = int(group_size)
        K = qweight.shape[0]
        N = qweight.shape[1] * 8
        if tuple(qzeros.shape) != (K // group_size, N // 8) or tuple(scales.shape) != (K // group_size, N):
            raise ValueError(f"qzeros {tuple(qzeros.shape)} / scales {tuple(scales.shape)} do not match group_size {group_size} "
                             f"for a [{K}, {N}] layer")
        if sum(self.partition_sizes) != N:
            raise ValueError(f"partition sizes {self.partition_sizes} do not sum to out_features {N}")
        if any(s % 16 for s in self.partition_sizes):
            raise ValueError(f"partition sizes must be multiples of 16, got {self.partition_sizes}")
        P = len(self.partition_sizes)
        if theta.dim() == 2:
            theta, pairs = theta[None], pairs[None]
        channel_scales = channel_scales.reshape(P, K)
        self.K, self.N = K, N
        self.theta = theta.to(torch.float16).contiguous()
        self.pairs = pairs.to(torch.int16).contiguous()
        self.channel_scales = channel_scales.to(torch.float16).contiguous()
        # kernel layouts: INT4 tiles in MFMA B-fragment order, one (scale, 16 + zero) word per (group,
        # column), one (i, j, theta) word per (group, pair lane, stage)  -- include/paro_abi.h
        # wide outputs are streamed 8 tiles per wave: keep a wave's tiles contiguous ([group][tile] order);
        # narrow outputs run one tile per 16-wave workgroup: keep a workgroup's groups contiguous
        self.wq_order = int(wq_order) if wq_order is not None else (1 if N // 16 >= 1024 else 0)
        self.wq, self.sz = torch.ops.paro.repack_awq(qweight, qzeros, scales.to(torch.float16),
                                                     self.partition_sizes, self.wq_order)
        self.rot = torch.ops.paro.pack_rotation(self.pairs, self.theta)
        self.bias = bias
        self._rmat = {}
        self.workspace = ops.get_workspace(qweight.device, ops.decode_workspace_bytes(K, N, P))

    @classmethod
    def from_packed(cls, wq, sz, rot, theta, pairs, channel_scales, partition_sizes: Sequence[int], K: int,
                    wq_order: int = 0, bias: Optional[torch.Tensor] = None) -> "PackedParoWeights":
        """Adopt buffers that are ALREADY in the kernel layout (``pack.load_prepacked``): no repack launches."""
        self = cls.__new__(cls)
        self.partition_sizes = [int(s) for s in partition_sizes]
        self.K, self.N = int(K), int(sum(self.partition_sizes))
        P = len(self.partition_sizes)
        self.theta, self.pairs = theta.to(torch.float16).contiguous(), pairs.to(torch.int16).contiguous()
        self.channel_scales = channel_scales.reshape(P, self.K).to(torch.float16).contiguous()
        self.wq_order = int(wq_order)
        self.wq, self.sz, self.rot = wq.contiguous(), sz.contiguous(), rot.contiguous()
        tsz = sum((n // 16 + 7) // 8 * 8 for n in self.partition_sizes)
        self.group_size = self.K // (self.sz.numel() // (tsz * 16))   # rows of the packed scale/zero array = K / group_size
        if self.group_size not in (64, 128):
            raise ValueError(f"packed scale/zero tensor of {self.sz.numel()} words does not belong to a [{self.K}, {self.N}] layer")
        self.bias = bias
        self._rmat = {}
        self.workspace = ops.get_workspace(wq.device, ops.decode_workspace_bytes(self.K, self.N, P))
        return self

    def rotation_matrices(self, dtype: torch.dtype) -> torch.Tensor:
        """Dense per-group rotation matrices for the prefill pre-pass on the matrix cores:
        ``rmat[p, g, n, k] = (diag(cs) G_1 .. G_krot)[k, n]`` of group g -- obtained by pushing the scaled
        identity through ``torch.ops.rotation.rotate`` (fp32) and rounding once.  Built lazily on the
        first large-M call (decode-only users never pay the P*K*256 bytes)."""
        r = self._rmat.get(dtype)
        if r is None:
            K, P = self.K, len(self.partition_sizes)
            G = K // 128
            eye = torch.zeros(128, G, 128, dtype=torch.float32, device=self.wq.device)
            eye[torch.arange(128), :, torch.arange(128)] = 1.0
            eye = eye.reshape(128, K)
            mats = []
            for p in range(P):
                full = torch.ops.rotation.rotate(eye, self.pairs[p], self.theta[p].float(),
                                                 self.channel_scales[p].float())          # [k, G*128 (n)]
                mats.append(full.view(128, G, 128).permute(1, 2, 0))                        # [g, n, k]
            r = torch.stack(mats).to(dtype).contiguous()
            self._rmat[dtype] = r
        return r

    def prepare_prefill(self, dtype: torch.dtype = torch.float16) -> "PackedParoWeights":
        """Build the dense prefill rotation matrices now (P * K * 256 bytes) instead of on the first call with
        >= 256 rows -- call it before capturing a prefill step in a HIP graph, where nothing may allocate."""
        self.rotation_matrices(dtype)
        return self

    def bind_stream(self, stream) -> "PackedParoWeights":
        """Use a workspace private to ``stream`` (needed only when launches of different layers may overlap in time
        on different streams: K-split granules must not be shared by concurrent launches)."""
        self.workspace = ops.get_workspace(self.wq.device, ops.decode_workspace_bytes(self.K, self.N, len(self.partition_sizes)),
                                           stream)
        return self

    def autotune(self, dtype: torch.dtype = torch.float16, **kw) -> dict:
        """Measure the batch-1 GEMV's launch shapes for this layer once (paroquant_amd/autotune.py; cached per distinct shape) and keep
        the fastest as ``launch_hint`` -- the one-time counterpart of the reference's ``process_weights_after_loading`` repack
        (vllm/plugin.py:251-279).  Allocates and launches: call it at load time, never inside a captured graph."""
        from .autotune import autotune_packed
        return autotune_packed(self, dtype, **kw)

    def apply(self, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        b = self.bias if bias is None else bias
        if b is not None and b.dtype != x.dtype:
            b = b.to(x.dtype)
        rows = x.numel() // self.K
        rmat = self.rotation_matrices(x.dtype) if rows >= 256 else None
        return torch.ops.paro.w4a16_linear(x, self.wq, self.sz, self.rot, self.pairs, self.theta,
                                           self.channel_scales, b, self.partition_sizes, self.workspace,
                                           self.wq_order, rmat, getattr(self, "launch_hint", 0))

    def fold_norm_weight(self, weight: torch.Tensor, plus_one: bool = False) -> "PackedParoWeights":
        """Fold the weight of the RMSNorm that feeds this linear into the channel scales
        (``cs'[p, k] = cs[p, k] * w[k]``, fp32 product rounded once): with it the whole norm in front of the linear
        reduces to the scalar ``rsqrt(mean(x^2) + eps)`` that the fused GEMV applies (``ops.w4a16_gemv_fused``,
        prologue RMSNORM).  Call once, after loading."""
        if getattr(self, "_norm_folded", False):
            raise RuntimeError("a norm weight has already been folded into these channel scales")
        w = weight.to(self.channel_scales.device).float().view(1, self.K)
        if plus_one:        # Gemma / Qwen3.5 RMSNorm: y = x_hat * (1 + weight)
            w = w + 1.0
        self.channel_scales = (self.channel_scales.float() * w).to(torch.float16).contiguous()
        self._norm_folded = True
        self._rmat = {}
        return self

    def stream_buffers(self):
        """The buffers a decode launch streams from HBM ."""
        return [self.wq, self.sz, self.rot]

    def nbytes(self) -> int:
        ts = [self.wq, self.sz, self.rot, self.channel_scales]
        return sum(t.numel() * t.element_size() for t in ts)


class RotateQuantizedLinear(nn.Module):
    """HF-side operator: ``y = rotate(x * channel_scales) @ dequant(qweight, qzeros, scales) (+ bias)`` on the fused
    CDNA4 kernels.

    Drop-in for the reference module of the same name (transformers/modules.py:16-71): same constructor, and the
    checkpoint tensors of a linear are plain buffers of THIS module under the reference's names, dtypes and shapes
    (modules.py:43-55) so that ``*-PARO`` safetensors keys such as ``...gate_proj.theta`` load without remapping:

        theta          f16   [krot, in/2]        pairs   int16 [krot, in]       channel_scales f16 [1, in]
        qweight        int32 [in, out/8]         qzeros  int32 [in/gs, out/8]   scales         f16 [in/gs, out]
        bias           f16   [out]  (optional)
    """

    def __init__(self, in_features: int, out_features: int, bias: bool = False, group_size: int = 128, bits: int = 4,
                 krot: int = 8):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.w_bit, self.group_size = bits, group_size
        per_word, groups = 32 // bits, in_features // group_size
        i16, i32, f16 = torch.int16, torch.int32, torch.float16
        for name, shape, dtype, fill in (
                ("theta", (krot, in_features // 2), f16, 0), ("pairs", (krot, in_features), i16, 0),
                ("channel_scales", (1, in_features), f16, 1),
                ("qweight", (in_features, out_features // per_word), i32, 0),
                ("qzeros", (groups, out_features // per_word), i32, 0), ("scales", (groups, out_features), f16, 0)):
            self.register_buffer(name, torch.full(shape, fill, dtype=dtype))

        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16))
        else:
            self.bias = None
        self._packed: Optional[PackedParoWeights] = None

    def prepare(self, prefill: bool = False) -> "RotateQuantizedLinear":
        """One-time repack of the checkpoint buffers into the CDNA4 tile layout (device side).

        Called lazily by ``forward``; call it explicitly after loading weights (the HF quantizer does,
        in ``_process_model_after_weight_loading``) so nothing allocates inside a captured HIP graph.
        ``prefill=True`` also builds the dense rotation matrices of the >= 256-row path up front."""
        if not self.qweight.is_cuda:
            raise RuntimeError("ParoQuant requires a GPU: RotateQuantizedLinear has no CPU path "
                               "(reference: transformers/quantizer.py:78-80)")
        qw, qz, sc, padded = pad_partitions(self.qweight, self.qzeros, self.scales, [self.out_features])
        bias = self.bias
        if bias is not None and padded[0] != self.out_features:
            bias = torch.nn.functional.pad(bias, (0, padded[0] - self.out_features))
        self._packed = PackedParoWeights(qw, qz, sc, self.theta, self.pairs, self.channel_scales, padded, bias,
                                         self.group_size, self.w_bit)
        if prefill:
            self._packed.prepare_prefill()
        from . import autotune
        if autotune.enabled():                  # PARO_AUTOTUNE=1: tuned with the module's activation type (scales carry it: modules.py:43-55)
            self._packed.autotune(torch.bfloat16 if self.scales.dtype == torch.bfloat16 else torch.float16)
        return self

    def release_checkpoint_buffers(self) -> "RotateQuantizedLinear":
        """Drop the AWQ-format ``qweight`` / ``qzeros`` / ``scales`` once the kernel-layout copy exists (they double
        the INT4 footprint otherwise: +35 GB on a 70B model).  Opt-in: afterwards the module can no longer be
        re-packed (``.to()`` another device) or saved -- the reference keeps the buffers because AutoAWQ computes
        on them directly (modules.py:61-70), the vLLM path frees them like this (plugin.py:276-279)."""
        if self._packed is None:
            self.prepare()
        for name in ("qweight", "qzeros", "scales"):
            setattr(self, name, torch.empty(0, dtype=getattr(self, name).dtype, device=getattr(self, name).device))
        self._released = True
        return self

    def _apply(self, fn, *args, **kwargs):   # .to()/.cuda() invalidate the packed copy
        if getattr(self, "_released", False):
            raise RuntimeError("RotateQuantizedLinear: checkpoint buffers were released; the module cannot be moved")
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dtype in (torch.float16, torch.bfloat16), f"Expected float16 input, got {x.dtype}"   # modules.py:59
        if self._packed is None:
            self.prepare()
        y = self._packed.apply(x)
        return y if y.shape[-1] == self.out_features else y[..., : self.out_features]

"""Torch operator surface of the MI355X-native ParoQuant hot path.

Registers, with the reference's exact schema,

    torch.ops.rotation.rotate(Tensor x, Tensor idx_ij, Tensor theta, Tensor? scales=None,
                              int group_size=128) -> Tensor

(reference: ``TORCH_LIBRARY(rotation)`` at paroquant/kernels/cuda/rotation.cu:128-135, fake kernel at
paroquant/kernels/cuda/__init__.py:54-61) and the operators this build adds behind
``RotateQuantizedLinear`` / ``ParoQuantLinearMethod``:

    torch.ops.paro.repack_awq(qweight, qzeros, scales, partition_sizes) -> (wq, sz)
    torch.ops.paro.pack_rotation(pairs, theta) -> rot
    torch.ops.paro.w4a16_linear(x, wq, sz, rot, pairs, theta, channel_scales, bias?,
                                partition_sizes, workspace) -> Tensor

All device work goes through the C ABI of ``libparo_mi355x.so`` (``_native``) on torch's current
HIP stream; implementations exist for the GPU dispatch key only, exactly like the reference.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _native as nat

_ROTATE_SCHEMA = "rotate(Tensor x, Tensor idx_ij, Tensor theta, Tensor? scales=None, int group_size=128) -> Tensor"

# --------------------------------------------------------------------------------------
# rotation::rotate
# --------------------------------------------------------------------------------------


def _rotate_impl(x: torch.Tensor, idx_ij: torch.Tensor, theta: torch.Tensor,
                 scales: Optional[torch.Tensor] = None, group_size: int = 128) -> torch.Tensor:
    lib = nat.load()
    # same checks, same order, same messages as rotate_dynamic / rotate_launcher (rotation.cu:111-124,62-66)
    if theta.size(0) != idx_ij.size(0):
        raise RuntimeError("theta.size(0) must equal idx_ij.size(0)")
    if idx_ij.dtype != torch.int16:
        raise RuntimeError(f"idx_ij must be int16, got {idx_ij.dtype}")
    h = x.size(-1)
    x = x.contiguous()                      # the reference reads raw data_ptr (rotation.cu:48)
    idx_ij = idx_ij.contiguous()
    theta = theta.contiguous()
    pd = theta.dtype
    if scales is not None and scales.numel() > 0:
        scales = scales.reshape(-1).to(dtype=pd).contiguous()
        if scales.numel() != h:
            raise RuntimeError(f"scales must have {h} elements, got {scales.numel()}")
        s_ptr = scales.data_ptr()
    else:
        s_ptr = None
    out = torch.empty_like(x)
    rows = x.numel() // h if h > 0 else 0
    if rows == 0:
        if group_size not in (64, 128):
            raise RuntimeError(f"Unsupported group_size: {group_size}; expected 64 or 128")
        return out
    with torch.cuda.device(x.device):     # launch on x's device, not the thread's current one
        nat.check(lib.paro_rotate(x.data_ptr(), out.data_ptr(), idx_ij.data_ptr(), theta.data_ptr(), s_ptr, rows, h,
                                  int(theta.size(0)), int(group_size), nat.dtype_code(x.dtype), nat.dtype_code(pd),
                                  nat.current_stream_ptr(x.device)))
    return out


def _rotate_fake(x, idx_ij, theta, scales=None, group_size=128):
    return torch.empty_like(x)


# --------------------------------------------------------------------------------------
# paro::repack_awq / paro::pack_rotation
# --------------------------------------------------------------------------------------


def _part_array(partition_sizes: Sequence[int]):
    sizes = [int(s) for s in partition_sizes]
    if not 1 <= len(sizes) <= nat.PARO_MAX_PARTS:
        raise ValueError(f"between 1 and {nat.PARO_MAX_PARTS} merged partitions are supported, got {len(sizes)}")
    return (ctypes.c_int32 * len(sizes))(*sizes), sizes


def _repack_impl(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                 partition_sizes: Sequence[int], wq_order: int = 0):
    lib = nat.load()
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        raise RuntimeError("qweight / qzeros must be int32 (AWQ packing, cli/convert.py:149-155)")
    if scales.dtype != torch.float16:
        raise RuntimeError("scales must be float16 (checkpoint dtype, cli/convert.py:201)")
    K, NW = qweight.shape
    N = NW * 8
    arr, sizes = _part_array(partition_sizes)
    if K % 128 != 0:
        raise ValueError(f"in_features must be a multiple of 128, got {K}")
    if any(s <= 0 or s % 16 for s in sizes) or sum(sizes) != N:
        raise ValueError(f"partition sizes {sizes} must be positive multiples of 16 summing to out_features {N}")
    # the quantisation group is what the checkpoint tensors say it is: K / rows(qzeros); 128 or 64 (the rotation
    # always works on 128-channel groups at inference, transformers/modules.py:59)
    rows_q = int(qzeros.shape[0]) if qzeros.dim() == 2 else 0
    gs = K // rows_q if rows_q and K % rows_q == 0 else 0
    if gs not in (64, 128) or tuple(qzeros.shape) != (K // gs, NW) or tuple(scales.shape) != (K // gs, N):
        raise ValueError(f"qzeros {tuple(qzeros.shape)} / scales {tuple(scales.shape)} do not match "
                         f"[K/gs, {NW}] / [K/gs, {N}] for a group_size gs of 64 or 128 (K = {K})")
    qweight, qzeros, scales = qweight.contiguous(), qzeros.contiguous(), scales.contiguous()
    wq = torch.empty(lib.paro_packed_qweight_bytes(K, N) // 4, dtype=torch.int32, device=qweight.device)
    sz = torch.empty(lib.paro_packed_sz_bytes(K, gs, len(sizes), arr) // 4, dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        nat.check(lib.paro_repack_awq(qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(), K, N, gs, len(sizes), arr,
                                      int(wq_order), wq.data_ptr(), sz.data_ptr(), nat.current_stream_ptr(qweight.device)))
    return wq, sz


def _repack_fake(qweight, qzeros, scales, partition_sizes, wq_order=0):
    K, NW = qweight.shape
    tsz = sum((int(s) // 16 + 7) // 8 * 8 for s in partition_sizes)
    return qweight.new_empty(K * NW), qweight.new_empty(int(qzeros.shape[0]) * tsz * 16)


def _pack_rotation_impl(pairs: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    lib = nat.load()
    if pairs.dtype != torch.int16 or theta.dtype != torch.float16:
        raise RuntimeError("pairs must be int16 and theta float16 (checkpoint dtypes, cli/convert.py:270-271)")
    if pairs.dim() != 3 or theta.dim() != 3 or pairs.shape[:2] != theta.shape[:2] or pairs.size(2) != 2 * theta.size(2):
        raise ValueError(f"expected pairs [P, krot, K] and theta [P, krot, K/2], got {tuple(pairs.shape)} / "
                         f"{tuple(theta.shape)}")
    P, krot, K = pairs.shape
    if krot > 8:
        return torch.empty(0, dtype=torch.int32, device=pairs.device)   # unfused route (rotate kernel + GEMV)
    pairs, theta = pairs.contiguous(), theta.contiguous()
    rot = torch.empty(lib.paro_packed_rot_bytes(K, P) // 4, dtype=torch.int32, device=pairs.device)
    # the library never allocates or synchronises: the "illegal pair" status word is ours, and so is the read-back
    status = torch.empty(1, dtype=torch.int32, device=pairs.device)
    with torch.cuda.device(pairs.device):
        nat.check(lib.paro_pack_rotation(pairs.data_ptr(), theta.data_ptr(), K, P, krot, rot.data_ptr(),
                                         status.data_ptr(), nat.current_stream_ptr(pairs.device)))
    if not torch.cuda.is_current_stream_capturing() and int(status.item()) != 0:
        raise RuntimeError("illegal pair: a rotation stage is not a perfect matching of its 128-channel group "
                           "(indices out of range, repeated or i == j)")     # optim/rotation.py:36-37
    return rot


def _pack_rotation_fake(pairs, theta):
    P, krot, K = pairs.shape
    return pairs.new_empty(0 if krot > 8 else P * (K // 128) * 768, dtype=torch.int32)


# --------------------------------------------------------------------------------------
# paro::w4a16_linear
# --------------------------------------------------------------------------------------


def make_desc(K: int, partition_sizes: Sequence[int], krot: int, act_dtype: torch.dtype, wq, sz, rot, pairs,
              theta, channel_scales, bias, wq_order: int = 0, rmat=None, group_size: int = 0, launch_hint: int = 0) -> nat.ParoLinearDesc:
    """``group_size`` 0: read it off the packed scale/zero tensor (rows = K / group_size); stacked expert tensors
    (moe.py) pass it explicitly."""
    d = nat.ParoLinearDesc()
    if group_size == 0:
        tsz = sum((int(n) // 16 + 7) // 8 * 8 for n in partition_sizes)
        rows_q = sz.numel() // (tsz * 16) if tsz else 0
        group_size = K // rows_q if rows_q and K % rows_q == 0 else -1
    if group_size not in (64, 128):
        raise ValueError(f"Unsupported group_size: {group_size}; expected 64 or 128")
    d.group_size = int(group_size)
    d.launch_hint = int(launch_hint)
    d.K = K
    d.N = int(sum(partition_sizes))
    d.n_parts = len(partition_sizes)
    d.krot = krot
    if len(partition_sizes) > nat.PARO_MAX_PARTS:
        raise ValueError(f"at most {nat.PARO_MAX_PARTS} merged partitions are supported")
    for i, n in enumerate(partition_sizes):
        d.part_cols[i] = int(n)
    d.act_dtype = nat.dtype_code(act_dtype)
    d.wq_order = int(wq_order)
    d.wq = wq.data_ptr()
    d.sz = sz.data_ptr()
    d.rot = rot.data_ptr() if rot is not None and rot.numel() > 0 else None
    d.pairs = pairs.data_ptr()
    d.theta = theta.data_ptr()
    d.channel_scales = channel_scales.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.rmat = rmat.data_ptr() if rmat is not None and rmat.numel() > 0 and rmat.dtype == act_dtype else None
    return d


def _check_linear_args(x, pairs, theta, channel_scales, bias, partition_sizes):
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"paro::w4a16_linear expects float16 or bfloat16 activations, got {x.dtype}")
    P = len(partition_sizes)
    K = x.size(-1)
    if theta.dtype != torch.float16 or channel_scales.dtype != torch.float16:
        raise RuntimeError("theta / channel_scales must be float16 (checkpoint dtype, cli/convert.py:264-277)")
    if pairs.dtype != torch.int16:
        raise RuntimeError("pairs must be int16")
    if pairs.dim() != 3 or pairs.size(0) != P or pairs.size(2) != K:
        raise ValueError(f"pairs must be [n_parts={P}, krot, K={K}], got {tuple(pairs.shape)}")
    if theta.dim() != 3 or theta.size(0) != P or theta.size(1) != pairs.size(1) or theta.size(2) != K // 2:
        raise ValueError(f"theta must be [n_parts={P}, krot, K/2], got {tuple(theta.shape)}")
    if channel_scales.numel() != P * K:
        raise ValueError(f"channel_scales must hold n_parts*K elements, got {channel_scales.numel()}")
    if bias is not None and bias.dtype != x.dtype:
        raise RuntimeError("bias dtype must match the activation dtype")


def _w4a16_impl(x: torch.Tensor, wq: torch.Tensor, sz: torch.Tensor, rot: torch.Tensor, pairs: torch.Tensor,
                theta: torch.Tensor, channel_scales: torch.Tensor, bias: Optional[torch.Tensor],
                partition_sizes: Sequence[int], workspace: torch.Tensor, wq_order: int = 0,
                rmat: Optional[torch.Tensor] = None, launch_hint: int = 0) -> torch.Tensor:
    lib = nat.load()
    partition_sizes = [int(s) for s in partition_sizes]
    _check_linear_args(x, pairs, theta, channel_scales, bias, partition_sizes)
    K = x.size(-1)
    N = sum(partition_sizes)
    x2 = x.reshape(-1, K).contiguous()
    rows = x2.size(0)
    y = torch.empty((rows, N), dtype=x.dtype, device=x.device)
    if rows == 0:
        return y.reshape(*x.shape[:-1], N)
    d = make_desc(K, partition_sizes, int(pairs.size(1)), x.dtype, wq, sz, rot, pairs, theta, channel_scales, bias,
                  wq_order, rmat, launch_hint=launch_hint)
    # Scratch: K-split granules / rotated activations / fp32 partial tiles.  NEVER uninitialised memory: the
    # granule protocol of the split-K GEMV reads {tag, partial} words and needs the slabs to start at zero, so a
    # workspace that is too small is replaced by the cached zero-filled one (grown with torch.zeros).
    need = lib.paro_linear_workspace_bytes(ctypes.byref(d), rows)
    ws = workspace
    if ws.numel() * ws.element_size() < need:
        ws = get_workspace(x.device, need)
    with torch.cuda.device(x.device):
        nat.check(lib.paro_w4a16_linear(ctypes.byref(d), x2.data_ptr(), y.data_ptr(), rows, ws.data_ptr(),
                                        ws.numel() * ws.element_size(), nat.current_stream_ptr(x.device)))
    return y.reshape(*x.shape[:-1], N)


def _w4a16_fake(x, wq, sz, rot, pairs, theta, channel_scales, bias, partition_sizes, workspace, wq_order=0,
                rmat=None, launch_hint=0):
    return x.new_empty((*x.shape[:-1], int(sum(partition_sizes))))


def w4a16_gemv_tuned(x, pk, tiles_per_wave: int = 0, ksplit: int = 0, waves: int = 0, mode: int = 0,
                     bias=None) -> torch.Tensor:
    """Direct call of ``paro_w4a16_gemv`` with explicit launch-shape knobs (benchmarks / tuning sweeps).
    ``pk`` is a :class:`paroquant_amd.linear.PackedParoWeights`.  ``mode = 2``: ``x`` is ``[n_parts, rows, K]``, already
    rotated per partition by the caller (``rotation::rotate`` or a producer kernel); returns ``[rows, N]``."""
    lib = nat.load()
    _check_linear_args(x, pk.pairs, pk.theta, pk.channel_scales, bias, pk.partition_sizes)
    K, N = pk.K, pk.N
    if mode == 2:
        P = len(pk.partition_sizes)
        if x.dim() != 3 or x.size(0) != P or x.size(-1) != K:
            raise ValueError(f"mode 2 takes pre-rotated activations [n_parts = {P}, rows, K = {K}], got {tuple(x.shape)}")
        x2 = x.contiguous()
        rows = x2.size(1)
        y = torch.empty((rows, N), dtype=x.dtype, device=x.device)
        d = make_desc(K, pk.partition_sizes, int(pk.pairs.size(1)), x.dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                      pk.channel_scales, bias, pk.wq_order)
        ws = pk.workspace
        with torch.cuda.device(x.device):
            nat.check(lib.paro_w4a16_gemv(ctypes.byref(d), x2.data_ptr(), y.data_ptr(), rows, ws.data_ptr(),
                                          ws.numel() * ws.element_size(), tiles_per_wave, ksplit, waves, 2,
                                          nat.current_stream_ptr(x.device)))
        return y
    x2 = x.reshape(-1, K).contiguous()
    rows = x2.size(0)
    y = torch.empty((rows, N), dtype=x.dtype, device=x.device)
    d = make_desc(K, pk.partition_sizes, int(pk.pairs.size(1)), x.dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                  pk.channel_scales, bias, pk.wq_order, launch_hint=getattr(pk, "launch_hint", 0))   # (explicit knobs override the hint: gemv.hip)
    ws = pk.workspace
    need = lib.paro_linear_workspace_bytes(ctypes.byref(d), rows)
    if ws.numel() * ws.element_size() < need:
        ws = get_workspace(x.device, need)
    with torch.cuda.device(x.device):
        nat.check(lib.paro_w4a16_gemv(ctypes.byref(d), x2.data_ptr(), y.data_ptr(), rows, ws.data_ptr(),
                                      ws.numel() * ws.element_size(), tiles_per_wave, ksplit, waves, mode,
                                      nat.current_stream_ptr(x.device)))
    return y.reshape(*x.shape[:-1], N)


def w4a16_gemm_forced(x, pk, bias=None, use_rmat: bool = True, variant: int = 0) -> torch.Tensor:
    """Direct call of ``paro_w4a16_gemm`` regardless of the row count (tests / benchmarks); ``variant`` forces one
    of the GEMM kernels (include/paro_abi.h: 1 = 128x128, 2 = 256x128, 3 = 256x256 2x4 waves, 4 = 256x256 1x8 waves)."""
    lib = nat.load()
    _check_linear_args(x, pk.pairs, pk.theta, pk.channel_scales, bias, pk.partition_sizes)
    K, N = pk.K, pk.N
    x2 = x.reshape(-1, K).contiguous()
    rows = x2.size(0)
    y = torch.empty((rows, N), dtype=x.dtype, device=x.device)
    d = make_desc(K, pk.partition_sizes, int(pk.pairs.size(1)), x.dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                  pk.channel_scales, bias, pk.wq_order, pk.rotation_matrices(x.dtype) if use_rmat else None)
    need = lib.paro_linear_workspace_bytes(ctypes.byref(d), rows)
    ws = get_workspace(x.device, need)
    with torch.cuda.device(x.device):
        nat.check(lib.paro_w4a16_gemm(ctypes.byref(d), x2.data_ptr(), y.data_ptr(), rows, ws.data_ptr(),
                                      ws.numel() * ws.element_size(),
                                      int(variant), nat.current_stream_ptr(x.device)))
    return y.reshape(*x.shape[:-1], N)


def dequant_packed(wq, sz, K: int, partition_sizes: Sequence[int], dtype=torch.float16, wq_order: int = 0) -> torch.Tensor:
    """Dense ``W[K, N] = (q - z) * s`` from the packed buffers (verification aid)."""
    lib = nat.load()
    N = int(sum(partition_sizes))
    out = torch.empty((K, N), dtype=dtype, device=wq.device)
    d = nat.ParoLinearDesc()
    d.K, d.N, d.n_parts, d.krot = K, N, len(partition_sizes), 8
    for i, n in enumerate(partition_sizes):
        d.part_cols[i] = int(n)
    d.act_dtype = nat.dtype_code(dtype)
    d.wq_order = int(wq_order)
    d.wq, d.sz = wq.data_ptr(), sz.data_ptr()
    tsz = sum((int(n) // 16 + 7) // 8 * 8 for n in partition_sizes)
    d.group_size = K // (sz.numel() // (tsz * 16))   # rows of the packed scale/zero array = K / group_size
    with torch.cuda.device(wq.device):
        nat.check(lib.paro_dequant_packed(ctypes.byref(d), out.data_ptr(), nat.current_stream_ptr(wq.device)))
    return out


def w4a16_gemv_fused(x: torch.Tensor, pk, prologue: int = 0, eps: float = 1e-6, residual: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, allreduce=None,
                     parts_out: Optional[torch.Tensor] = None, parts_in: Optional[torch.Tensor] = None,
                     x_out: Optional[torch.Tensor] = None, parts_n: int = 0, attn_in: Optional[torch.Tensor] = None,
                     attn_head_dim: int = 0, dtype: Optional[torch.dtype] = None, attn_tail: Optional[dict] = None) -> torch.Tensor:
    """Decode-layer fusions around one fused linear (``paro_w4a16_gemv_fused``; rows <= 4):
    ``prologue`` = nat.PROLOGUE_RMSNORM  -> ``y = linear(x) * rsqrt(mean(x^2) + eps)`` (norm weight pre-folded into
    ``pk.channel_scales``, see ``PackedParoWeights.fold_norm_weight``), nat.PROLOGUE_SILU_MUL -> x is the merged
    gate_up output ``[rows, 2 K]`` and the linear consumes ``silu(gate) * up``; ``residual [rows, N]`` is added to
    the output (nat.PROLOGUE_GELU_TANH_MUL: ``gelu_tanh(gate) * up``, the Gemma MLP).  ``out`` may be given (e.g. a static buffer of a captured decode step).  ``allreduce`` (a
    ``paroquant_amd.tp.OneShotAllReduce``): ``pk`` is a row-parallel shard and the output becomes the sum over the ranks
    (+ bias + residual, once), exchanged inside this launch -- one row, every rank issuing the same launches.

    Deferred K-split reduction (one row; include/paro_abi.h, v12): ``parts_out`` (float32 ``[N, 4]``) -- the launch's K-splits
    (``parts_n`` of them; 0 = :func:`gemv_parts_count`) leave their partial sums there, in the in-launch reducer's summation
    order, unused slots zero, instead of reducing them in the launch; nothing else is written and ``parts_out`` is returned.
    With the RMSNorm prologue the buffer is ``[N + 1, 4]``: the partial sums are UN-normalised and the last row receives the
    K-slices' sums of squares -- the consumer applies ``rsqrt(sum / K + eps)`` (:func:`attn_decode` does for the qkv projection).
    ``parts_in`` (float32 ``[K, 4]``, a producer's ``parts_out``): ``x`` is the residual stream BEFORE the producer's output
    and the kernel completes ``x' = round(x + sum(parts_in, 1))`` while it seeds its rotation (prologue NONE or RMSNORM; a plain
    launch may K-split and may itself leave ``parts_out``); ``x_out [K]`` (not aliasing ``x``) receives ``x'``.

    ``attn_in`` (float32 ``[attn_parts_floats(K // attn_head_dim, attn_head_dim)]``, v14): the linear's input is the attention output
    handed over un-merged by :func:`attn_decode` ``(..., split_out=)``; ``x`` is ignored (pass ``None``), the activation type comes from
    ``dtype`` (or ``out``).  The launch completes the merge over the slots while it seeds its rotation -- the same bits as
    :func:`attn_finish` followed by the plain launch.  One row, no prologue, no residual; ``parts_out`` allowed.

    ``attn_tail`` (v18; a dict of :func:`attn_decode`'s arguments: kcache, vcache, pos, rope, n_heads, n_kv_heads, head_dim, q_norm_w,
    k_norm_w, eps, split_out, workspace): this is the qkv projection (one row, RMSNorm prologue, ``parts_out``) and the decode attention
    that consumes it runs in the SAME launch (include/paro_abi.h, ``paro_attn_tail_t``; :func:`attn_tail_supported` says when).
    ``parts_out`` is then float32 ``[N + 1, 8]`` -- 8-byte {partial sum, launch tag} granules -- and the result is ``split_out``."""
    lib = nat.load()
    K, N = pk.K, pk.N
    if attn_in is not None:
        dt = dtype or (out.dtype if out is not None else torch.float16)
        if attn_in.dtype != torch.float32 or not attn_in.is_contiguous() or attn_head_dim <= 0 or K % attn_head_dim \
                or attn_in.numel() != attn_parts_floats(K // attn_head_dim, attn_head_dim):
            raise ValueError(f"attn_in must be the contiguous float32 slot buffer of {K // max(attn_head_dim, 1)} heads x {attn_head_dim}")
        x = torch.empty((1, K), dtype=dt, device=attn_in.device) if x is None else x     # never read: shape / dtype / device carrier
    width = 2 * K if prologue in (nat.PROLOGUE_SILU_MUL, nat.PROLOGUE_GELU_TANH_MUL) else K
    if x.size(-1) != width:
        raise ValueError(f"x must have {width} columns for this prologue, got {x.size(-1)}")
    x2 = x.reshape(-1, width)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.size(0)
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"expected float16 or bfloat16 activations, got {x.dtype}")
    # (with the RMSNorm prologue a producer also leaves its K-slices' sums of squares: one more row)
    for name, t, width_t, slots in (("parts_out", parts_out, N + (1 if prologue == nat.PROLOGUE_RMSNORM else 0), nat.PARO_MAX_PARTIALS * (2 if attn_tail else 1)),
                                    ("parts_in", parts_in, K, nat.PARO_MAX_PARTIALS)):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (width_t, slots) or not t.is_contiguous()
                              or t.device != x.device):
            raise ValueError(f"{name} must be a contiguous float32 [{width_t}, {slots}] tensor on {x.device}")
    if x_out is not None and (parts_in is None or x_out.numel() != K or x_out.dtype != x.dtype or not x_out.is_contiguous() or x_out.device != x.device):
        raise ValueError(f"x_out (with parts_in) must be a contiguous [{K}] tensor of {x.dtype} on {x.device}")
    if parts_out is not None:
        y = None
    else:
        y = out if out is not None else torch.empty((rows, N), dtype=x.dtype, device=x.device)
    if out is not None and (out.numel() != rows * N or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device):
        raise ValueError(f"out must be a contiguous [{rows}, {N}] tensor of {x.dtype} on {x.device} (the kernel writes through its raw pointer)")
    if residual is not None and (residual.dtype != x.dtype or residual.numel() != rows * N or not residual.is_contiguous()):
        raise ValueError("residual must be a contiguous [rows, N] tensor of the activation dtype")
    d = make_desc(K, pk.partition_sizes, int(pk.pairs.size(1)), x.dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                  pk.channel_scales, bias if bias is not None else pk.bias, pk.wq_order, launch_hint=getattr(pk, "launch_hint", 0))
    f = nat.ParoFusion()
    f.prologue, f.eps, f.x_stride = int(prologue), float(eps), int(x2.stride(0))
    f.residual = residual.data_ptr() if residual is not None else None
    if allreduce is not None:
        f.ar_peers, f.ar_own, f.ar_state, f.ar_world, f.ar_rank, f.ar_max_elems = allreduce.fusion_args()
    if parts_out is not None:
        f.parts_out = parts_out.data_ptr()
        f.parts_out_n = int(parts_n) if parts_n else lib.paro_gemv_parts_count(ctypes.byref(d))
        if f.parts_out_n < 2:
            raise RuntimeError(f"parts_out: this layer does not K-split on its own (paro_gemv_parts_count = {f.parts_out_n}); "
                               "pass parts_n = 2..4 or use the ordinary route")
    if parts_in is not None:
        f.parts_in = parts_in.data_ptr()
        f.x_out = x_out.data_ptr() if x_out is not None else None
    if attn_in is not None:
        f.attn_in, f.attn_head_dim = attn_in.data_ptr(), int(attn_head_dim)
    tail = None
    if attn_tail is not None:
        t = attn_tail
        if parts_out is None or prologue != nat.PROLOGUE_RMSNORM or rows != 1:
            raise ValueError("attn_tail rides the one-row RMSNorm-prologue projection with parts_out")
        nh, nkv, hd = int(t["n_heads"]), int(t["n_kv_heads"]), int(t["head_dim"])
        so, aws, kc = t["split_out"], t["workspace"], t["kcache"]
        if so.dtype != torch.float32 or not so.is_contiguous() or so.numel() != attn_parts_floats(nh, hd):
            raise ValueError(f"attn_tail.split_out must be a contiguous float32 tensor of attn_parts_floats({nh}, {hd}) elements")
        tail = nat.ParoAttnTail()
        tail.kcache, tail.vcache, tail.attn_parts = kc.data_ptr(), t["vcache"].data_ptr(), so.data_ptr()
        tail.pos, tail.rope = t["pos"].data_ptr(), t["rope"].data_ptr()
        tail.q_norm_w = t["q_norm_w"].data_ptr() if t.get("q_norm_w") is not None else None
        tail.k_norm_w = t["k_norm_w"].data_ptr() if t.get("k_norm_w") is not None else None
        tail.eps, tail.scale = float(t.get("eps", 1e-6)), float(t.get("scale") or hd ** -0.5)
        tail.n_heads, tail.n_kv_heads, tail.head_dim, tail.max_positions = nh, nkv, hd, int(kc.size(1))
        tail.workspace, tail.workspace_bytes = aws.data_ptr(), aws.numel() * aws.element_size()
        f.attn_tail = ctypes.pointer(tail)
    ws = pk.workspace
    with torch.cuda.device(x.device):
        nat.check(lib.paro_w4a16_gemv_fused(ctypes.byref(d), x2.data_ptr(), y.data_ptr() if y is not None else None, rows, ws.data_ptr(),
                                            ws.numel() * ws.element_size(), ctypes.byref(f), nat.current_stream_ptr(x.device)))
    if tail is not None:
        return attn_tail["split_out"]
    return y if y is not None else parts_out


def attn_tail_supported(pk, n_heads: int, n_kv_heads: int, head_dim: int, max_positions: int, act_dtype: torch.dtype = torch.float16) -> bool:
    """Whether ``w4a16_gemv_fused(..., attn_tail=)`` can run this qkv projection with its decode attention in one launch
    (``paro_attn_tail_supported``: head_dim 128, at most 4 query heads per KV head, group_size 128, a deferred launch shape of 4 or 8
    waves that K-splits)."""
    n = nat.load().paro_attn_tail_supported(ctypes.byref(pk_desc(pk, act_dtype)), int(n_heads), int(n_kv_heads), int(head_dim), int(max_positions))
    if n < 0:
        nat.check(n)
    return n == 1


def gemv_parts_count(pk, act_dtype: torch.dtype = torch.float16) -> int:
    """How many fp32 partial sums ``w4a16_gemv_fused(..., parts_out=)`` of this layer leaves (``paro_gemv_parts_count``: the K-split
    of the launch shape chosen for a launch nobody polls in -- o / down 4, mid-width projections such as qkv 2); 0 = the layer does
    not split (or has a bias): use the ordinary route."""
    n = nat.load().paro_gemv_parts_count(ctypes.byref(pk_desc(pk, act_dtype)))
    if n < 0:
        nat.check(n)
    return int(n)


def parts_finish(parts: torch.Tensor, x: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``out[k] = round(x[k] + sum(parts[k, :]))`` in the order of the in-launch reducer (``paro_parts_finish``): completes a
    producer's ``parts_out [K, 4]`` when no linear follows (the last layer's down_proj in front of the final norm).  Not for the
    ``[N + 1, 4]`` buffer of an RMSNorm-prologue producer (its sums still lack the norm's scalar)."""
    dt = x.dtype if x is not None else (dtype or torch.float16)
    K = int(parts.size(0))
    if parts.dtype != torch.float32 or parts.dim() != 2 or parts.size(1) != nat.PARO_MAX_PARTIALS or not parts.is_contiguous() \
            or (x is not None and (x.numel() != K or not x.is_contiguous())):
        raise ValueError(f"parts must be contiguous float32 [K, {nat.PARO_MAX_PARTIALS}]; x contiguous [K]")
    if out is None:
        out = torch.empty(K, dtype=dt, device=parts.device)
    elif out.numel() != K or out.dtype != dt or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous [{K}] tensor of {dt}")
    with torch.cuda.device(parts.device):
        nat.check(nat.load().paro_parts_finish(x.data_ptr() if x is not None else None, parts.data_ptr(), K, out.data_ptr(),
                                               nat.DTYPE_F16 if dt == torch.float16 else nat.DTYPE_BF16, nat.current_stream_ptr(parts.device)))
    return out


def pk_desc(pk, act_dtype: torch.dtype, bias=None, rmat=None) -> nat.ParoLinearDesc:
    """``paro_linear_t`` of a :class:`paroquant_amd.linear.PackedParoWeights`."""
    return make_desc(pk.K, pk.partition_sizes, int(pk.pairs.size(1)), act_dtype, pk.wq, pk.sz, pk.rot, pk.pairs, pk.theta,
                     pk.channel_scales, bias if bias is not None else pk.bias, pk.wq_order, rmat, launch_hint=int(getattr(pk, "launch_hint", 0)))


def rotate_parts(x: torch.Tensor, pk, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Head of a decode chain: ``x [rows, K]`` rotated with every merged partition's parameters of ``pk`` in one launch
    -> ``[n_parts, rows, K]`` (``paro_rotate_parts``; the stage kernel behind ``rotation::rotate``, from 256 rows on the dense
    per-group product on the matrix cores -- the prefill GEMM's pre-pass on its own)."""
    lib = nat.load()
    K, P = pk.K, len(pk.partition_sizes)
    x2 = x.reshape(-1, K).contiguous()
    rows = x2.size(0)
    rmat = pk.rotation_matrices(x.dtype) if rows >= 256 else None
    if out is None:
        out = torch.empty((P, rows, K), dtype=x.dtype, device=x.device)
    elif out.numel() != P * rows * K or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous [{P}, {rows}, {K}] tensor of {x.dtype}")
    d = pk_desc(pk, x.dtype, rmat=rmat)
    with torch.cuda.device(x.device):
        nat.check(lib.paro_rotate_parts(ctypes.byref(d), x2.data_ptr(), out.data_ptr(), rows, nat.current_stream_ptr(x.device)))
    return out


def chain_gemv(x_rot: torch.Tensor, pk, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               ssq_in: Optional[torch.Tensor] = None, norm_dim: int = 0, eps: float = 1e-6, ssq_out: Optional[torch.Tensor] = None,
               next_pk=None, next_x: Optional[torch.Tensor] = None, next_col0: int = 0, act: int = 0, write_y: bool = True,
               ksplit: int = 0, waves: int = 0, bias: Optional[torch.Tensor] = None):
    """One linear of a decode chain (``paro_w4a16_gemv_chain``): ``x_rot [n_parts, rows, K]`` arrives rotated;
    ``y = x_rot @ dequant(W) * rstd + bias + residual`` with ``rstd = rsqrt(sum(ssq_in) / norm_dim + eps)`` when
    ``ssq_in [rows, blocks]`` (a producer's ``ssq_out``) is given; ``ssq_out [rows, N / 128]`` receives the blocks' sums of
    squares of y; with ``next_pk`` the launch also writes ``next_x [next.n_parts, rows, next.K]`` =
    ``rotate_next(act(y[:, next_col0 : next_col0 + next.K]))`` (``act`` = nat.CHAIN_ACT_SILU_MUL: ``pk`` is the merged
    gate|up projection and the consumer reads ``silu(gate) * up``).  Returns ``(y, next_x)``; ``y`` is None when
    ``write_y`` is False (only the consumer reads the result).  rows <= 16."""
    lib = nat.load()
    K, N, P = pk.K, pk.N, len(pk.partition_sizes)
    if x_rot.dim() != 3 or x_rot.size(0) != P or x_rot.size(2) != K or not x_rot.is_contiguous():
        raise ValueError(f"x_rot must be a contiguous [n_parts = {P}, rows, K = {K}] tensor, got {tuple(x_rot.shape)}")
    if x_rot.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"expected float16 or bfloat16 activations, got {x_rot.dtype}")
    rows = x_rot.size(1)
    dt, dev = x_rot.dtype, x_rot.device
    y = None
    if write_y:
        y = out if out is not None else torch.empty((rows, N), dtype=dt, device=dev)
        if y.numel() != rows * N or y.dtype != dt or not y.is_contiguous():
            raise ValueError(f"out must be a contiguous [{rows}, {N}] tensor of {dt}")
    if residual is not None and (residual.dtype != dt or residual.numel() != rows * N or not residual.is_contiguous()):
        raise ValueError("residual must be a contiguous [rows, N] tensor of the activation dtype")
    c = nat.ParoChain()
    c.x_rot = x_rot.data_ptr()
    c.y = y.data_ptr() if y is not None else None
    c.residual = residual.data_ptr() if residual is not None else None
    if ssq_in is not None:
        if ssq_in.dtype != torch.float32 or ssq_in.dim() != 2 or ssq_in.size(0) != rows or not ssq_in.is_contiguous():
            raise ValueError("ssq_in must be a contiguous fp32 [rows, blocks] tensor")
        c.ssq_in, c.ssq_in_blocks, c.norm_dim, c.eps = ssq_in.data_ptr(), int(ssq_in.size(1)), int(norm_dim), float(eps)
    if ssq_out is not None:
        if ssq_out.dtype != torch.float32 or ssq_out.numel() != rows * (N // 128) or not ssq_out.is_contiguous():
            raise ValueError(f"ssq_out must be a contiguous fp32 [rows, {N // 128}] tensor")
        c.ssq_out = ssq_out.data_ptr()
    d = pk_desc(pk, dt, bias)
    dn = None
    if next_pk is not None:
        Pn, Kn = len(next_pk.partition_sizes), next_pk.K
        if next_x is None:
            next_x = torch.empty((Pn, rows, Kn), dtype=dt, device=dev)
        elif next_x.numel() != Pn * rows * Kn or next_x.dtype != dt or not next_x.is_contiguous():
            raise ValueError(f"next_x must be a contiguous [{Pn}, {rows}, {Kn}] tensor of {dt}")
        dn = pk_desc(next_pk, dt)
        c.next = ctypes.pointer(dn)
        c.next_x_rot, c.next_col0, c.next_act = next_x.data_ptr(), int(next_col0), int(act)
    elif act:
        raise ValueError("act without next_pk")
    ws = pk.workspace
    need = lib.paro_chain_workspace_bytes(ctypes.byref(d), rows)
    if ws.numel() * ws.element_size() < need:
        ws = get_workspace(dev, need)
    with torch.cuda.device(dev):
        nat.check(lib.paro_w4a16_gemv_chain(ctypes.byref(d), ctypes.byref(c), rows, ws.data_ptr(), ws.numel() * ws.element_size(),
                                            int(ksplit), int(waves), nat.current_stream_ptr(dev)))
    return y, (next_x if next_pk is not None else None)


_attn_ws: dict = {}


def attn_workspace(device, n_heads: int, n_kv_heads: int, head_dim: int, max_positions: int) -> torch.Tensor:
    """Zero-filled scratch of ``paro_attn_decode`` (arrival tickets + partial results of the position chunks); one per
    (device, geometry), shared by every caller that does not pass its own -- fine for launches on ONE stream; anything
    that may overlap in time (two models, two streams) must own its workspace (``ParoDecoderLM`` does)."""
    lib = nat.load()
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), n_heads, n_kv_heads, head_dim, max_positions)
    ws = _attn_ws.get(key)
    if ws is None:
        need = lib.paro_attn_decode_workspace_bytes(n_heads, n_kv_heads, head_dim, max_positions)
        if need < 0:
            raise ValueError("bad attention geometry")
        ws = torch.zeros(need, dtype=torch.uint8, device=device)
        _attn_ws[key] = ws
    return ws


def attn_parts_floats(n_heads: int, head_dim: int) -> int:
    """Elements of the float32 slot buffer of a split attention launch (``paro_attn_parts_floats``)."""
    n = nat.load().paro_attn_parts_floats(int(n_heads), int(head_dim))
    if n < 0:
        raise ValueError("bad n_heads / head_dim")
    return int(n)


def attn_finish(attn_parts: torch.Tensor, n_heads: int, head_dim: int, dtype: torch.dtype = torch.float16,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Complete a split attention launch's slots into the attention output ``[n_heads * head_dim]`` (``paro_attn_finish``): what the
    ``attn_in`` prologue of :func:`w4a16_gemv_fused` computes, as its own launch."""
    if attn_parts.dtype != torch.float32 or not attn_parts.is_contiguous() or attn_parts.numel() != attn_parts_floats(n_heads, head_dim):
        raise ValueError("attn_parts must be the contiguous float32 slot buffer of a split attention launch")
    y = out if out is not None else torch.empty(n_heads * head_dim, dtype=dtype, device=attn_parts.device)
    if y.numel() != n_heads * head_dim or y.dtype != dtype or not y.is_contiguous():
        raise ValueError(f"out must be a contiguous [{n_heads * head_dim}] tensor of {dtype}")
    with torch.cuda.device(attn_parts.device):
        nat.check(nat.load().paro_attn_finish(attn_parts.data_ptr(), int(n_heads), int(head_dim), y.data_ptr(), nat.dtype_code(dtype),
                                              nat.current_stream_ptr(attn_parts.device)))
    return y


# ---- prompt pass of the decode harness (ABI v19, csrc/prompt.hip): the element-wise work between a layer's four fused linears at T rows

def prompt_row_rms(h: torch.Tensor, eps: float) -> torch.Tensor:
    """``rs[t] = rsqrt(mean(h[t]^2) + eps)``, float32 ``[T]`` (``paro_prompt_row_rms``): the scalar an RMSNorm leaves once its weight is
    folded into the consumer's channel scales."""
    if h.dim() != 2 or not h.is_contiguous() or h.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("h must be a contiguous [T, hidden] fp16 / bf16 tensor")
    rs = torch.empty(h.size(0), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        nat.check(nat.load().paro_prompt_row_rms(h.data_ptr(), rs.data_ptr(), int(h.size(0)), int(h.size(1)), float(eps), nat.dtype_code(h.dtype),
                                                 nat.current_stream_ptr(h.device)))
    return rs


def prompt_qkv_post(qkv: torch.Tensor, rs: Optional[torch.Tensor], rope: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor,
                    n_heads: int, n_kv_heads: int, head_dim: int, q_norm_w=None, k_norm_w=None, eps: float = 1e-6, pos0: int = 0):
    """The merged qkv projection's raw output ``[T, (n_heads + 2 n_kv_heads) head_dim]`` -> row scale, q / k head RMSNorm, rotary embedding of
    positions ``pos0 + t`` -> ``(q [T, n_heads, hd], k [T, n_kv_heads, hd], v [T, n_kv_heads, hd])`` and the decode caches
    (``kcache [n_kv_heads, T_max, hd]``, ``vcache [n_kv_heads, hd, T_max]``) in one launch (``paro_prompt_qkv_post``)."""
    T = int(qkv.size(0))
    dt = qkv.dtype
    if qkv.dim() != 2 or not qkv.is_contiguous() or qkv.size(1) != (n_heads + 2 * n_kv_heads) * head_dim or dt not in (torch.float16, torch.bfloat16):
        raise ValueError("qkv must be a contiguous [T, (n_heads + 2 n_kv_heads) head_dim] fp16 / bf16 tensor")
    T_max = int(kcache.size(1))
    if tuple(kcache.shape) != (n_kv_heads, T_max, head_dim) or tuple(vcache.shape) != (n_kv_heads, head_dim, T_max) or kcache.dtype != dt \
            or vcache.dtype != dt or not kcache.is_contiguous() or not vcache.is_contiguous():
        raise ValueError("kcache [n_kv_heads, T_max, head_dim] / vcache [n_kv_heads, head_dim, T_max] must be contiguous tensors of qkv's dtype")
    if rope.dtype != torch.float32 or not rope.is_contiguous() or rope.dim() != 2 or rope.size(1) != head_dim or rope.size(0) < pos0 + T:
        raise ValueError("rope must be a contiguous float32 [positions, head_dim] table (cos | sin halves) covering pos0 + T")
    for w in (q_norm_w, k_norm_w):
        if w is not None and (w.dtype != dt or w.numel() != head_dim or not w.is_contiguous()):
            raise ValueError("q_norm_w / k_norm_w must be contiguous [head_dim] tensors of qkv's dtype")
    if rs is not None and (rs.dtype != torch.float32 or rs.numel() != T or not rs.is_contiguous()):
        raise ValueError("rs must be a contiguous float32 [T] tensor")
    q = torch.empty(T, n_heads, head_dim, dtype=dt, device=qkv.device)
    k = torch.empty(T, n_kv_heads, head_dim, dtype=dt, device=qkv.device)
    v = torch.empty(T, n_kv_heads, head_dim, dtype=dt, device=qkv.device)
    with torch.cuda.device(qkv.device):
        nat.check(nat.load().paro_prompt_qkv_post(qkv.data_ptr(), rs.data_ptr() if rs is not None else None,
                                                  q_norm_w.data_ptr() if q_norm_w is not None else None,
                                                  k_norm_w.data_ptr() if k_norm_w is not None else None, rope.data_ptr(), q.data_ptr(), k.data_ptr(),
                                                  v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), T, int(pos0), int(n_heads), int(n_kv_heads),
                                                  int(head_dim), T_max, float(eps), nat.dtype_code(dt), nat.current_stream_ptr(qkv.device)))
    return q, k, v


def prompt_silu_mul(gate_up: torch.Tensor, rs: Optional[torch.Tensor]) -> torch.Tensor:
    """``silu(gate * rs) * (up * rs)`` of the merged gate_up projection's raw output ``[T, 2 I]`` -> ``[T, I]`` (``paro_prompt_silu_mul``)."""
    if gate_up.dim() != 2 or not gate_up.is_contiguous() or gate_up.size(1) % 16 or gate_up.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("gate_up must be a contiguous [T, 2 I] fp16 / bf16 tensor, I a multiple of 8")
    T, I = int(gate_up.size(0)), int(gate_up.size(1)) // 2
    if rs is not None and (rs.dtype != torch.float32 or rs.numel() != T or not rs.is_contiguous()):
        raise ValueError("rs must be a contiguous float32 [T] tensor")
    out = torch.empty(T, I, dtype=gate_up.dtype, device=gate_up.device)
    with torch.cuda.device(gate_up.device):
        nat.check(nat.load().paro_prompt_silu_mul(gate_up.data_ptr(), rs.data_ptr() if rs is not None else None, out.data_ptr(), T, I,
                                                  nat.dtype_code(gate_up.dtype), nat.current_stream_ptr(gate_up.device)))
    return out


def attn_decode(qkv: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, pos: torch.Tensor, rope: torch.Tensor,
                n_heads: int, n_kv_heads: int, head_dim: int, q_norm_w=None, k_norm_w=None, eps: float = 1e-6,
                out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, norm_dim: int = 0,
                norm_eps: float = 1e-6, split_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One decoder layer's batch-1 attention in one launch (``paro_attn_decode``): q/k norm + RoPE + KV-cache append at
    ``pos`` (int32 device tensor) + GQA over positions 0..pos.  ``kcache``: [n_kv_heads, T_max, head_dim];
    ``vcache``: [n_kv_heads, head_dim, T_max] (position-contiguous); T_max a multiple of 8; both must hold finite values
    (allocate them zero-filled).  ``qkv`` is the merged projection's output in the cache dtype, or -- float32
    ``[(n_heads + 2 n_kv_heads) * head_dim + 1, 4]`` -- the partial sums a K-split qkv projection left
    (``w4a16_gemv_fused(..., parts_out=)``; ``paro_attn_decode_parts``): the kernel completes each element as it reads it, scaled by
    ``rsqrt(sum(last row) / norm_dim + norm_eps)`` when ``norm_dim > 0`` (the projection ran with the RMSNorm prologue).

    ``split_out`` (float32 ``[attn_parts_floats(n_heads, head_dim)]``, zero-filled once; ``paro_attn_decode_split``, v14): the merge over
    position chunks is left to the consumer -- the launch stores per slot (at most four) the un-normalised outputs, maxima and sums;
    ``w4a16_gemv_fused(None, o_proj, attn_in=split_out, attn_head_dim=head_dim)`` or :func:`attn_finish` completes them.  Returns
    ``split_out``."""
    lib = nat.load()
    T_max = kcache.size(1)
    parts = qkv.dtype == torch.float32
    act_dtype = kcache.dtype
    if tuple(kcache.shape) != (n_kv_heads, T_max, head_dim) or tuple(vcache.shape) != (n_kv_heads, head_dim, T_max) \
            or not kcache.is_contiguous() or not vcache.is_contiguous() or vcache.dtype != act_dtype or (not parts and qkv.dtype != act_dtype):
        raise ValueError(f"caches must be contiguous {act_dtype} tensors: K [{n_kv_heads}, T, {head_dim}], V [{n_kv_heads}, {head_dim}, T] "
                         f"(got K {tuple(kcache.shape)}, V {tuple(vcache.shape)})")
    if tuple(rope.shape) != (T_max, head_dim) or rope.dtype != torch.float32 or pos.dtype != torch.int32:
        raise ValueError("rope must be fp32 [T, head_dim] (cos then sin per position) and pos an int32 device scalar")
    split = split_out is not None
    if split and (split_out.dtype != torch.float32 or not split_out.is_contiguous() or split_out.device != qkv.device
                  or split_out.numel() != attn_parts_floats(n_heads, head_dim)):
        raise ValueError(f"split_out must be a contiguous float32 tensor of attn_parts_floats({n_heads}, {head_dim}) elements on {qkv.device}")
    y = out if (out is not None or split) else torch.empty(n_heads * head_dim, dtype=act_dtype, device=qkv.device)
    if out is not None and (out.numel() != n_heads * head_dim or out.dtype != act_dtype or not out.is_contiguous() or out.device != qkv.device):
        raise ValueError(f"out must be a contiguous tensor of {n_heads * head_dim} {act_dtype} elements on {qkv.device}")
    n_qkv = (n_heads + 2 * n_kv_heads) * head_dim
    if parts:
        if tuple(qkv.shape) != (n_qkv + 1, nat.PARO_MAX_PARTIALS) or not qkv.is_contiguous():
            raise ValueError(f"partial sums of qkv must be a contiguous float32 [{n_qkv + 1}, {nat.PARO_MAX_PARTIALS}] tensor")
    elif qkv.numel() != n_qkv or not qkv.is_contiguous():
        raise ValueError(f"qkv must be a contiguous vector of (n_heads + 2 n_kv_heads) * head_dim = {n_qkv} elements")
    ws = workspace if workspace is not None else attn_workspace(qkv.device, n_heads, n_kv_heads, head_dim, T_max)
    tail = (kcache.data_ptr(), vcache.data_ptr(), split_out.data_ptr() if split else y.data_ptr(), pos.data_ptr(), rope.data_ptr(), None if q_norm_w is None else q_norm_w.data_ptr(),
            None if k_norm_w is None else k_norm_w.data_ptr(), float(eps), float(head_dim) ** -0.5, n_heads, n_kv_heads, head_dim, T_max,
            nat.dtype_code(act_dtype), ws.data_ptr(), ws.numel(), nat.current_stream_ptr(qkv.device))
    with torch.cuda.device(qkv.device):
        if split:
            nat.check(lib.paro_attn_decode_split(None if parts else qkv.data_ptr(), qkv.data_ptr() if parts else None, int(norm_dim), float(norm_eps), *tail))
            return split_out
        if parts:
            nat.check(lib.paro_attn_decode_parts(qkv.data_ptr(), int(norm_dim), float(norm_eps), *tail))
        else:
            nat.check(lib.paro_attn_decode(qkv.data_ptr(), *tail))
    return y


def lm_head_workspace(device, vocab: int) -> torch.Tensor:
    lib = nat.load()
    return torch.zeros(lib.paro_lm_head_workspace_bytes(int(vocab)), dtype=torch.uint8, device=device)


def lm_head(x: torch.Tensor, norm_weight: torch.Tensor, weight: torch.Tensor, logits: torch.Tensor, eps: float,
            workspace: torch.Tensor) -> torch.Tensor:
    """``logits = lm_head(rmsnorm(x))`` for one token (``paro_lm_head``): x [hidden], weight [vocab, hidden] contiguous."""
    lib = nat.load()
    V, H = weight.shape
    with torch.cuda.device(x.device):
        nat.check(lib.paro_lm_head(x.data_ptr(), norm_weight.data_ptr(), weight.data_ptr(), logits.data_ptr(), V, H, float(eps),
                                   nat.dtype_code(x.dtype), workspace.data_ptr(), workspace.numel(), nat.current_stream_ptr(x.device)))
    return logits


def argmax_advance(workspace: torch.Tensor, vocab: int, token: torch.Tensor, pos: torch.Tensor, out_tokens: Optional[torch.Tensor]) -> None:
    """Greedy next token from the partial maxima ``lm_head`` left in ``workspace``: ``out_tokens[pos] = token``;
    ``token = argmax``; ``pos += 1`` -- all on the device (``paro_argmax_advance``)."""
    lib = nat.load()
    if token.dtype != torch.int64 or pos.dtype != torch.int32 or (out_tokens is not None and out_tokens.dtype != torch.int64):
        raise RuntimeError("token / out_tokens must be int64 and pos int32")
    with torch.cuda.device(token.device):
        nat.check(lib.paro_argmax_advance(workspace.data_ptr(), int(vocab), token.data_ptr(), pos.data_ptr(),
                                          None if out_tokens is None else out_tokens.data_ptr(),
                                          0 if out_tokens is None else out_tokens.numel(), nat.current_stream_ptr(token.device)))


_workspaces: dict = {}


_retired_workspaces = []


def get_workspace(device: torch.device, nbytes: int, stream=None) -> torch.Tensor:
    """Zero-initialised scratch shared by all layers that run on one (device, stream): split-K granules + arrival
    counters + status word, rotated activations of the unfused routes, fp32 partial tiles.

    The K-split granules are tagged with per-block epochs kept in the first 16 KiB (see ``paro_abi.h``), so one buffer
    serves every layer that runs on the same stream, decode and prefill alike; it only ever grows, and always with
    ``torch.zeros`` (epochs and the status word must start from a known state).  ``stream=None`` is the default workspace of the device: correct for any number of
    streams that use it ONE AT A TIME (eager warm-up stream, then a graph-capture stream).  Launches that may
    overlap in time on different streams must not share granules: give each such stream its own workspace with
    ``get_workspace(device, nbytes, stream)`` / ``PackedParoWeights.bind_stream(stream)``."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx, None if stream is None else int(stream.cuda_stream))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            # a HIP graph captured earlier holds the OLD buffer's address: it stays allocated (replays keep writing their granules and
            # rotated rows there), only new calls move to the larger one
            _retired_workspaces.append(ws)
            nbytes = max(int(nbytes), ws.numel() + ws.numel() // 2)     # grow geometrically: the retired buffers sum to < 3x the live one
        ws = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=torch.device(device.type, idx))
        _workspaces[key] = ws
    return ws


def check_workspace(ws: torch.Tensor) -> None:
    """Raise if a K-split launch on this workspace ever gave up waiting for a partial sum (its outputs were NaN) or
    the workspace was not zero-filled.  Synchronises: call it after graph capture / at teardown, not per token."""
    lib = nat.load()
    with torch.cuda.device(ws.device):
        nat.check(lib.paro_workspace_status(ws.data_ptr(), nat.current_stream_ptr(ws.device)))


def decode_workspace_bytes(K: int, N: int, n_parts: int, rows: int = 16) -> int:
    """Mirror of ``paro_linear_workspace_bytes`` for rows <= 16."""
    return nat.PARO_WS_COUNTER_BYTES + 16 * rows * N * 8 + n_parts * K * max(((rows + 1) // 2) * 8, 32)


# --------------------------------------------------------------------------------------
# registration
# --------------------------------------------------------------------------------------

_libs = []


def _register() -> None:
    rot = torch.library.Library("rotation", "DEF")
    rot.define(_ROTATE_SCHEMA)
    rot.impl("rotate", _rotate_impl, "CUDA")       # ROCm tensors dispatch under the CUDA key
    torch.library.register_fake("rotation::rotate", _rotate_fake, lib=rot)
    _libs.append(rot)

    par = torch.library.Library("paro", "DEF")
    par.define("repack_awq(Tensor qweight, Tensor qzeros, Tensor scales, int[] partition_sizes, int wq_order=0) "
               "-> (Tensor, Tensor)")
    par.impl("repack_awq", _repack_impl, "CUDA")
    torch.library.register_fake("paro::repack_awq", _repack_fake, lib=par)
    par.define("pack_rotation(Tensor pairs, Tensor theta) -> Tensor")
    par.impl("pack_rotation", _pack_rotation_impl, "CUDA")
    torch.library.register_fake("paro::pack_rotation", _pack_rotation_fake, lib=par)
    par.define("w4a16_linear(Tensor x, Tensor wq, Tensor sz, Tensor rot, Tensor pairs, Tensor theta, "
               "Tensor channel_scales, Tensor? bias, int[] partition_sizes, Tensor workspace, int wq_order=0, "
               "Tensor? rmat=None, int launch_hint=0) -> Tensor")
    par.impl("w4a16_linear", _w4a16_impl, "CUDA")
    torch.library.register_fake("paro::w4a16_linear", _w4a16_fake, lib=par)
    _libs.append(par)


_register()

"""ctypes binding of ``libparo_mi355x.so`` (C ABI declared in ``include/paro_abi.h``).

The library is hand-written HIP for gfx950, built in-tree by ``__graft_entry__.build()`` /
``make -C paroquant_amd/csrc``.  There is NO CPU or PyTorch fallback: if the shared object is
missing or a call fails, the product path raises (the reference behaves the same way -- its
``rotation::rotate`` exists for the CUDA dispatch key only, rotation.cu:133-135, and
``ParoQuantHfQuantizer.validate_environment`` raises without a GPU, transformers/quantizer.py:78-80).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_int, c_int32, c_int64, c_void_p

import torch  # noqa: F401  -- must be imported first so libamdhip64.so.7 resolves to torch's runtime

PARO_ABI_VERSION = 19
PARO_MAX_PARTS = 8
PARO_WS_COUNTER_BYTES = 16384
PARO_WS_STATUS_OFFSET = PARO_WS_COUNTER_BYTES - 4
PARO_WS_STATUS_GIVEUP = 0xDEAD
PARO_MAX_PARTIALS = 4
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2

# PARO_LIB_DIR: directory (relative to the package) of an experiment build of the library (A/B runs of kernel variants)
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PARO_LIB_DIR", "_lib"), "libparo_mi355x.so")

EXPORTS = (
    "paro_abi_version",
    "paro_last_error",
    "paro_rotate",
    "paro_packed_qweight_bytes",
    "paro_packed_sz_bytes",
    "paro_packed_rot_bytes",
    "paro_repack_awq",
    "paro_pack_rotation",
    "paro_linear_workspace_bytes",
    "paro_workspace_status",
    "paro_gemv_launch_shape",
    "paro_w4a16_gemv",
    "paro_w4a16_gemv_fused",
    "paro_w4a16_gemv_experts",
    "paro_attn_decode_workspace_bytes",
    "paro_attn_decode",
    "paro_attn_decode_parts",
    "paro_attn_parts_floats",
    "paro_attn_decode_split",
    "paro_attn_finish",
    "paro_lm_head_workspace_bytes",
    "paro_lm_head",
    "paro_argmax_advance",
    "paro_w4a16_gemm",
    "paro_w4a16_linear",
    "paro_w4a16_gemm_grouped",
    "paro_dequant_packed",
    "paro_chain_workspace_bytes",
    "paro_chain_launch_shape",
    "paro_w4a16_gemv_chain",
    "paro_rotate_parts",
    "paro_gemv_parts_count",
    "paro_attn_tail_supported",
    "paro_parts_finish",
    "paro_gemm_launch_shape",
    "paro_prompt_row_rms",
    "paro_prompt_qkv_post",
    "paro_prompt_silu_mul",
    "paro_allreduce_buffer_bytes",
    "paro_allreduce_buffer_create",
    "paro_allreduce_buffer_open",
    "paro_allreduce_buffer_close",
    "paro_allreduce_buffer_destroy",
    "paro_allreduce_status",
    "paro_allreduce_oneshot",
    "paro_gdn_prep",
    "paro_gdn_step",
    "paro_gdn_workspace_bytes",
    "paro_gdn_sequence",
    "paro_gdn_fused_step",
    "paro_attn_decode_gated",
)


# include/paro_abi_experimental.h: only in a library built with `make EXPERIMENTAL=1` (the persistent decode engines; measured slower
# than one launch per linear, profiles/NOTES.md 4.2 / 5.1).  Bound when present, never required.
EXPERIMENTAL_EXPORTS = (
    "paro_engine_plan",
    "paro_engine_build",
    "paro_engine_describe",
    "paro_engine_run",
    "paro_engine_trace",
    "paro_engine2_plan",
    "paro_engine2_build",
    "paro_engine2_describe",
    "paro_engine2_run",
    "paro_engine2_trace",
)


class ParoLinearDesc(Structure):
    """``paro_linear_t`` (include/paro_abi.h)."""

    _fields_ = [
        ("K", c_int64),
        ("N", c_int64),
        ("n_parts", c_int32),
        ("krot", c_int32),
        ("part_cols", c_int32 * PARO_MAX_PARTS),
        ("act_dtype", c_int32),
        ("wq_order", c_int32),
        ("wq", c_void_p),
        ("sz", c_void_p),
        ("rot", c_void_p),
        ("pairs", c_void_p),
        ("theta", c_void_p),
        ("channel_scales", c_void_p),
        ("bias", c_void_p),
        ("rmat", c_void_p),
        ("group_size", c_int32),
        ("launch_hint", c_int32),
    ]


class ParoAttnTail(Structure):
    """``paro_attn_tail_t`` (include/paro_abi.h, v18)."""

    _fields_ = [("kcache", c_void_p), ("vcache", c_void_p), ("attn_parts", c_void_p), ("pos", c_void_p), ("rope", c_void_p),
                ("q_norm_w", c_void_p), ("k_norm_w", c_void_p), ("eps", ctypes.c_float), ("scale", ctypes.c_float),
                ("n_heads", c_int32), ("n_kv_heads", c_int32), ("head_dim", c_int32), ("max_positions", c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_int64)]


class ParoFusion(Structure):
    """``paro_fusion_t`` (include/paro_abi.h)."""

    _fields_ = [("prologue", c_int32), ("eps", ctypes.c_float), ("x_stride", c_int64), ("residual", c_void_p),
                ("ar_peers", c_void_p), ("ar_own", c_void_p), ("ar_state", c_void_p), ("ar_world", c_int32), ("ar_rank", c_int32), ("ar_max_elems", c_int64),
                ("parts_out", c_void_p), ("parts_in", c_void_p), ("x_out", c_void_p), ("parts_out_n", c_int32), ("attn_head_dim", c_int32),
                ("attn_in", c_void_p), ("attn_tail", POINTER(ParoAttnTail))]


class ParoExperts(Structure):
    """``paro_experts_t`` (include/paro_abi.h)."""

    _fields_ = [("expert_idx", c_void_p), ("n_slots", c_int32), ("x_slot_div", c_int32), ("wq_stride_bytes", c_int64),
                ("sz_stride_bytes", c_int64), ("x_slot_stride", c_int64), ("y_slot_stride", c_int64), ("n_experts", c_int32),
                ("reserved0", c_int32)]


class ParoEnginePhase(Structure):
    """``paro_engine_phase_t`` (include/paro_abi.h)."""

    _fields_ = [("L", POINTER(ParoLinearDesc)), ("in_col0", c_int64), ("flags", c_int32), ("reserved0", c_int32)]


class ParoEngine(Structure):
    """``paro_engine_t`` (include/paro_abi.h)."""

    _fields_ = [("n_phases", c_int32), ("n_cus", c_int32), ("act_dtype", c_int32), ("last_split", c_int32), ("plan_bytes", c_int64),
                ("workspace_bytes", c_int64), ("in_features", c_int64), ("out_features", c_int64), ("last_out_offset", c_int64),
                ("last_bias", c_void_p), ("n_shapes", c_int32), ("shape_off", c_int32 * 8), ("reserved0", c_int32)]


class ParoChain(Structure):
    """``paro_chain_t`` (include/paro_abi.h)."""

    _fields_ = [("x_rot", c_void_p), ("y", c_void_p), ("residual", c_void_p), ("ssq_in", c_void_p), ("ssq_in_blocks", c_int32),
                ("eps", ctypes.c_float), ("norm_dim", c_int64), ("ssq_out", c_void_p), ("next", POINTER(ParoLinearDesc)),
                ("next_x_rot", c_void_p), ("next_col0", c_int64), ("next_act", c_int32), ("reserved0", c_int32)]


PROLOGUE_NONE, PROLOGUE_RMSNORM, PROLOGUE_SILU_MUL, PROLOGUE_GELU_TANH_MUL = 0, 1, 2, 3
CHAIN_ACT_NONE, CHAIN_ACT_SILU_MUL, CHAIN_ACT_GELU_TANH_MUL = 0, 1, 2

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the shared object (once) and declare every prototype; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"paroquant_amd: native library not found at {_LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C paroquant_amd/csrc`. "
            "There is no CPU fallback for the ParoQuant hot path."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    lib.paro_abi_version.restype = c_int
    lib.paro_abi_version.argtypes = []
    lib.paro_last_error.restype = c_char_p
    lib.paro_last_error.argtypes = []
    lib.paro_rotate.restype = c_int
    lib.paro_rotate.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                c_int, c_int, c_void_p]
    lib.paro_packed_qweight_bytes.restype = c_int64
    lib.paro_packed_qweight_bytes.argtypes = [c_int64, c_int64]
    lib.paro_packed_sz_bytes.restype = c_int64
    lib.paro_packed_sz_bytes.argtypes = [c_int64, c_int, c_int, POINTER(c_int32)]
    lib.paro_packed_rot_bytes.restype = c_int64
    lib.paro_packed_rot_bytes.argtypes = [c_int64, c_int]
    lib.paro_repack_awq.restype = c_int
    lib.paro_repack_awq.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, POINTER(c_int32), c_int,
                                    c_void_p, c_void_p, c_void_p]
    lib.paro_pack_rotation.restype = c_int
    lib.paro_pack_rotation.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.paro_linear_workspace_bytes.restype = c_int64
    lib.paro_linear_workspace_bytes.argtypes = [POINTER(ParoLinearDesc), c_int64]
    lib.paro_gemv_launch_shape.restype = c_int
    lib.paro_gemv_launch_shape.argtypes = [POINTER(ParoLinearDesc), c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                           POINTER(c_int)]
    lib.paro_w4a16_gemv.restype = c_int
    lib.paro_w4a16_gemv.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                    c_int, c_int, c_int, c_void_p]
    lib.paro_w4a16_gemv_fused.restype = c_int
    lib.paro_w4a16_gemv_fused.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                          POINTER(ParoFusion), c_void_p]
    lib.paro_w4a16_gemv_experts.restype = c_int
    lib.paro_w4a16_gemv_experts.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                            POINTER(ParoFusion), POINTER(ParoExperts), c_void_p]
    lib.paro_attn_decode_workspace_bytes.restype = c_int64
    lib.paro_attn_decode_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.paro_attn_decode.restype = c_int
    lib.paro_attn_decode.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     ctypes.c_float, ctypes.c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                     c_void_p]
    lib.paro_attn_decode_parts.restype = c_int
    lib.paro_attn_decode_parts.argtypes = [c_void_p, c_int64, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           ctypes.c_float, ctypes.c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]
    lib.paro_attn_parts_floats.restype = c_int64
    lib.paro_attn_parts_floats.argtypes = [c_int, c_int]
    lib.paro_attn_decode_split.restype = c_int
    lib.paro_attn_decode_split.argtypes = [c_void_p, c_void_p, c_int64, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           ctypes.c_float, ctypes.c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]
    lib.paro_attn_finish.restype = c_int
    lib.paro_attn_finish.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]
    lib.paro_lm_head_workspace_bytes.restype = c_int64
    lib.paro_lm_head_workspace_bytes.argtypes = [c_int64]
    lib.paro_lm_head.restype = c_int
    lib.paro_lm_head.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, ctypes.c_float, c_int, c_void_p,
                                 c_int64, c_void_p]
    lib.paro_argmax_advance.restype = c_int
    lib.paro_argmax_advance.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.paro_w4a16_gemm.restype = c_int
    lib.paro_w4a16_gemm.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                    c_void_p]
    lib.paro_w4a16_gemm_grouped.restype = c_int
    lib.paro_w4a16_gemm_grouped.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_int32,
                                            c_void_p]
    lib.paro_workspace_status.restype = c_int
    lib.paro_workspace_status.argtypes = [c_void_p, c_void_p]
    lib.paro_w4a16_linear.restype = c_int
    lib.paro_w4a16_linear.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                      c_void_p]
    lib.paro_chain_workspace_bytes.restype = c_int64
    lib.paro_chain_workspace_bytes.argtypes = [POINTER(ParoLinearDesc), c_int64]
    lib.paro_chain_launch_shape.restype = c_int
    lib.paro_chain_launch_shape.argtypes = [POINTER(ParoLinearDesc), POINTER(ParoChain), c_int64, POINTER(c_int), POINTER(c_int)]
    lib.paro_w4a16_gemv_chain.restype = c_int
    lib.paro_w4a16_gemv_chain.argtypes = [POINTER(ParoLinearDesc), POINTER(ParoChain), c_int64, c_void_p, c_int64, c_int, c_int,
                                          c_void_p]
    lib.paro_rotate_parts.restype = c_int
    lib.paro_rotate_parts.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p, c_int64, c_void_p]
    lib.paro_gemv_parts_count.restype = c_int
    lib.paro_gemv_parts_count.argtypes = [POINTER(ParoLinearDesc)]
    lib.paro_attn_tail_supported.restype = c_int
    lib.paro_attn_tail_supported.argtypes = [POINTER(ParoLinearDesc), c_int, c_int, c_int, c_int]
    lib.paro_gemm_launch_shape.restype = c_int
    lib.paro_gemm_launch_shape.argtypes = [POINTER(ParoLinearDesc), c_int64, POINTER(c_int), POINTER(c_int)]
    lib.paro_prompt_row_rms.restype = c_int
    lib.paro_prompt_row_rms.argtypes = [c_void_p, c_void_p, c_int64, c_int64, ctypes.c_float, c_int, c_void_p]
    lib.paro_prompt_qkv_post.restype = c_int
    lib.paro_prompt_qkv_post.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int64, c_int, c_int, c_int, c_int, c_int, ctypes.c_float, c_int, c_void_p]
    lib.paro_prompt_silu_mul.restype = c_int
    lib.paro_prompt_silu_mul.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]
    lib.paro_parts_finish.restype = c_int
    lib.paro_parts_finish.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]
    lib.paro_allreduce_buffer_bytes.restype = c_int64
    lib.paro_allreduce_buffer_bytes.argtypes = [c_int, c_int64]
    lib.paro_allreduce_buffer_create.restype = c_int
    lib.paro_allreduce_buffer_create.argtypes = [c_int64, POINTER(c_void_p), c_void_p]
    lib.paro_allreduce_buffer_open.restype = c_int
    lib.paro_allreduce_buffer_open.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.paro_allreduce_buffer_close.restype = c_int
    lib.paro_allreduce_buffer_close.argtypes = [c_void_p]
    lib.paro_allreduce_buffer_destroy.restype = c_int
    lib.paro_allreduce_buffer_destroy.argtypes = [c_void_p]
    lib.paro_allreduce_status.restype = c_int
    lib.paro_allreduce_status.argtypes = [c_void_p, c_void_p]
    lib.paro_allreduce_oneshot.restype = c_int
    lib.paro_allreduce_oneshot.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]
    lib.paro_dequant_packed.restype = c_int
    lib.paro_dequant_packed.argtypes = [POINTER(ParoLinearDesc), c_void_p, c_void_p]
    f32 = ctypes.c_float
    lib.paro_gdn_prep.restype = c_int
    lib.paro_gdn_prep.argtypes = [c_void_p, c_void_p, c_void_p, f32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.paro_gdn_step.restype = c_int
    lib.paro_gdn_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, f32, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.paro_gdn_fused_step.restype = c_int
    lib.paro_gdn_fused_step.argtypes = [c_void_p, c_void_p, c_void_p, f32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, f32, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.paro_gdn_sequence.restype = c_int
    lib.paro_gdn_sequence.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.paro_gdn_workspace_bytes.restype = c_int64
    lib.paro_gdn_workspace_bytes.argtypes = [c_int]
    lib.paro_attn_decode_gated.restype = c_int
    lib.paro_attn_decode_gated.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, f32, f32, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_void_p]
    if all(hasattr(lib, sym) for sym in EXPERIMENTAL_EXPORTS):      # a `make EXPERIMENTAL=1` library (include/paro_abi_experimental.h)
        sigs = {"plan": [POINTER(ParoEnginePhase), c_int, c_int, POINTER(ParoEngine)],
                "build": [POINTER(ParoEnginePhase), POINTER(ParoEngine), c_void_p],
                "describe": [POINTER(ParoEnginePhase), POINTER(ParoEngine), c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)],
                "run": [POINTER(ParoEngine), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
                "trace": [POINTER(ParoEngine), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]}
        for prefix in ("paro_engine_", "paro_engine2_"):               # engine2.hip: the same signatures
            for fn, argtypes in sigs.items():
                f = getattr(lib, prefix + fn)
                f.restype, f.argtypes = c_int, argtypes
    if lib.paro_abi_version() != PARO_ABI_VERSION:
        raise RuntimeError(f"paroquant_amd: ABI version mismatch (library {lib.paro_abi_version()}, "
                           f"binding {PARO_ABI_VERSION})")
    _lib = lib
    return lib


def has_experimental() -> bool:
    """True when the loaded library was built with ``make EXPERIMENTAL=1`` (the persistent decode engines)."""
    lib = load()
    return all(hasattr(lib, sym) for sym in EXPERIMENTAL_EXPORTS)


def check(rc: int) -> None:
    """Negative return code -> RuntimeError(paro_last_error()), mirroring TORCH_CHECK (rotation.cu:66,114)."""
    if rc != 0:
        msg = load().paro_last_error()
        raise RuntimeError((msg or b"unknown error").decode("utf-8", "replace"))


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return DTYPE_F16
    if dt == torch.bfloat16:
        return DTYPE_BF16
    if dt == torch.float32:
        return DTYPE_F32
    raise RuntimeError(f"rotate supports Float, Half, and BFloat16, got {dt}")   # rotation.cu:92


def current_stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream

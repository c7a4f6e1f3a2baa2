"""vLLM quantisation plug-in surface, MI355X-native.

Mirrors ``paroquant/inference/backends/vllm/plugin.py`` (ParoQuantConfig :79-164,
ParoQuantLinearMethod :167-311, rotation weight loaders :33-76) with the same names, argument
meaning and error behaviour, but

* derives from ``LinearMethodBase`` (vLLM-on-ROCm has no Marlin, so the reference's
  ``AWQMarlinLinearMethod`` parent and ``check_marlin_supports_layer`` gate cannot be kept) and
  creates the AWQ parameters itself;
* ``process_weights_after_loading`` repacks the WHOLE merged ``qweight`` once into the CDNA4 tile
  layout (the reference converts AWQ -> Marlin per partition, :251-275);
* ``apply`` is ONE fused launch for all merged partitions (the reference issues, per partition, a
  rotate + a Marlin GEMM, then ``torch.cat`` and a bias add, :288-311).

vLLM is optional at import time: with vLLM installed the classes subclass / register with it
(entry point ``vllm.general_plugins: paroquant = paroquant_amd.vllm_plugin:register``, cf.
pyproject.toml:22-23 of the reference); without it they are plain classes with the same contract
so the host logic is testable anywhere.
"""
from __future__ import annotations

from typing import Any, Optional

import torch
from torch.nn import Parameter

from .linear import PackedParoWeights, coalesce_partitions, pad_partitions

try:  # pragma: no cover - exercised only where vLLM is installed
    from vllm.model_executor.layers.linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
    from vllm.model_executor.layers.quantization import register_quantization_config
    from vllm.model_executor.layers.quantization.base_config import QuantizationConfig
    from vllm.model_executor.layers.quantization.utils.quant_utils import is_layer_skipped
    from vllm.model_executor.parameter import GroupQuantScaleParameter, PackedvLLMParameter

    HAVE_VLLM = True
except Exception:  # ImportError or a broken install
    HAVE_VLLM = False
    LinearBase = torch.nn.Module

    class LinearMethodBase:  # minimal stand-ins so the contract can be exercised without vLLM
        pass

    class UnquantizedLinearMethod(LinearMethodBase):
        pass

    class QuantizationConfig:
        packed_modules_mapping: dict = {}

        def __init__(self):
            self.packed_modules_mapping = {}

        @staticmethod
        def get_from_keys_or(config: dict, keys: list, default):
            for k in keys:
                if k in config:
                    return config[k]
            return default

    def register_quantization_config(name):
        def deco(cls):
            return cls
        return deco

    def is_layer_skipped(prefix, ignored, fused_mapping=None, skip_with_substr=False):
        if not ignored:
            return False
        return any((m in prefix) if skip_with_substr else (m == prefix) for m in ignored)

    GroupQuantScaleParameter = PackedvLLMParameter = None

_QKV_SLOT = {"q": 0, "k": 1, "v": 2}              # slot of a fused-QKV shard id (reference plugin.py:28)
_SUPPORTED_BITS = (4,)


def _tp_rank() -> int:
    if HAVE_VLLM:  # pragma: no cover
        from vllm.distributed import get_tensor_model_parallel_rank
        return get_tensor_model_parallel_rank()
    dist = torch.distributed
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _maybe_shard_input(target: torch.Tensor, loaded_weight: torch.Tensor, tp_rank: Optional[int] = None) -> torch.Tensor:
    """The slice of a checkpoint rotation tensor that belongs in ``target``.

    Rotation parameters run along the linear's INPUT dimension.  A row-parallel layer allocates them
    ``input_size_per_partition`` wide while the checkpoint stores the full width, so rank r takes columns
    ``[r * w, (r + 1) * w)`` (the rotation is block diagonal per 128-channel group, which makes the slice
    self-contained).  Same contract as the reference loader, plugin.py:33-50."""
    have, want = loaded_weight.shape[-1], target.shape[-1]
    if have == want:
        return loaded_weight
    n_shards, rest = divmod(have, want)
    if rest:
        raise ValueError(f"ParoQuant rotation loader: incompatible shapes target={tuple(target.shape)} "
                         f"loaded={tuple(loaded_weight.shape)}")
    rank = _tp_rank() if tp_rank is None else int(tp_rank)
    if not 0 <= rank < n_shards:
        raise ValueError(f"ParoQuant rotation loader: tp rank {rank} outside the {n_shards} input shards")
    return loaded_weight.narrow(-1, rank * want, want)


def _partition_slots(shard_id) -> tuple:
    """Partition indices addressed by a vLLM shard id: ``"q"/"k"/"v"`` (fused QKV), an int (gate/up and
    other merged projections) or a tuple of those (one checkpoint tensor feeding several partitions)."""
    ids = shard_id if isinstance(shard_id, tuple) else (shard_id,)
    return tuple(_QKV_SLOT.get(i, i) if isinstance(i, str) else i for i in ids)


def _rotation_weight_loader(param: Parameter, loaded_weight: torch.Tensor,
                            loaded_shard_id: int | str | tuple | None = None) -> None:
    """``weight_loader`` of the ``theta`` / ``pairs`` / ``channel_scales`` parameters, which carry one
    leading slot per merged partition (reference dispatch: plugin.py:53-76)."""
    store = param.data
    if loaded_shard_id is None:   # un-merged projection: the only slot (or a slot-less tensor)
        slots = [store[0] if store.dim() > loaded_weight.dim() else store]
    else:
        slots = [store[i] for i in _partition_slots(loaded_shard_id)]
    for slot in slots:
        slot.copy_(_maybe_shard_input(slot, loaded_weight))


@register_quantization_config("paroquant")
class ParoQuantConfig(QuantizationConfig):
    """Same fields / classmethods as the reference config (plugin.py:79-164)."""

    def __init__(self, bits: int, group_size: int, krot: int, zero_point: bool) -> None:
        super().__init__()
        if bits not in _SUPPORTED_BITS:
            raise ValueError(f"Unsupported bits={bits}. Supported: {list(_SUPPORTED_BITS)}")
        self.bits, self.group_size, self.krot, self.zero_point = bits, group_size, krot, zero_point
        self.pack_factor = 32 // bits            # INT4 columns per int32 word of qweight / qzeros
        self.modules_to_not_convert = None       # filled from the checkpoint header (maybe_update_config)

    def __repr__(self) -> str:
        return (f"ParoQuantConfig(bits={self.bits}, group_size={self.group_size}, krot={self.krot}, "
                f"zero_point={self.zero_point})")

    @classmethod
    def get_name(cls) -> str:
        return "paroquant"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        # the reference returns 75 (a CUDA compute capability, plugin.py:99-101); vLLM's ROCm platform reports gfx9xx as (9, x), so
        # the gate that matters here is 'a CDNA GPU': 90 admits gfx90a / gfx942 / gfx950 and nothing older (the kernels are gfx950
        # only -- the library refuses to load elsewhere, _native.load)
        return 90

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return ["config.json"]

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "ParoQuantConfig":
        return cls(
            bits=cls.get_from_keys_or(config, ["bits"], 4),
            group_size=cls.get_from_keys_or(config, ["group_size"], 128),
            krot=cls.get_from_keys_or(config, ["krot"], 8),
            zero_point=cls.get_from_keys_or(config, ["zero_point"], True),
        )

    @staticmethod
    def unquantized_modules_from_metadata(metadata: dict) -> list[str]:
        """Which modules vLLM must NOT quantise, judged from the safetensors header alone (the pure part
        of the reference's ``maybe_update_config``, plugin.py:123-151): a module owning a ``.weight`` whose
        tensors are all plain floating point.  Names are reported the way vLLM prefixes layers -- without
        the leading ``model.`` and, inside the decoder stack, from ``layers.`` on."""
        float_dtypes = ("F16", "BF16", "F32")
        owner = lambda key: key.rpartition(".")[0]
        with_weight = {owner(k) for k in metadata if k.endswith(".weight")}
        non_float = {owner(k) for k, info in metadata.items() if info.get("dtype") and info["dtype"] not in float_dtypes}

        def vllm_name(module: str) -> str:
            module = module[len("model."):] if module.startswith("model.") else module
            at = module.find("layers.")
            return module if at < 0 else module[at:]

        return sorted(vllm_name(m) for m in with_weight - non_float)

    def maybe_update_config(self, model_name: str, revision: str | None = None):
        """Auto-detect unquantized layers from safetensors metadata."""
        if self.modules_to_not_convert:
            return
        from vllm.transformers_utils.config import get_safetensors_params_metadata   # pragma: no cover
        metadata = get_safetensors_params_metadata(model_name, revision=revision)     # pragma: no cover
        self.modules_to_not_convert = self.unquantized_modules_from_metadata(metadata)  # pragma: no cover

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if not isinstance(layer, LinearBase):
            return None
        if is_layer_skipped(prefix, self.modules_to_not_convert, self.packed_modules_mapping, skip_with_substr=True):
            return UnquantizedLinearMethod()
        if self.group_size not in (64, 128):
            raise ValueError(f"Unsupported group_size={self.group_size}: the MI355X kernels take 64 or 128")
        return ParoQuantLinearMethod(self)


def _plain_param(data: torch.Tensor, **attrs) -> Parameter:
    p = Parameter(data, requires_grad=False)
    for k, v in attrs.items():
        setattr(p, k, v)
    return p


class ParoQuantLinearMethod(LinearMethodBase):
    """Per-projection rotation fused with the INT4 matmul (one launch for all merged partitions)."""

    def __init__(self, quant_config: ParoQuantConfig) -> None:
        self.quant_config = quant_config

    def create_weights(
        self,
        layer: torch.nn.Module,
        input_size_per_partition: int,
        output_partition_sizes: list[int],
        input_size: int,
        output_size: int,
        params_dtype: torch.dtype,
        **extra_weight_attrs,
    ) -> None:
        cfg = self.quant_config
        gs = cfg.group_size if cfg.group_size != -1 else input_size
        if input_size_per_partition % gs != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        out = sum(output_partition_sizes)
        if out % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        weight_loader = extra_weight_attrs.get("weight_loader")
        n_groups = input_size_per_partition // gs
        qweight = torch.zeros(input_size_per_partition, out // cfg.pack_factor, dtype=torch.int32)
        qzeros = torch.zeros(n_groups, out // cfg.pack_factor, dtype=torch.int32)
        # the group scales take params_dtype, as vLLM's AWQ create_weights (which the reference inherits, plugin.py:183-192) allocates
        # them: a bf16 model loads the checkpoint's fp16 scales through a bf16 parameter; the repack below converts back to fp16
        scales = torch.zeros(n_groups, out, dtype=params_dtype if params_dtype in (torch.float16, torch.bfloat16) else torch.float16)
        if HAVE_VLLM:  # pragma: no cover
            layer.register_parameter("qweight", PackedvLLMParameter(
                data=qweight, input_dim=0, output_dim=1, packed_dim=1, packed_factor=cfg.pack_factor,
                weight_loader=weight_loader))
            layer.register_parameter("qzeros", PackedvLLMParameter(
                data=qzeros, input_dim=0, output_dim=1, packed_dim=1, packed_factor=cfg.pack_factor,
                weight_loader=weight_loader))
            layer.register_parameter("scales", GroupQuantScaleParameter(
                data=scales, input_dim=0, output_dim=1, weight_loader=weight_loader))
        else:
            layer.register_parameter("qweight", _plain_param(qweight, input_dim=0, output_dim=1, packed_dim=1,
                                                            packed_factor=cfg.pack_factor))
            layer.register_parameter("qzeros", _plain_param(qzeros, input_dim=0, output_dim=1, packed_dim=1,
                                                           packed_factor=cfg.pack_factor))
            layer.register_parameter("scales", _plain_param(scales, input_dim=0, output_dim=1))

        n_parts = len(output_partition_sizes)
        krot = cfg.krot
        for name, shape, dtype in [                                   # plugin.py:195-203
            ("theta", (n_parts, krot, input_size_per_partition // 2), torch.float16),
            ("pairs", (n_parts, krot, input_size_per_partition), torch.int16),
            ("channel_scales", (n_parts, 1, input_size_per_partition), torch.float16),
        ]:
            init_fn = torch.ones if name == "channel_scales" else torch.zeros
            p = Parameter(init_fn(shape, dtype=dtype), requires_grad=False)
            p.weight_loader = _rotation_weight_loader
            layer.register_parameter(name, p)

        layer.num_partitions = n_parts
        layer.output_partition_sizes = list(output_partition_sizes)
        layer.input_size_per_partition = input_size_per_partition

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        """One-time repack into the CDNA4 tile layout; the checkpoint-format params are released
        (as the reference releases them after the Marlin repack, plugin.py:267,276-279)."""
        sizes = list(layer.output_partition_sizes)
        pack = self.quant_config.pack_factor
        qw, qz, sc = layer.qweight.data, layer.qzeros.data, layer.scales.data.to(torch.float16)
        qw, qz, sc, padded = pad_partitions(qw, qz, sc, sizes, pack)   # reference pads to the Marlin tile (plugin.py:210-217)
        # slots that received the same checkpoint rotation (tuple shard ids, plugin.py:60-76) become ONE kernel partition
        theta, pairs, cs, kernel_sizes, _ = coalesce_partitions(layer.theta.data, layer.pairs.data, layer.channel_scales.data, padded)
        layer.paro_packed = PackedParoWeights(qw.contiguous(), qz.contiguous(), sc.contiguous(), theta, pairs, cs, kernel_sizes, None,
                                              self.quant_config.group_size, self.quant_config.bits)
        from . import autotune
        if autotune.enabled():                  # PARO_AUTOTUNE=1: measured launch shapes, once per distinct layer shape (autotune.py)
            layer.paro_packed.autotune(torch.bfloat16 if layer.scales.dtype == torch.bfloat16 else torch.float16)
        layer.kernel_partition_sizes = kernel_sizes
        layer.padded_partition_sizes = padded
        layer.rot_theta = layer.paro_packed.theta
        layer.rot_pairs = layer.paro_packed.pairs
        layer.rot_scales = layer.paro_packed.channel_scales.reshape(len(kernel_sizes), 1, -1)
        del layer.qweight, layer.qzeros, layer.scales
        del layer.theta, layer.pairs, layer.channel_scales

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
        sizes = layer.output_partition_sizes
        padded = layer.padded_partition_sizes
        if padded == sizes:
            return layer.paro_packed.apply(x, bias)
        y = layer.paro_packed.apply(x, None)
        outs, col = [], 0
        for s, p in zip(sizes, padded):
            outs.append(y[..., col:col + s])
            col += p
        y = torch.cat(outs, dim=-1)
        return y + bias if bias is not None else y


def register() -> None:
    """``vllm.general_plugins`` entry point (reference: backends/vllm/__init__.py:6-9)."""
    from . import ops  # noqa: F401  -- registers torch.ops.rotation.rotate / torch.ops.paro.*

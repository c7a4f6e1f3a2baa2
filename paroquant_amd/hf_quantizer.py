"""Transformers integration: the ``"paroquant"`` quantisation method on MI355X.

What the reference's HF backend does (``paroquant/inference/backends/transformers/quantizer.py``) and
what this module therefore has to provide under the same registry name:

* a config class registered as ``"paroquant"`` carrying ``bits / group_size / krot /
  modules_to_not_convert`` (reference :47-65);
* a quantizer registered as ``"paroquant"`` that, before the weights are read, swaps every ``nn.Linear``
  whose checkpoint entry owns a ``.qweight`` tensor for a :class:`RotateQuantizedLinear` (reference
  :30-44, :95-115) -- vision towers, ``lm_head`` and norms have no such entry and stay as they are;
* fp16 activations only and a hard failure without a GPU (reference :78-86).

Specific to this implementation: after loading, every swapped layer is repacked once into the CDNA4
tile layout (``RotateQuantizedLinear.prepare``), so the first forward -- and any HIP-graph capture --
allocates nothing; and ``validate_environment`` also fails loudly when ``libparo_mi355x.so`` is absent.
Importing the module registers ``torch.ops.rotation.rotate`` through ``ops``.
"""
from __future__ import annotations

import json
import logging
import re
from pathlib import Path
from typing import Iterable, Iterator, Optional, Sequence

import torch
import torch.nn as nn
from transformers.quantizers.auto import register_quantization_config, register_quantizer
from transformers.quantizers.base import HfQuantizer
from transformers.utils.quantization_config import QuantizationConfigMixin

from . import _native, ops  # noqa: F401  (ops: registers rotation::rotate and the paro:: operators)
from .linear import RotateQuantizedLinear

log = logging.getLogger(__name__)

_QWEIGHT_SUFFIX = ".qweight"
_SHARD_INDEX = "model.safetensors.index.json"
# mixture-of-experts export of the reference (cli/convert.py:381-405): per expert `{base}.{e}.{gate,up,down}_proj.{qweight,qzeros,scales}`
# next to ONE rotation per projection, `{base}.gate_up_weight_{theta,pairs,channel_scales}` / `{base}.down_weight_*`; `base` is the
# path of the model's fused experts module (`...mlp.experts`).  The reference consumes it in its MLX back-end only
# (mlx/load.py:121-200 `_stack_moe_expert_weights` / `_remap_shared_moe_rotation`, mlx/modules.py:159-212 RotateSwitchGLU).
_EXPERT_QWEIGHT_RE = re.compile(r"^(.+)\.(\d+)\.(gate_proj|up_proj|down_proj)\.qweight$")
_SHARED_ROT_RE = re.compile(r"^(.+)\.(gate_up_weight|down_weight)_(theta|pairs|channel_scales)$")


# ----------------------------------------------------------------------------- checkpoint inspection
def _checkpoint_dir(model_path: str) -> Path:
    """A local directory holding the checkpoint: the path itself, or the hub snapshot of a repo id."""
    p = Path(model_path)
    if p.is_dir():
        return p
    from huggingface_hub import snapshot_download

    return Path(snapshot_download(model_path))


def _tensor_names(ckpt: Path) -> Iterator[str]:
    """Names of all tensors of a safetensors checkpoint, from the shard index when there is one
    (no shard has to be opened), else from the shard headers."""
    index = ckpt / _SHARD_INDEX
    if index.is_file():
        yield from json.loads(index.read_text()).get("weight_map", {})
        return
    from safetensors import safe_open

    for shard in sorted(ckpt.glob("*.safetensors")):
        with safe_open(str(shard), framework="pt") as handle:
            yield from handle.keys()


def _find_quantized_modules(model_path: str) -> set[str]:
    """Module paths that are ParoQuant-quantised in the checkpoint: exactly those with a ``.qweight``."""
    names = _tensor_names(_checkpoint_dir(model_path))
    return {n[: -len(_QWEIGHT_SUFFIX)] for n in names if n.endswith(_QWEIGHT_SUFFIX)}


def _find_moe_expert_blocks(names: Iterable[str]) -> dict:
    """``{experts module path: number of experts}`` for every fused-experts block the checkpoint stores in the reference's
    MoE export format: per-expert ``{base}.{e}.{proj}.qweight`` AND the shared rotation ``{base}.gate_up_weight_theta``."""
    names = list(names)
    shared = {m.group(1) for m in map(_SHARED_ROT_RE.match, names) if m}
    blocks: dict = {}
    for m in map(_EXPERT_QWEIGHT_RE.match, names):
        if m and m.group(1) in shared:
            blocks[m.group(1)] = max(blocks.get(m.group(1), 0), int(m.group(2)) + 1)
    return blocks


# ----------------------------------------------------------------------------- config
@register_quantization_config("paroquant")
class ParoQuantConfig(QuantizationConfigMixin):
    """``quantization_config`` block of a ``*-PARO`` checkpoint (W4A16, group 128, ``krot`` Givens stages)."""

    def __init__(self, bits: int = 4, group_size: int = 128, krot: int = 8,
                 modules_to_not_convert: Optional[Sequence[str]] = None, free_checkpoint_buffers: bool = False, **_ignored):
        self.quant_method = "paroquant"
        self.bits, self.group_size, self.krot = int(bits), int(group_size), int(krot)
        # this build only: drop the AWQ-format buffers after the one-time repack (halves the INT4 footprint;
        # the model can then not be saved or moved) -- RotateQuantizedLinear.release_checkpoint_buffers
        self.free_checkpoint_buffers = bool(free_checkpoint_buffers)
        self.modules_to_not_convert = None if modules_to_not_convert is None else list(modules_to_not_convert)
        post_init = getattr(self, "post_init", None)   # newer transformers validate here
        if callable(post_init):
            post_init()


# ----------------------------------------------------------------------------- module surgery
def replace_linears(model: nn.Module, quantized_modules: Iterable[str], qcfg) -> int:
    """Put a :class:`RotateQuantizedLinear` of the same geometry wherever ``quantized_modules`` names an
    ``nn.Linear`` of ``model``; anything else under those names is left alone.  Returns the number of
    layers replaced.  Driven by the name list (one ``get_submodule`` per name), not by a walk over the
    whole module tree."""
    replaced = 0
    for path in sorted(set(quantized_modules)):
        owner_path, _, leaf = path.rpartition(".")
        try:
            owner = model.get_submodule(owner_path) if owner_path else model
            old = getattr(owner, leaf)
        except AttributeError:
            continue   # e.g. a fused-expert tensor that is not a module of this architecture
        if type(old) is not nn.Linear and not isinstance(old, nn.Linear):
            continue
        new = RotateQuantizedLinear(old.in_features, old.out_features, bias=old.bias is not None,
                                    group_size=qcfg.group_size, bits=qcfg.bits, krot=qcfg.krot)
        setattr(owner, leaf, new)
        replaced += 1
    return replaced


class _AwqBuffers(nn.Module):
    """qweight / qzeros / scales of one expert projection under the checkpoint's names (cli/convert.py:149-203)."""

    def __init__(self, K: int, N: int, group_size: int):
        super().__init__()
        self.register_buffer("qweight", torch.zeros(K, N // 8, dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros(K // group_size, N // 8, dtype=torch.int32))
        self.register_buffer("scales", torch.zeros(K // group_size, N, dtype=torch.float16))


class _ExpertBuffers(nn.Module):
    def __init__(self, H: int, I: int, group_size: int):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _AwqBuffers(H, I, group_size), _AwqBuffers(H, I, group_size), _AwqBuffers(I, H, group_size)


class ParoHfExperts(nn.Module):
    """Stands where a model's fused experts module stood (transformers' ``Qwen3MoeExperts`` and its relatives: ``gate_up_proj
    [E, 2 I, H]``, ``down_proj [E, H, I]``, called as ``experts(hidden_states, top_k_index, top_k_weights)``) and owns the checkpoint
    tensors of the reference's MoE export under their on-disk names -- children ``"0" .. "E-1"`` with ``gate_proj / up_proj /
    down_proj . qweight / qzeros / scales`` and the shared rotations ``gate_up_weight_* / down_weight_*`` (cli/convert.py:381-405) --
    so ``from_pretrained`` loads them without a key remap.  After loading, :meth:`prepare` hands them to
    :class:`paroquant_amd.moe.ParoMoEExperts` (rotate once per projection, the routed experts of all tokens as two launches at
    decode sizes / one grouped W4A16 GEMM per projection at prefill sizes), which is what the reference's ``RotateSwitchGLU`` does on
    MLX (mlx/modules.py:159-212); the router's weights are applied here, as the surrounding block expects."""

    def __init__(self, num_experts: int, hidden: int, inter: int, group_size: int = 128, krot: int = 8):
        super().__init__()
        self.num_experts, self.hidden_dim, self.intermediate_dim = int(num_experts), int(hidden), int(inter)
        for e in range(self.num_experts):
            self.add_module(str(e), _ExpertBuffers(hidden, inter, group_size))
        for name, K in (("gate_up_weight", hidden), ("down_weight", inter)):
            self.register_buffer(f"{name}_theta", torch.zeros(krot, K // 2, dtype=torch.float16))
            self.register_buffer(f"{name}_pairs", torch.zeros(krot, K, dtype=torch.int16))
            self.register_buffer(f"{name}_channel_scales", torch.ones(1, K, dtype=torch.float16))
        self._packed = None

    def prepare(self, release: Optional[bool] = None) -> "ParoHfExperts":
        """``release``: drop the AWQ-format per-expert buffers once the kernel-layout copy exists; None = what the quantization config's
        ``free_checkpoint_buffers`` said when the module was built (the lazy path of ``forward`` honours it too)."""
        if release is None:
            release = bool(getattr(self, "free_checkpoint_buffers", False))
        from .moe import ParoMoEExperts
        dev = self.gate_up_weight_theta.device
        if dev.type != "cuda":
            raise RuntimeError("ParoQuant requires a GPU: the expert kernels have no CPU path")
        tensors = {k: v for k, v in self.state_dict().items()}
        self._packed = ParoMoEExperts(tensors, self.num_experts, dev)
        if release:      # the kernel-layout copy is complete: drop the AWQ-format per-expert buffers (they double the footprint)
            for e in range(self.num_experts):
                for proj in ("gate_proj", "up_proj", "down_proj"):
                    m = getattr(getattr(self, str(e)), proj)
                    for name in ("qweight", "qzeros", "scales"):
                        setattr(m, name, torch.empty(0, dtype=getattr(m, name).dtype, device=dev))
        return self

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, top_k_index: torch.Tensor, top_k_weights: torch.Tensor) -> torch.Tensor:
        if self._packed is None:
            self.prepare()
        x = hidden_states.reshape(-1, self.hidden_dim)
        y = self._packed(x, top_k_index.reshape(x.size(0), -1))                       # [T, k, H]: per-(token, expert) outputs
        out = (y.float() * top_k_weights.reshape(x.size(0), -1, 1).float()).sum(1)    # the block's weighted sum (fp32, one rounding)
        return out.to(hidden_states.dtype).reshape(hidden_states.shape)


def replace_experts(model: nn.Module, blocks: dict, qcfg) -> int:
    """Put a :class:`ParoHfExperts` wherever ``blocks`` ({module path: experts}) names a fused experts module of ``model``."""
    replaced = 0
    for path, n_experts in sorted(blocks.items()):
        owner_path, _, leaf = path.rpartition(".")
        try:
            owner = model.get_submodule(owner_path) if owner_path else model
            old = getattr(owner, leaf)
        except AttributeError:
            continue
        H = getattr(old, "hidden_dim", None) or getattr(old, "hidden_size", None)
        I = getattr(old, "intermediate_dim", None) or getattr(old, "intermediate_size", None)
        gu = getattr(old, "gate_up_proj", None)
        if (H is None or I is None) and gu is not None and gu.dim() == 3:        # [E, 2 I, H]
            I, H = gu.shape[1] // 2, gu.shape[2]
        if H is None or I is None:
            log.warning("ParoQuant: cannot read the geometry of %s (%s); its experts stay unconverted.", path, type(old).__name__)
            continue
        mod = ParoHfExperts(n_experts, int(H), int(I), qcfg.group_size, qcfg.krot)
        mod.free_checkpoint_buffers = bool(getattr(qcfg, "free_checkpoint_buffers", False))   # (the lazy prepare() of forward honours it)
        setattr(owner, leaf, mod)
        replaced += 1
    return replaced


# ----------------------------------------------------------------------------- quantizer
@register_quantizer("paroquant")
class ParoQuantHfQuantizer(HfQuantizer):
    """Loads pre-quantised ``*-PARO`` checkpoints onto the fused MI355X kernels (inference only)."""

    requires_calibration = True   # pre-quantised checkpoints only: there is nothing to calibrate here

    # -- environment / dtype policy
    def validate_environment(self, *args, **kwargs):
        if not torch.cuda.is_available():
            raise RuntimeError("ParoQuant needs a GPU (ROCm/HIP device): the rotation + INT4 kernels have no CPU path.")
        _native.load()   # raises with the build hint if the HIP library has not been built

    def update_dtype(self, dtype):
        wanted = torch.float16
        if dtype is not wanted:
            log.warning("ParoQuant runs fp16 activations; replacing dtype=%s by %s.", dtype, wanted)
        return wanted

    update_torch_dtype = update_dtype   # name used by older transformers releases

    # -- hooks around weight loading
    def _process_model_before_weight_loading(self, model, **kwargs):
        cfg = self.quantization_config
        names = list(_tensor_names(_checkpoint_dir(model.config._name_or_path)))
        targets = {n[: -len(_QWEIGHT_SUFFIX)] for n in names if n.endswith(_QWEIGHT_SUFFIX)}
        targets.difference_update(cfg.modules_to_not_convert or ())
        # mixture-of-experts blocks first: their per-expert `.qweight` entries are not nn.Linear modules of the model
        blocks = {b: e for b, e in _find_moe_expert_blocks(names).items() if b not in (cfg.modules_to_not_convert or ())}
        n_moe = replace_experts(model, blocks, cfg)
        targets = {t for t in targets if not any(t.startswith(b + ".") for b in blocks)}
        n = replace_linears(model, targets, cfg)
        log.info("ParoQuant: %d of %d quantised checkpoint modules mapped onto RotateQuantizedLinear, %d expert blocks onto ParoHfExperts.",
                 n, len(targets), n_moe)

    def _process_model_after_weight_loading(self, model, **kwargs):
        for layer in model.modules():
            if isinstance(layer, RotateQuantizedLinear) and layer.qweight.is_cuda:
                layer.prepare()
                if getattr(self.quantization_config, "free_checkpoint_buffers", False):
                    layer.release_checkpoint_buffers()
            elif isinstance(layer, ParoHfExperts) and layer.gate_up_weight_theta.is_cuda:
                layer.prepare(release=getattr(self.quantization_config, "free_checkpoint_buffers", False))
        return model

    # -- capabilities
    @property
    def is_trainable(self) -> bool:
        return False

    def is_serializable(self, *args, **kwargs) -> bool:
        return True

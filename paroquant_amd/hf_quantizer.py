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
        targets = _find_quantized_modules(model.config._name_or_path)
        targets.difference_update(cfg.modules_to_not_convert or ())
        n = replace_linears(model, targets, cfg)
        log.info("ParoQuant: %d of %d quantised checkpoint modules mapped onto RotateQuantizedLinear.", n, len(targets))

    def _process_model_after_weight_loading(self, model, **kwargs):
        for layer in model.modules():
            if isinstance(layer, RotateQuantizedLinear) and layer.qweight.is_cuda:
                layer.prepare()
                if getattr(self.quantization_config, "free_checkpoint_buffers", False):
                    layer.release_checkpoint_buffers()
        return model

    # -- capabilities
    @property
    def is_trainable(self) -> bool:
        return False

    def is_serializable(self, *args, **kwargs) -> bool:
        return True

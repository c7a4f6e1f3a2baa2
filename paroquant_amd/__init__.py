"""paroquant_amd -- MI355X-native (gfx950 / CDNA4) implementation of ParoQuant's inference hot
path: fused pairwise (Givens) rotation + INT4 dequantisation + GEMV/GEMM behind the reference's
operator API (``torch.ops.rotation.rotate``, ``RotateQuantizedLinear``, the vLLM / Transformers
``paroquant`` plug-ins).  See DESIGN.md and INTEGRATION.md.

Importing the package registers the torch operators; the native library itself is loaded lazily on
first use and there is no CPU fallback (``_native.load`` raises if ``libparo_mi355x.so`` is missing).
"""
from . import ops  # noqa: F401  -- registers torch.ops.rotation.rotate, torch.ops.paro.*
from .linear import PackedParoWeights, RotateQuantizedLinear

__all__ = ["RotateQuantizedLinear", "PackedParoWeights", "ops"]
__version__ = "0.1.0"

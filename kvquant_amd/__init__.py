"""kvquant_amd -- MI355X-native (gfx950) implementation of the KVQuant
deployment hot path: NUQ quantize-and-pack on append, LUT-dequant attention
matvecs over the packed 2/3/4-bit KV cache (K with fused RoPE) and the fused
sparse-outlier matvec, behind the reference's own operator surface.

  kvquant_amd.quant_cuda   the reference's 34-function extension module surface
  kvquant_amd.QuantK/QuantV the cache-owning operator classes
  include/kvq.h            the C ABI underneath (libkvq.so)

The product path is HIP only; importing works without a GPU, calling does not.
"""
import sys

__version__ = "0.1.0"


def install_as_quant_cuda():
    """Make ``import quant_cuda`` (as the reference's patched modeling_llama.py
    does at import time) resolve to this package's implementation."""
    from . import quant_cuda
    sys.modules["quant_cuda"] = quant_cuda
    return quant_cuda


def __getattr__(name):
    if name in ("QuantK", "QuantV"):
        from . import cache
        return getattr(cache, name)
    if name == "quant_cuda":
        import importlib
        return importlib.import_module(".quant_cuda", __name__)
    raise AttributeError(name)

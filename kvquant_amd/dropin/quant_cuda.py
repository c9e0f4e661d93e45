"""Put this directory on PYTHONPATH and the reference's `import quant_cuda`
(modeling_llama.py:53) picks up the MI355X implementation unchanged."""
from kvquant_amd.quant_cuda import *  # noqa: F401,F403
from kvquant_amd.quant_cuda import _NS as _NS

globals().update(_NS)

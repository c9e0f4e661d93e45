#!/usr/bin/env python
"""BASELINE config 4: the causal attention of an S-token prompt -- kvq_prefill_attention (hand-written MFMA flash
kernel) next to torch SDPA (ROCm's flash / CK backend; the reference calls flash-attn here, ML:1861-1874).
usage: python tools/prefill_attn_bench.py [S ...]"""
import json
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
if os.environ.get("KVQ_LIB"):
    import kvquant_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, HD = 32, 128


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for S in [int(a) for a in sys.argv[1:]] or [8192]:
    x = [torch.randn(S, H, HD, device=dev, dtype=torch.float16) for _ in range(3)]
    q, k, v = (t.transpose(0, 1) for t in x)                       # [H, S, D] views of token-major activations
    flops = 4.0 * H * S * S * HD / 2          # causal: half of the S x S tiles
    ms_sdpa = timeit(lambda: F.scaled_dot_product_attention(q.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), is_causal=True))
    ms_kvq = timeit(lambda: ops.prefill_attention(q, k, v))
    print(json.dumps({"op": "causal attention fp16, 32 heads x 128", "S": S,
                      "sdpa_ms": round(ms_sdpa, 3), "sdpa_TFLOPs": round(flops / ms_sdpa / 1e9, 1),
                      "kvq_mfma_ms": round(ms_kvq, 3), "kvq_mfma_TFLOPs": round(flops / ms_kvq / 1e9, 1),
                      "kvq_frac_of_2500_TF_dense_fp16_peak": round(flops / ms_kvq / 1e9 / 2500, 3)}), flush=True)

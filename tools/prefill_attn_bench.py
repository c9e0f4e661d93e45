#!/usr/bin/env python
"""BASELINE config 4 context: the causal attention of an 8K-token prompt (the reference calls flash-attn there,
ML:1861-1874; here torch SDPA = the ROCm flash / CK backend, on MFMA) next to the prefill packs it shares the prompt
with.  Decides whether a hand-written MFMA attention kernel is worth it.  usage: python tools/prefill_attn_bench.py [S]"""
import json
import sys
import torch
import torch.nn.functional as F

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
H, HD = 32, 128
q, k, v = (torch.randn(1, H, S, HD, device=dev, dtype=torch.float16) for _ in range(3))
for _ in range(3):
    F.scaled_dot_product_attention(q, k, v, is_causal=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    F.scaled_dot_product_attention(q, k, v, is_causal=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
flops = 4.0 * H * S * S * HD / 2          # causal: half of the S x S tiles
print(json.dumps({"op": "causal SDPA fp16, 32 heads x 128", "S": S, "ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 1),
                  "frac_of_2500_TF_dense_fp16_peak": round(flops / ms / 1e9 / 2500, 3)}))

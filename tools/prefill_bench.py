#!/usr/bin/env python
"""Prefill pack timing: S prompt tokens, nuq4 + 1 % outliers, one layer.
fused = kvq_pack_{k,v}_fused (one launch); glue = pack kernel + torch.topk / gather / sort on the GPU
(the reference's structure).  usage: python tools/prefill_bench.py [S] [bits]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from kvquant_amd.cache import QuantK, QuantV  # noqa: E402
from tests import decode_check  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
H, HD, C = 32, 128, 4096
quant, scale, shift = decode_check.quantizer(bits, seed=1)
g = torch.Generator().manual_seed(1234)
k = (torch.randn(C, S, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, S).to(dev)
v = (torch.randn(C, S, generator=g) * 1.7).reshape(H, HD, S).to(dev)
kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=S, include_sparse=True,
          sparsity_threshold=0.99, first_few_fp16=0, device=dev)
kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
for c in (kc, vc):
    c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def kf():
    kc.reset(); kc.parallel_pack(k)


def kg():
    kc.reset(); kc.parallel_pack(k, fused=False)


def vf():
    vc.reset(); vc.parallel_pack(v)


def vg():
    vc.reset(); vc.parallel_pack(v, *vc.topk_inputs(v.reshape(-1, S).t().contiguous()))


def rs():
    kc.reset()


t_reset = timed(rs)
print("S=%d bits=%d  (cache reset alone %.2f ms)" % (S, bits, t_reset))
for name, fn in (("K fused", kf), ("K glue", kg), ("V fused", vf), ("V glue", vg)):
    ms = timed(fn) - t_reset
    gb = (C * S * 4 + C * S * bits / 8) / 1e9
    print("%-8s %8.2f ms   %6.1f GB/s (input + packed bytes)" % (name, ms, gb / ms * 1e3))

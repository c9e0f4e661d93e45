#!/usr/bin/env python
"""Kernel-level timing of the two decode matvecs (development tool; the judged
numbers come from bench.py).  usage: python tools/kbench.py [bits] [L ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import os  # noqa: E402
if os.environ.get("KVQ_LIB"):      # ablation builds (tools/abl): timing only
    import kvquant_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import quant_cuda as qc  # noqa: E402

H, HD, C = 32, 128, 4096


def run(bits, L, sparse=True, iters=20, nrot=3):
    n = 2 ** bits
    W = HD // 32 * bits
    max_len = L + int(__import__("os").environ.get("KB_PAD", "8"))
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    mats = [torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
            for _ in range(nrot)]
    lut = torch.randn(H, HD, n, device=dev).sort(dim=-1).values.contiguous()
    rows = torch.randn(max_len, n, device=dev).sort(dim=-1).values.contiguous()
    q = torch.randn(1, H, HD, device=dev)
    p = torch.softmax(torch.randn(1, H, L, device=dev), dim=-1).contiguous()
    vals = torch.randn(max_len, 42, device=dev)
    idx = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, dtype=torch.int32), dim=-1).values.contiguous()
    mul_k = torch.zeros(1, H, L, device=dev)
    mul_v = torch.zeros(1, H, HD, device=dev)
    kname = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt%s" % (bits, "2" if sparse else "")
    vname = "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt%s" % (bits, "2" if sparse else "")

    def kcall(i):
        if sparse:
            getattr(qc, kname)(q, mats[i % nrot], mul_k, lut, L, vals, idx, 10000.0, 0)
        else:
            getattr(qc, kname)(q, mats[i % nrot], mul_k, lut, L, 10000.0, 0)

    def vcall(i):
        if sparse:
            getattr(qc, vname)(p, mats[i % nrot], mul_v, rows, L, vals, idx)
        else:
            getattr(qc, vname)(p, mats[i % nrot], mul_v, rows, L)

    out = {}
    for nm, fn, bpt in (("K", kcall, C * bits // 8 + (336 if sparse else 0) + 128),
                        ("V", vcall, C * bits // 8 + (336 if sparse else 0) + 4 * n + 128)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        gbs = L * bpt / us / 1e3
        out[nm] = (us, gbs)
        print("bits=%d L=%7d sparse=%d %s: %9.1f us  %8.1f GB/s (%.1f%% of 8 TB/s)" % (bits, L, sparse, nm, us, gbs, gbs / 80), flush=True)
    return out


if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    Ls = [int(a) for a in sys.argv[2:]] or [4096, 32768, 131072]
    for L in Ls:
        run(bits, L, True)
        run(bits, L, False)

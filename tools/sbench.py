#!/usr/bin/env python
"""time the small decode kernels (softmax, fused appends) -- development tool"""
import math, sys, torch
sys.path.insert(0, ".")
from kvquant_amd import ops
H, HD, C = 32, 128, 4096
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for L in (131072, 131075):
    raw = torch.randn(H, L, device=dev) * 20
    print("softmax L=%d: %.1f us" % (L, timeit(lambda: ops.softmax_scale(raw, 1 / math.sqrt(HD)))))
bits, n, max_len = 4, 16, 256
mat = torch.zeros(H, 16, max_len, dtype=torch.int32, device=dev)
lut = torch.randn(H, HD, n, device=dev).sort(-1).values.contiguous()
lo = torch.full((C,), -2.5, device=dev); hi = torch.full((C,), 2.5, device=dev)
x = torch.randn(C, device=dev)
outl = torch.zeros(max_len, 42, device=dev); oidx = torch.zeros(max_len, 42, dtype=torch.int32, device=dev)
rows = torch.zeros(max_len, n, device=dev); lsort = torch.linspace(-1, 1, n, device=dev)
print("append_k_fused: %.1f us" % timeit(lambda: ops.append_k_fused(bits, mat, lut, lut, x, lo, hi, outl, oidx, 21, 3)))
print("append_v_fused: %.1f us" % timeit(lambda: ops.append_v_fused(bits, mat, rows, lsort, x, outl, oidx, 21, 3)))

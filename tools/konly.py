#!/usr/bin/env python
"""run one op repeatedly (for PMC collection): python tools/konly.py K|V dense|sparse [L]"""
import sys
import torch
sys.path.insert(0, ".")
from kvquant_amd import ops
which, mode = sys.argv[1], sys.argv[2]
L = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
H, HD, C, bits, n = 32, 128, 4096, 4, 16
dev = torch.device("cuda")
max_len = L + 8
mat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, 16, max_len), device=dev, dtype=torch.int64).to(torch.int32)
lut = torch.randn(H, HD, n, device=dev).sort(dim=-1).values.contiguous()
rows = torch.randn(max_len, n, device=dev).sort(dim=-1).values.contiguous()
q = torch.randn(1, H, HD, device=dev)
p = torch.softmax(torch.randn(1, H, L, device=dev), dim=-1).contiguous()
vals = torch.randn(max_len, 42, device=dev)
idx = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, dtype=torch.int32), dim=-1).values.contiguous()
mk = torch.empty(1, H, L, device=dev)
mv = torch.empty(1, H, HD, device=dev)
sp = mode == "sparse"
for i in range(6):
    if which == "K":
        ops.score_k(bits, q, mat, mk, lut, L, 10000.0, 0, vals if sp else None, idx if sp else None, accumulate=False)
    else:
        ops.mix_v(bits, p, mat, mv, rows, L, vals if sp else None, idx if sp else None, accumulate=False)
torch.cuda.synchronize()

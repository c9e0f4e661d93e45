import sys, torch
sys.path.insert(0, ".")
from tests import util
from tests.test_ops_gpu import _random_cache, H, HD, C
from kvquant_amd import quant_cuda as qc
from oracle import quant_cuda_ref as orc
bits, L = 4, 70000
n = 16
max_len = (L + 64 + 63) // 64 * 64
mat = _random_cache(bits, L, max_len, 5 + L)
g = torch.Generator().manual_seed(L + 7)
rows = torch.zeros(max_len, n)
rows[:L] = util.centroids(bits).unsqueeze(0) * (torch.rand(L, 1, generator=g) + 0.5) + torch.randn(L, 1, generator=g) * 0.1
p = torch.softmax(torch.randn(1, H, L, generator=g) * 2, dim=-1).half().float().contiguous()
vals = torch.zeros(max_len, 42); idx = torch.zeros(max_len, 42, dtype=torch.int32)
idx[:L] = (torch.arange(42) * 97 + 5).int()
vals[:L] = torch.randn(L, 42, generator=g) * 3
name = "vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2"
ref = torch.zeros(1, H, HD); out = torch.zeros(1, H, HD).cuda()
getattr(orc, name)(p, mat, ref, rows, L, vals, idx)
getattr(qc, name)(p.cuda(), mat.cuda(), out, rows.cuda(), L, vals.cuda(), idx.cuda())
d = (out.cpu() - ref).reshape(-1)
print("max abs diff", float(d.abs().max()), "at channel", int(d.abs().argmax()), "ref there", float(ref.reshape(-1)[d.abs().argmax()]), "row max", float(ref.abs().max()))
bad = (d.abs() > 1e-4).nonzero().flatten().tolist()
print("channels off by > 1e-4:", bad[:50], len(bad))
print("outlier channels:", (torch.arange(42) * 97 + 5).tolist())

import math, sys, torch
sys.path.insert(0, '.')
from kvquant_amd import ops
H, HD, C = 32, 128, 4096
bits, L, n_sink = 4, 20000, 3
gpu = torch.device('cuda:0')
n = 2 ** bits
max_len = (L + 64) // 64 * 64
g = torch.Generator().manual_seed(bits * 1000 + L)
mat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, HD // 32 * bits, max_len), dtype=torch.int64, generator=g).to(torch.int32)
lut = torch.randn(H, HD, n, generator=g).sort(dim=-1).values.contiguous()
q = torch.randn(1, H, HD, generator=g)
vals = torch.randn(max_len, 42, generator=g)
idx = torch.sort(torch.randint(0, C, (max_len, 42), generator=g, dtype=torch.int64), dim=-1).values.to(torch.int32)
inv = 1.0 / math.sqrt(HD)
sink = (torch.randn(H, n_sink, generator=g) * 3).half().to(gpu)
mg, lg, vg, ig = mat.to(gpu), lut.to(gpu), vals.to(gpu), idx.to(gpu)
s2 = torch.zeros(1, H, L, device=gpu)
ops.score_k(bits, q.to(gpu), mg, s2, lg, L, 10000.0, 0, vg, ig, accumulate=False)
p2, sp2 = ops.softmax_scale(s2[0], inv, sink)
ws = ops._workspace(mg.device, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
s1 = torch.zeros(1, H, L, device=gpu)
p1, sp1 = ops.score_k_softmax(bits, mg, s1, lg, L, 10000.0, 0, ws, vg, ig, inv, sink)
d = (p1 - p2).abs()
bad = d > p2.abs() * 2e-3 + 1e-7
print("bad count", int(bad.sum()), "of", bad.numel())
hh, tt = torch.nonzero(bad, as_tuple=True)
for h, t in list(zip(hh.tolist(), tt.tolist()))[:10]:
    print(h, t, float(p1[h, t]), float(p2[h, t]), float(s1[0, h, t]))
# exact reference in float64
x = (s1[0].half().float() * inv).half().double()
xs = sink.double()
allx = torch.cat((xs, x), dim=1)
pr = torch.softmax(allx, dim=1)[:, n_sink:]
for name, p in (("fused", p1), ("twopass", p2)):
    rel = ((p.double() - pr).abs() / pr.clamp_min(1e-30))
    big = pr > 1e-6
    print(name, "max rel err vs f64 (p>1e-6):", float(rel[big].max()), "heads with bad:", sorted(set(hh.tolist())))
for h in sorted(set(hh.tolist()))[:3]:
    print("head", h, "max x", float(x[h].max()), "max sink", float(xs[h].max()), "Z", float(torch.exp(allx[h]-allx[h].max()).sum()))

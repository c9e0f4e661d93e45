import sys, math, argparse, torch
sys.path.insert(0, ".")
import bench
from kvquant_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
ctx, total, sinks = 8192, 13, 0
max_len = (ctx + total + 8 + 63) // 64 * 64
lay = bench.Layer(4, max_len, gen, dev, sinks)
pos = ctx // 2
plant_q = (torch.randn(32, 128, generator=gen, device=dev) * 3.0).half()
P = ctx + sinks + total
kq = bench.rope_rotate(bench.rope_rotate(plant_q, P), pos + sinks, -1.0)
lay.fill(ctx, gen, dev, plant=(pos, kq.reshape(-1), torch.randn(4096, generator=gen, device=dev)))
q = bench.rope_rotate(plant_q, P).half()
L = lay.k.klen
s = torch.zeros(1, 32, L, device=dev)
ops.score_k(4, q.float().unsqueeze(0).contiguous(), lay.k.kcache, s, lay.k.lookup_table, L, bench.THETA, 0, lay.k.outliers, lay.k.outlier_indices, accumulate=False)
sc = s[0] / math.sqrt(128)
print("planted score per head (first 6):", sc[:6, pos].tolist())
print("expected ~", float((plant_q.float() ** 2).sum(-1)[:1] / 3 / math.sqrt(128)))
print("max other:", float(torch.cat((sc[:, :pos], sc[:, pos + 1:]), 1).max()), "argmax head0:", int(sc[0].argmax()), "pos", pos)
print(bench.check_retrieval(lay, q, dev))

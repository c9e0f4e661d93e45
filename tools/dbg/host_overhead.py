"""development: host time per layer of decode_kv vs GPU time (is the decode loop host-bound on this box?)"""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
import bench
from kvquant_amd import sharding
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
ctx, nl, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 8, 20
max_len = (ctx + 64 + 63) // 64 * 64
layers = []
for li in range(nl):
    lay = bench.Layer(4, max_len, gen, dev, 0)
    lay.fill(ctx, gen, dev)
    layers.append(lay)
k, v = bench.synth_tokens(steps + 3, layers[0].scale, layers[0].shift, gen, dev)
q = torch.randn(steps + 3, 32, 128, generator=gen, device=dev).half()
def run(n0, n):
    for st in range(n0, n0 + n):
        for lay in layers:
            out = bench.layer_step(lay, q[st], k[st], v[st])
    return out
run(0, 3)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(3, 10); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("10 steps x %d layers @128K: host issue %.1f us per layer, wall %.1f us per layer" % (nl, t_host / 10 / nl * 1e6, t_all / 10 / nl * 1e6))
# tiny context: pure host cost
small = []
for li in range(nl):
    lay = bench.Layer(4, 1024, gen, dev, 0)
    lay.fill(512, gen, dev)
    small.append(lay)
layers = small
run(0, 3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(3, 10); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("ctx 512: host issue %.1f us per layer, wall %.1f us per layer" % (t_host / 10 / nl * 1e6, t_all / 10 / nl * 1e6))
cProfile.run("run(13, 5)", "/tmp/hp")
pstats.Stats("/tmp/hp").sort_stats("tottime").print_stats(14)

"""development: what would a hipGraph of the decode step buy?  One token through 32 layers at a FIXED context length
(the replays overwrite the same cache column: timing only), eager vs captured-and-replayed."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from kvquant_amd import cache as kc
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nl = 32
max_len = (ctx + 64 + 63) // 64 * 64
layers = []
for li in range(nl):
    lay = bench.Layer(4, max_len, gen, dev, 0)
    lay.fill(ctx, gen, dev)
    layers.append(lay)
k, v = bench.synth_tokens(4, layers[0].scale, layers[0].shift, gen, dev)
q = torch.randn(4, 32, 128, generator=gen, device=dev).half()


def step(i):
    for lay in layers:
        lay.k.klen = ctx
        lay.v.vlen = ctx
        out = bench.layer_step(lay, q[i], k[i], v[i])
    return out


for i in range(3):
    step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    step(i % 4)
e1.record(); torch.cuda.synchronize()
print("ctx %d eager: %.3f ms/step" % (ctx, e0.elapsed_time(e1) / 20))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(0)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    step(0)
torch.cuda.synchronize()
for i in range(3):
    g.replay()
torch.cuda.synchronize()
e0.record()
for i in range(20):
    g.replay()
e1.record(); torch.cuda.synchronize()
print("ctx %d graph replay: %.3f ms/step" % (ctx, e0.elapsed_time(e1) / 20))

"""development: phase timeline of the score kernel's head loop (build with -DKVQ_TRACE=1, see tools/abl/build_var.sh)"""
import math, os, sys
import torch
sys.path.insert(0, ".")
import kvquant_amd._lib as _l
_l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops
H, HD, C = 32, 128, 4096
bits, L = 4, 131072 + 77
dev = torch.device("cuda")
n, W = 16, 16
max_len = (L + 127) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(0)
k = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
kv = torch.randn(max_len, 42, device=dev, generator=g)
ki = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
kvt, kit = kv.t().contiguous(), ki.t().contiguous()
lut = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
q = torch.randn(1, H, HD, device=dev, generator=g)
trace = torch.zeros(1024 * 8 * 32 * 8, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
s = torch.zeros(1, H, L, device=dev)
ops.score_k(bits, q, k, torch.zeros(1, H, 1, device=dev), lut, 1, 10000.0, 0, accumulate=False)
ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
for it in range(3):
    trace.zero_()
    ops.score_k_prepared_softmax(bits, k, s, lut, L, 10000.0, 0, ws, kv, ki, 1 / math.sqrt(HD), n_parts, kvt, kit)
torch.cuda.synchronize()
t = trace.view(1024, 8, 32, 8).cpu().double()
t = t[:512]                                   # full tiles only
st = t[..., :6]
d = st[..., 1:] - st[..., :-1]               # wait_vm, barrier, fetch, dense, sparse
loop = st[:, :, 1:, 0] - st[:, :, :-1, 0]    # head period
names = ["vm_wait", "barrier", "fetch", "dense", "sparse"]
print("mean cycles per head iteration (memtime ticks = 100 MHz? report raw): period %.0f" % loop.mean())
for i, nm in enumerate(names):
    print("  %-8s mean %8.1f  p50 %8.1f  p90 %8.1f" % (nm, d[..., i].mean(), d[..., i].flatten().median(), d[..., i].flatten().quantile(0.9)))
back = st[:, :, 1:, 0] - st[:, :, :-1, 5]
print("  loop back-edge %.1f" % back.mean())
tot = st[:, :, 31, 5] - st[:, :, 0, 0]
print("head loop total per wave: mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
start = st[:, :, 0, 0]
print("first-head start spread over blocks: %.0f .. %.0f (ticks), end %.0f .. %.0f" % (start.min() - start.min(), start.max() - start.min(), (st[:, :, 31, 5]).min() - start.min(), (st[:, :, 31, 5]).max() - start.min()))

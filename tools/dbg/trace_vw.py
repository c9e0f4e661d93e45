"""development: phase timeline of the 1024-lane p.V kernel (kvq_mix_v_wide.hip; build with -DKVQ_TRACE=1, select with KVQ_LIB)"""
import math, os, sys
import torch
sys.path.insert(0, ".")
import kvquant_amd._lib as _l
_l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops
H, HD, C = 32, 128, 4096
bits = int(os.environ.get("BITS", "4"))
L = int(os.environ.get("CTX", str(131072 + 77)))
dev = torch.device("cuda")
n, W = 1 << bits, 4 * bits
max_len = (L + 127) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
vv = torch.randn(max_len, 42, device=dev, generator=g)
vi = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
rows = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
trace = torch.zeros(512 * 8 * 16, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
out = torch.zeros(1, H, HD, device=dev)
raw = torch.randn(1, H, L, device=dev, generator=g) * 30
inv = 1.0 / math.sqrt(HD)
sc = (raw[0].half() * torch.tensor(inv, device=dev).half()).float()
M = sc.max(dim=-1).values
Z = torch.exp(sc - M[:, None]).sum(dim=-1)
parts = torch.stack((M, Z), dim=-1).reshape(H, 1, 2).contiguous()
for it in range(3):
    trace.zero_()
    ops.mix_v_softmax(bits, raw, parts, 1, inv, v, out, rows, L, vv, vi)
torch.cuda.synchronize()
t = trace.view(256, 16, 16).cpu().double()
nc = t[:, 0, 11].max()
ok = t[:, 0, 11] == nc
print("blocks with %d chunks: %d of %d" % (nc, int(ok.sum()), int((t[:, 0, 11] > 0).sum())))
t = t[t[:, 0, 11] > 0]
names = ["data wait", "barrier", "entries+issue+convert", "look-up loop", "prologue", "epilogue"]
tot = t[..., :6].sum(-1)
for i, nm in enumerate(names):
    per = t[..., i] / (t[..., 11] if i < 4 else 1)
    print("  %-22s mean %9.1f per %s   (%.1f %% of the wave)" % (nm, per.mean(), "chunk" if i < 4 else "wave", 100 * t[..., i].mean() / tot.mean()))
print("wave total %.0f cycles (min %.0f max %.0f)" % (tot.mean(), tot.min(), tot.max()))
clk = t[..., 12] / t[..., 13] * 100.0
print("effective shader clock: %.0f MHz; wave lifetime %.1f us (min %.1f max %.1f)" % (clk.mean(), (t[..., 13] / 100.0).mean(), (t[..., 13] / 100.0).min(), (t[..., 13] / 100.0).max()))
w = t.mean(dim=0)
print("per wave (mean over blocks): wait / barrier per chunk")
for k in range(16):
    print("   wave %2d  wait %7.1f  barrier %7.1f  loop %7.1f" % (k, w[k, 0] / w[k, 11], w[k, 1] / w[k, 11], w[k, 3] / w[k, 11]))

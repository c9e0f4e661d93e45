"""development: phase timeline of the p.V kernel's chunk loop (build with -DKVQ_TRACE=1)"""
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
v = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
vv = torch.randn(max_len, 42, device=dev, generator=g)
vi = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
rows = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
p = torch.softmax(torch.randn(1, H, L, device=dev, generator=g), dim=-1).contiguous()
trace = torch.zeros(512 * 8 * 16, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
out = torch.zeros(1, H, HD, device=dev)
fused = os.environ.get("FUSED") == "1"
if fused:
    raw = torch.randn(1, H, L, device=dev, generator=g) * 30
    inv = 1.0 / math.sqrt(HD)
    sc = (raw[0].half() * torch.tensor(inv, device=dev).half()).float()
    M = sc.max(dim=-1).values
    Z = torch.exp(sc - M[:, None]).sum(dim=-1)
    parts = torch.stack((M, Z), dim=-1).reshape(H, 1, 2).contiguous()
for it in range(3):
    trace.zero_()
    if fused:
        ops.mix_v_softmax(bits, raw, parts, 1, inv, v, out, rows, L, vv, vi)
    else:
        ops.mix_v(bits, p, v, out, rows, L, vv, vi, accumulate=False)
torch.cuda.synchronize()
t = trace.view(512, 8, 16).cpu().double()
nc = t[:, 0, 11].max()
ok = t[:, 0, 11] == nc
t = t[ok]
print("blocks with %d chunks:" % nc, int(ok.sum()))
names = ["loop back-edge", "dma_wait", "barrier", "issue", "math", "slot reduce", "sp: loads+stage+zero", "sp: barrier", "sp: consume", "sp: barrier2", "sp: slab write"]
tot = t[..., :11].sum(-1)
for i, nm in enumerate(names):
    per = t[..., i] / (nc if i < 5 else 1)
    print("  %-22s mean %9.1f per %s   (%.1f %% of the wave)" % (nm, per.mean(), "chunk" if i < 5 else "wave", 100 * t[..., i].mean() / tot.mean()))
print("wave total %.0f cycles (min %.0f max %.0f)" % (tot.mean(), tot.min(), tot.max()))
clk = t[..., 12] / t[..., 13] * 100.0
print("effective shader clock while the kernel runs: %.0f MHz (min %.0f max %.0f); wave lifetime %.1f us" % (clk.mean(), clk.min(), clk.max(), (t[..., 13] / 100.0).mean()))
# who is slow: the first / second workgroup of a CU (dispatch order: blocks [0, 256) come first), odd / even blocks
tt = trace.view(512, 8, 16).cpu().double()
life = tt[..., 13].mean(dim=1) / 100.0
print("wave lifetime (us): blocks 0..255 mean %.1f, blocks 256..511 mean %.1f; even blocks %.1f, odd blocks %.1f; min %.1f max %.1f"
      % (life[:256].mean(), life[256:].mean(), life[0::2].mean(), life[1::2].mean(), life.min(), life.max()))
print("outlier-phase passes per workgroup: mean %.2f, workgroups with a second pass: %d of %d" % (tt[:, 0, 14].mean(), int((tt[:, 0, 14] > 1).sum()), tt.shape[0]))

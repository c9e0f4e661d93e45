"""development: kernel-level timeline of the score kernel at a SHORT context (4-wave tiles, small head groups): entry, loop
start, loop end, exit of every wave (100 MHz wall clock) -- build with -DKVQ_TRACE=1 (tools/abl/build_var.sh trk "-DKVQ_TRACE=1" "")
usage: KVQ_LIB=tools/abl/libkvq_trk.so python tools/dbg/trace_k_short.py [L]"""
import math, os, sys
import torch
sys.path.insert(0, ".")
import kvquant_amd._lib as _l
_l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops
H, HD, C = 32, 128, 4096
bits = 4
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NW = 8 if L >= 16384 else 4
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
trace = torch.zeros(1024 * 8 * 32 * 8 + 1024 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
s = torch.zeros(1, H, L, device=dev)
ops.score_k(bits, q, k, torch.zeros(1, H, 1, device=dev), lut, 1, 10000.0, 0, accumulate=False)
ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for it in range(3):
    trace.zero_()
    flush.fill_(it)                      # (cold caches, as in a 32-layer rotation)
    ops.score_k_prepared_softmax(bits, k, s, lut, L, 10000.0, 0, ws, kv, ki, 1 / math.sqrt(HD), n_parts, kvt, kit)
torch.cuda.synchronize()
tl = trace[1024 * 8 * 32 * 8:].view(-1, 8).cpu()
used = (tl[:, 0] != 0)
T = tl[used][:, [0, 1, 2, 3, 6]].double() / 100.0
T = T - T[:, 0].min()
print("L = %d: %d waves traced (%d workgroups of %d waves)" % (L, int(used.sum()), int(used.sum()) // NW, NW))
for nm, kk in (("entry", 0), ("loop start", 1), ("loop end", 2), ("tails end", 4), ("exit", 3)):
    x = T[:, kk]
    print("  %-10s min %6.1f  mean %6.1f  p90 %6.1f  max %6.1f us" % (nm, x.min(), x.mean(), x.quantile(0.9), x.max()))
print("  per wave: prologue %.1f  head loop %.1f  outlier tails %.1f  epilogue %.1f us"
      % ((T[:, 1] - T[:, 0]).mean(), (T[:, 2] - T[:, 1]).mean(), (T[:, 4] - T[:, 2]).mean(), (T[:, 3] - T[:, 4]).mean()))

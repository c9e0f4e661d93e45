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
trace = torch.zeros(1024 * 8 * 32 * 8 + 1024 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
s = torch.zeros(1, H, L, device=dev)
ops.score_k(bits, q, k, torch.zeros(1, H, 1, device=dev), lut, 1, 10000.0, 0, accumulate=False)
ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
for it in range(3):
    trace.zero_()
    ops.score_k_prepared_softmax(bits, k, s, lut, L, 10000.0, 0, ws, kv, ki, 1 / math.sqrt(HD), n_parts, kvt, kit)
torch.cuda.synchronize()
tl = trace[1024 * 8 * 32 * 8:].view(1024, 8, 8).cpu()
t = trace[:1024 * 8 * 32 * 8].view(1024, 8, 32, 8).cpu().double()
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
_tl = trace[1024 * 8 * 32 * 8:].view(1024, 8, 8).cpu()[:512].double()
_us = ((_tl[..., 2] - _tl[..., 1]) / 100.0)
print("shader clock during the head loop: %.0f MHz (s_memtime cycles of the loop / its s_memrealtime span, mean over waves)"
      % (tot / _us).mean())
# kernel-level timeline (100 MHz wall clock): entry, loop start, loop end, exit of every wave; where it ran
nb = 512 + 32 if (tl[512:544, 0, 0] != 0).all() else 512
T = tl[:nb, :, :4].double() / 100.0            # us
t0 = T[..., 0].min()
T = T - t0
full = T[:512]
print("timeline (us from the first wave's entry), full-tile workgroups:")
for nm, k in (("entry", 0), ("loop start", 1), ("loop end", 2), ("exit", 3)):
    x = full[..., k]
    print("  %-10s min %6.1f  mean %6.1f  p90 %6.1f  max %6.1f" % (nm, x.min(), x.mean(), x.flatten().quantile(0.9), x.max()))
print("  prologue %.1f  loop %.1f (min %.1f max %.1f)  epilogue %.1f   [per wave means]"
      % ((full[..., 1] - full[..., 0]).mean(), (full[..., 2] - full[..., 1]).mean(), (full[..., 2] - full[..., 1]).min(),
         (full[..., 2] - full[..., 1]).max(), (full[..., 3] - full[..., 2]).mean()))
if nb > 512:
    tail = T[512:nb]
    print("  ragged-tile workgroups: entry %.1f .. %.1f, exit %.1f .. %.1f" % (tail[..., 0].min(), tail[..., 0].max(), tail[..., 3].min(), tail[..., 3].max()))
print("kernel span (first entry .. last exit): %.1f us" % T[..., 3].max())
xcc = tl[:512, 0, 5] & 0xf
hw = tl[:512, 0, 4]
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
wg_loop = (full[..., 2] - full[..., 1]).mean(dim=1)
wg_exit = full[..., 3].max(dim=1).values
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCC %d: %3d workgroups, loop %.1f us (min %.1f max %.1f), last exit %.1f" % (x, int(m.sum()), wg_loop[m].mean(), wg_loop[m].min(), wg_loop[m].max(), wg_exit[m].max()))
# the two workgroups of a CU
key = (xcc * 4096 + se * 256 + sh * 64 + cu).tolist()
from collections import defaultdict
d = defaultdict(list)
for i, k in enumerate(key):
    d[k].append(i)
sizes = defaultdict(int)
for k, v in d.items():
    sizes[len(v)] += 1
print("  workgroups per CU histogram:", dict(sizes))
pairs = [v for v in d.values() if len(v) == 2]
if pairs:
    import statistics
    dif = [abs(float(wg_loop[a] - wg_loop[b])) for a, b in pairs]
    ent = [abs(float(full[a, 0, 0] - full[b, 0, 0])) for a, b in pairs]
    print("  co-resident pairs: |loop time difference| mean %.2f us, |entry difference| mean %.2f us" % (statistics.mean(dif), statistics.mean(ent)))
order = torch.argsort(wg_loop)
print("  slowest 8 workgroups (block, xcc, loop us):", [(int(i), int(xcc[i]), round(float(wg_loop[i]), 1)) for i in order[-8:]])
print("  fastest 8 workgroups (block, xcc, loop us):", [(int(i), int(xcc[i]), round(float(wg_loop[i]), 1)) for i in order[:8]])

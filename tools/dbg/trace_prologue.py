"""development: phase stamps (100 MHz s_memrealtime) of the K / V selection workgroups of the decode prologue.
build: tools/abl/build_var.sh ptrace "" "" "-DKVQ_TRACE=1" is NOT enough (KVQ_TRACE also switches the matvec kernels);
use: KVQ_LIB=tools/abl/libkvq_ptrace.so python tools/dbg/trace_prologue.py"""
import os, sys
import torch
sys.path.insert(0, ".")
import kvquant_amd._lib as _l
_l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops
H, HD, C, bits = 32, 128, 4096, 4
n, W, max_len = 16, 16, 4096
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
kmat = torch.zeros(H, W, max_len, dtype=torch.int32, device=dev)
vmat = torch.zeros(H, W, max_len, dtype=torch.int32, device=dev)
klut = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
lo = -torch.rand(C, device=dev, generator=g) - 1
hi = torch.rand(C, device=dev, generator=g) + 1
vrows = torch.zeros(max_len, n, device=dev)
vsorted = torch.linspace(-1, 1, n, device=dev)
ko, vo = torch.zeros(max_len, 42, device=dev), torch.zeros(max_len, 42, device=dev)
ki, vi = torch.zeros(max_len, 42, dtype=torch.int32, device=dev), torch.zeros(max_len, 42, dtype=torch.int32, device=dev)
kot, kit = torch.zeros(42, max_len, device=dev), torch.zeros(42, max_len, dtype=torch.int32, device=dev)
ends = torch.stack((klut.reshape(C, n)[:, 0], klut.reshape(C, n)[:, -1]), dim=-1).contiguous()
q = torch.randn(H, HD, device=dev, generator=g).half()
k = (torch.randn(C, device=dev, generator=g) * 1.5).half()
v = (torch.randn(C, device=dev, generator=g) * 1.5).half()
trace = torch.zeros(32, dtype=torch.int64, device=dev)
os.environ["KVQ_TRACE_PTR"] = str(trace.data_ptr())
acc = torch.zeros(2, 6)
N = 50
for it in range(N + 5):
    trace.zero_()
    ops.decode_prologue(bits, kmat, klut, klut, k, lo, hi, ko, ki, it, vmat, vrows, vsorted, v, vo, vi, it, q, 21, kot, kit, ends)
    torch.cuda.synchronize()
    t = trace.cpu().double().view(2, 16)
    if it >= 5:
        for r in range(2):
            last = 6 if r == 1 else 5
            for p in range(last):
                acc[r, p] += (t[r, p + 1] - t[r, p]) / 100.0   # us
names = ["loads + keys", "radix select", "tie ranks + membership", "V rows + codes", "scan + outlier row", "pack"]
for r, nm in enumerate(("K selection workgroup", "V append workgroup")):
    print(nm)
    for p in range(6):
        print("   %-26s %6.2f us" % (names[p], acc[r, p] / N))
    print("   total %.2f us" % (acc[r].sum() / N))
# pruning select (kvq_select.h), last iteration: stamps 8 .. 12 = zero + barrier | collect | barrier | resolve | barrier
t = trace.cpu().double().view(2, 16)
for r, nm in enumerate(("K", "V")):
    if t[r, 8] > 0:
        print(nm, "select detail (us): bounds %.2f, barrier %.2f, collect %.2f, barrier %.2f, resolve %.2f, barrier %.2f, ->2 %.2f" % (
            (t[r, 7] - t[r, 1]) / 100, (t[r, 8] - t[r, 7]) / 100, (t[r, 9] - t[r, 8]) / 100, (t[r, 10] - t[r, 9]) / 100, (t[r, 11] - t[r, 10]) / 100,
            (t[r, 12] - t[r, 11]) / 100, (t[r, 2] - t[r, 12]) / 100))

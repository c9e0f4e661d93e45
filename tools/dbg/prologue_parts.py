"""development: the decode prologue against its parts launched alone (K fused append, V fused append): where do
its ~17 us go?  usage: python tools/dbg/prologue_parts.py"""
import os, sys
import torch
sys.path.insert(0, ".")
if os.environ.get("KVQ_LIB"):
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
kf, vf = k.float(), v.float()


def t(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


col = [0]
def pro():
    ops.decode_prologue(bits, kmat, klut, klut, k, lo, hi, ko, ki, col[0] % max_len, vmat, vrows, vsorted, v, vo, vi,
                        col[0] % max_len, q, 21, kot, kit, ends)
    col[0] += 1
def ka():
    ops.append_k_fused(bits, kmat, klut, klut, kf, lo, hi, ko, ki, 21, col[0] % max_len, kot, kit)
    col[0] += 1
def va():
    ops.append_v_fused(bits, vmat, vrows, vsorted, vf, vo, vi, 21, col[0] % max_len)
    col[0] += 1
def nop():
    ops.rope_freqs(10000.0) if False else None
x = torch.zeros(1, device=dev)
print("back-to-back launches of a trivial kernel: %.1f us" % t(lambda: x.add_(1)))
print("decode prologue (one launch):              %.1f us" % t(pro))
print("K fused append alone:                      %.1f us" % t(ka))
print("V fused append alone:                      %.1f us" % t(va))

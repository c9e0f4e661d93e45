"""Shapes of the one-workgroup-per-CU p.V kernel (kvq_mix_v_wide.hip) that the other suites do not reach at their sizes:
head counts other than 32 (partial unit groups, two unit groups), ragged lengths, both softmax modes, compact entries, no
entries.  Collected only when named (tests/test_wide_gpu.py runs it in a process with KVQ_V_WIDE_FROM=1, i.e. with the wide
geometry at every length); the checker is a plain fp64 restatement of KCU:3211-3433 + 437-470 (dequantise, add the sparse
residuals, multiply with the probabilities)."""
import math

import pytest
import torch

from kvquant_amd import ops
from tests import util

pytestmark = pytest.mark.gpu
HD = 128


def _dequant(bits, mat, rows, L):
    """[H, HD, L] f64 from packed words [H, W, max_len] and per-token codebook rows [max_len, N]"""
    H = mat.shape[0]
    codes = util.unpack_codes(mat.cpu(), bits, L)                     # [C, L]
    r = rows[:L].cpu().double()                                       # [L, N]
    return r[torch.arange(L)[None, :], codes].reshape(H, HD, L)


def _reference(bits, mat, rows, p, L, vals, idx):
    H = mat.shape[0]
    v = _dequant(bits, mat, rows, L)                # [H, HD, L]
    if idx is not None:
        flat = v.reshape(H * HD, L)
        iv = idx[:L].cpu().long()
        if vals is None:                             # compact: fp16 residual << 16 | channel
            ch = iv & 0xffff
            x = ((iv >> 16) & 0xffff).to(torch.int32).to(torch.int16).view(torch.float16).double()
        else:
            ch, x = iv, vals[:L].cpu().double()
        for s in range(ch.shape[1]):
            ok = ch[:, s] < H * HD
            flat[ch[ok, s], torch.arange(L)[ok]] += x[ok, s]
    return torch.einsum("hl,hcl->hc", p.cpu().double(), v)


def _case(bits, H, L, seed, sparse=True, compact=False):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    n, W = 2 ** bits, HD // 32 * bits
    max_len = (L + 64 + 63) // 64 * 64
    mat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), generator=g, dtype=torch.int64).to(torch.int32)
    rows = torch.zeros(max_len, n)
    rows[:L] = util.centroids(bits).unsqueeze(0) * (torch.rand(L, 1, generator=g) + 0.5) + torch.randn(L, 1, generator=g) * 0.1
    C = H * HD
    vals = idx = None
    if sparse:
        n_out = 42 if C >= 42 else C
        idx = torch.zeros(max_len, n_out, dtype=torch.int32)
        for t in range(L):
            idx[t] = torch.sort(torch.randperm(C, generator=g)[:n_out]).values.int()
        vals = torch.zeros(max_len, n_out)
        vals[:L] = torch.randn(L, n_out, generator=g) * 3
        vals[:L][torch.rand(L, n_out, generator=g) < 0.3] = 0.0
        if compact:
            h = vals.half()
            idx = ((h.view(torch.int16).to(torch.int32) & 0xffff) << 16) | idx
            vals = None
    return dev, mat, rows, vals, idx, max_len


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("H", [1, 4, 11, 28, 32, 40, 64])
@pytest.mark.parametrize("L", [1, 31, 33, 65, 1000, 7013])
def test_wide_probabilities_from_memory(bits, H, L):
    dev, mat, rows, vals, idx, max_len = _case(bits, H, L, 7 * H + L + bits)
    g = torch.Generator().manual_seed(L)
    p = torch.softmax(torch.randn(1, H, L, generator=g) * 2, dim=-1).half().float().contiguous()
    out = torch.zeros(1, H, HD, device=dev)
    ops.mix_v(bits, p.to(dev), mat.to(dev), out, rows.to(dev), L, vals.to(dev), idx.to(dev), accumulate=False)
    ref = _reference(bits, mat, rows, p[0], L, vals, idx)
    err = util.rel_err(out.cpu().reshape(1, -1), ref.float().reshape(1, -1))
    assert err < 1e-3, (bits, H, L, err)


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("H,L,mode", [(32, 777, "compact"), (11, 300, "compact"), (32, 900, "dense"), (40, 129, "dense"),
                                      (32, 20000, "sparse"), (8, 40000, "sparse")])
def test_wide_fused_softmax_and_formats(bits, H, L, mode):
    dev, mat, rows, vals, idx, max_len = _case(bits, H, L, 3 * H + L + bits, sparse=mode != "dense", compact=mode == "compact")
    g = torch.Generator().manual_seed(L + 5)
    raw = (torch.randn(1, H, L, generator=g) * 8).contiguous()
    inv = 1.0 / math.sqrt(HD)
    sc = (raw[0].half() * torch.tensor(inv).half()).float()
    p = torch.softmax(sc, dim=-1).half().float()
    # the score kernel's per-tile partials, restated: (max, sum exp) of every 256-token tile (128 below 16K tokens)
    T = 256 if L >= 16384 else 128
    n_parts = (L + T - 1) // T
    parts = torch.zeros(H, n_parts, 2)
    for i in range(n_parts):
        blk = sc[:, i * T:(i + 1) * T]
        m = blk.max(dim=-1).values
        parts[:, i, 0] = m
        parts[:, i, 1] = torch.exp(blk - m[:, None]).sum(dim=-1)
    out = torch.zeros(1, H, HD, device=dev)
    if mode == "dense":         # (no outlier rows: the probabilities-from-memory entry point; the fused one is the sparse decode path's)
        ops.mix_v(bits, p[None].contiguous().to(dev), mat.to(dev), out, rows.to(dev), L, accumulate=False)
    else:
        ops.mix_v_softmax(bits, raw.to(dev), parts.contiguous().to(dev), n_parts, inv, mat.to(dev), out, rows.to(dev), L,
                          None if vals is None else vals.to(dev), idx.to(dev))
    ref = _reference(bits, mat, rows, p, L, vals, idx)
    err = util.rel_err(out.cpu().reshape(1, -1), ref.float().reshape(1, -1))
    # (the probabilities are rounded to fp16 by the kernel's own exponential -- <= 2 ulp of the fp32 quotient before the
    #  rounding, DESIGN.md 3.3 -- and by torch's here: a dominant probability landing on the neighbouring fp16 value moves the
    #  output by 1e-3 of itself; the kernel-level bars against the reference are in tests/test_atsize_gpu.py)
    assert err < 4e-3, (bits, H, L, mode, err)

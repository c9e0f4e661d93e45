"""Seeded shape fuzz of the two matvecs through the C ABI against the C oracle: head counts other than 32 (partial
unit groups of the p.V kernel, head groups of the score kernel), cache lengths around every tile / chunk / range
boundary, rows that are and are not 16-byte aligned (DMA kernel vs row-per-lane fallback), q_len > 1, every bit width,
dense and sparse, accumulate and overwrite."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
HD = 128


def _cases():
    g = torch.Generator().manual_seed(2024)
    out = []
    heads = [1, 2, 5, 8, 12, 32, 37, 40, 64]
    lens = [1, 2, 15, 16, 17, 31, 33, 64, 127, 129, 255, 256, 257, 511, 777, 1024, 1500, 4095, 4097, 9000]
    for i in range(72):
        H = heads[int(torch.randint(0, len(heads), (1,), generator=g))]
        L = lens[int(torch.randint(0, len(lens), (1,), generator=g))]
        bits = [4, 3, 2][i % 3]
        pad = [0, 1, 3, 4, 8, 64][int(torch.randint(0, 6, (1,), generator=g))]
        max_len = L + pad
        if i % 2 == 0:
            max_len = (max_len + 3) // 4 * 4          # the DMA kernel's shapes at least half of the time
        q_len = 1 if i % 5 else 2
        sparse = bool(i % 4 != 3)
        out.append((bits, H, L, max_len, q_len, sparse))
    return out


@pytest.mark.parametrize("bits,H,L,max_len,q_len,sparse", _cases())
def test_matvec_shapes(bits, H, L, max_len, q_len, sparse):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    from oracle import ckernels as ck
    dev = torch.device("cuda:0")
    C = H * HD
    n = 2 ** bits
    W = HD // 32 * bits
    g = torch.Generator().manual_seed(bits * 1000003 + H * 1009 + L)
    kmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), dtype=torch.int64, generator=g).to(torch.int32)
    vmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), dtype=torch.int64, generator=g).to(torch.int32)
    klut = torch.randn(H, HD, n, generator=g).sort(dim=-1).values.contiguous()
    rows = (torch.randn(max_len, n, generator=g) * 0.7).sort(dim=-1).values.contiguous()
    n_out = 42 if C >= 4096 else 6
    vals = torch.randn(max_len, n_out, generator=g) * 0.5
    idx = torch.sort(torch.randint(0, C, (max_len, n_out), generator=g), dim=-1).values.to(torch.int32)
    vals[torch.rand(max_len, n_out, generator=g) < 0.2] = 0.0          # capped-away slots
    q = torch.randn(q_len, H, HD, generator=g)
    p = torch.softmax(torch.randn(q_len, H, L, generator=g) * 2, dim=-1).half().float().contiguous()
    pos_offset = int(torch.randint(0, 7, (1,), generator=g))
    # ---- q.K^T
    ref = torch.zeros(q_len, H, L)
    ck.score_k(bits, q, kmat, ref, klut, L, 10000.0, pos_offset)
    use_sparse_k = sparse and q_len == 1                               # (the reference's sparse score op is q_len = 1)
    if use_sparse_k:
        ck.spmv_k_rope(vals, idx, q, ref, L, 10000.0, pos_offset)
    out = torch.full((q_len, H, L), 3.0, device=dev)
    ops.score_k(bits, q.to(dev), kmat.to(dev), out, klut.to(dev), L, 10000.0, pos_offset,
                vals.to(dev) if use_sparse_k else None, idx.to(dev) if use_sparse_k else None, accumulate=False)
    assert util.rel_err(out.cpu().reshape(q_len, -1), ref.reshape(q_len, -1)) < 1e-4
    ops.score_k(bits, q.to(dev), kmat.to(dev), out, klut.to(dev), L, 10000.0, pos_offset,
                vals.to(dev) if use_sparse_k else None, idx.to(dev) if use_sparse_k else None, accumulate=True)
    assert util.rel_err(out.cpu().reshape(q_len, -1), 2 * ref.reshape(q_len, -1)) < 1e-4
    # ---- p.V
    refv = torch.zeros(q_len, H, HD)
    ck.mix_v(bits, p, vmat, refv, rows, L)
    if sparse:
        ck.spmv_v(vals, idx, p, refv, L)
    outv = torch.full((q_len, H, HD), -2.0, device=dev)
    ops.mix_v(bits, p.to(dev), vmat.to(dev), outv, rows.to(dev), L, vals.to(dev) if sparse else None,
              idx.to(dev) if sparse else None, accumulate=False)
    assert util.rel_err(outv.cpu().reshape(q_len, -1), refv.reshape(q_len, -1)) < 1e-4
    ops.mix_v(bits, p.to(dev), vmat.to(dev), outv, rows.to(dev), L, vals.to(dev) if sparse else None,
              idx.to(dev) if sparse else None, accumulate=True)
    assert util.rel_err(outv.cpu().reshape(q_len, -1), 2 * refv.reshape(q_len, -1)) < 1e-4

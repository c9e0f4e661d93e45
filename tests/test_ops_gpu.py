"""GPU parity: every legacy quant_cuda op of kvquant_amd (through the C ABI)
against the CPU oracle on the same seeded inputs.
Bit-exact for packed codes / rescaled values; q.K^T and p.V within 1e-3 relative
(the north-star tolerance; measured errors are ~1e-6 and asserted at 2e-5)."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

H, HD, C = util.H, util.HD, util.C
TOL = 2e-5   # asserted; the contract is 1e-3 (BASELINE.json north_star)


@pytest.fixture(scope="module")
def qc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import quant_cuda
    return quant_cuda


@pytest.fixture(scope="module")
def orc():
    from oracle import quant_cuda_ref
    return quant_cuda_ref


def W(bits):
    return HD // 32 * bits


def dev(*ts):
    return [t.cuda() for t in ts]


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_append_k_dense_and_sparse(qc, orc, bits):
    lut, lo, hi, scale, shift = util.k_tables(bits)
    xs = util.k_tokens(5, scale, shift)
    max_len = 9
    m_ref = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    m_ref2 = m_ref.clone()
    m_gpu = m_ref.clone().cuda()
    m_gpu2 = m_ref.clone().cuda()
    for i in range(5):
        x = xs[i].contiguous()
        col = [0, 3, 8, 1, 5][i]
        getattr(orc, "vecquant%dappendvecK" % bits)(m_ref, lut, x, col)
        getattr(qc, "vecquant%dappendvecK" % bits)(m_gpu, lut.cuda(), x.cuda(), col)
        r_ref = x.clone()
        r_gpu = x.clone().cuda()
        getattr(orc, "vecquant%dappendvecKsparse" % bits)(m_ref2, lut, x, r_ref, lo, hi, col)
        getattr(qc, "vecquant%dappendvecKsparse" % bits)(m_gpu2, lut.cuda(), x.cuda(), r_gpu, lo.cuda(), hi.cuda(), col)
        assert torch.equal(r_ref.view(torch.int32), r_gpu.cpu().view(torch.int32)), "rescaled not bit-exact"
    assert torch.equal(m_ref, m_gpu.cpu())
    assert torch.equal(m_ref2, m_gpu2.cpu())
    assert torch.equal(m_ref, m_ref2)


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_append_v_dense_and_sparse(qc, orc, bits):
    n = 2 ** bits
    xs = util.v_tokens(4)
    max_len = 6
    rows = torch.zeros(max_len, n)
    m_ref = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    m_ref2 = m_ref.clone()
    for i in range(4):
        x = xs[i].contiguous()
        hi_t = torch.topk(x, 22).values[-1]
        lo_t = torch.topk(x, 22, largest=False).values[-1]
        rows[i] = util.centroids(bits) * ((hi_t - lo_t) / 2) + (hi_t + lo_t) / 2
    m_gpu = m_ref.clone().cuda()
    m_gpu2 = m_ref.clone().cuda()
    rows_g = rows.cuda()
    for i in range(4):
        x = xs[i].contiguous()
        hi_t = float(torch.topk(x, 22).values[-1])
        lo_t = float(torch.topk(x, 22, largest=False).values[-1])
        getattr(orc, "vecquant%dappendvecV" % bits)(m_ref, rows, x, i)
        getattr(qc, "vecquant%dappendvecV" % bits)(m_gpu, rows_g, x.cuda(), i)
        getattr(orc, "vecquant%dappendvecVsparse" % bits)(m_ref2, rows, x, 0.0, lo_t, hi_t, i)
        getattr(qc, "vecquant%dappendvecVsparse" % bits)(m_gpu2, rows_g, x.cuda(), 0.0, lo_t, hi_t, i)
    assert torch.equal(m_ref, m_gpu.cpu())
    assert torch.equal(m_ref2, m_gpu2.cpu())
    assert not torch.equal(m_ref, m_ref2)  # clipping to the zero-point code must show


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("S", [1, 300])
def test_pack_parallel(qc, orc, bits, S):
    n = 2 ** bits
    lut, lo, hi, scale, shift = util.k_tables(bits)
    max_len = S + 7
    k = util.k_tokens(S, scale, shift).t().contiguous().view(H, HD, S)
    v = util.v_tokens(S).t().contiguous().view(H, HD, S)
    # K
    m_ref = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    r_ref = k.clone()
    getattr(orc, "vecquant%dappendvecKsparseParallel" % bits)(m_ref, lut, k, r_ref, lo, hi)
    m_gpu = torch.zeros_like(m_ref).cuda()
    r_gpu = k.clone().cuda()
    getattr(qc, "vecquant%dappendvecKsparseParallel" % bits)(m_gpu, lut.cuda(), k.cuda(), r_gpu, lo.cuda(), hi.cuda())
    assert torch.equal(m_ref, m_gpu.cpu())
    assert torch.equal(r_ref.view(torch.int32), r_gpu.cpu().view(torch.int32))
    # V
    vt = v.reshape(C, S).t()
    hi_t = torch.topk(vt, 22, dim=-1).values[:, -1].contiguous()
    lo_t = torch.topk(vt, 22, dim=-1, largest=False).values[:, -1].contiguous()
    rows = torch.zeros(max_len, n)
    rows[:S] = util.centroids(bits).unsqueeze(0) * ((hi_t - lo_t) / 2).unsqueeze(1) + ((hi_t + lo_t) / 2).unsqueeze(1)
    m_ref = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    getattr(orc, "vecquant%dappendvecVsparseParallel" % bits)(m_ref, rows, v, lo_t, hi_t)
    m_gpu = torch.zeros_like(m_ref).cuda()
    getattr(qc, "vecquant%dappendvecVsparseParallel" % bits)(m_gpu, rows.cuda(), v.cuda(), lo_t.cuda(), hi_t.cuda())
    assert torch.equal(m_ref, m_gpu.cpu())


def _random_cache(bits, L, max_len, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W(bits), max_len), generator=g, dtype=torch.int64).to(torch.int32)
    m[:, :, L:] = 0
    return m


def _outliers(L, max_len, seed, n_out=42):
    g = torch.Generator().manual_seed(seed)
    vals = torch.zeros(max_len, n_out)
    idx = torch.zeros(max_len, n_out, dtype=torch.int32)
    for t in range(L):
        idx[t] = torch.sort(torch.randperm(C, generator=g)[:n_out]).values.int()
    vals[:L] = torch.randn(L, n_out, generator=g) * 3
    vals[:L][torch.rand(L, n_out, generator=g) < 0.3] = 0.0   # capped slots
    return vals, idx


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("L,q_len,pos_offset,sparse", [
    (1, 1, 0, True), (77, 1, 5, True), (700, 2, 0, True), (5000, 1, 0, False), (4097, 1, 123456, True),
    (300, 1, 1000000, True)])
def test_score_k(qc, orc, bits, L, q_len, pos_offset, sparse):
    max_len = L + 3
    lut, _, _, _, _ = util.k_tables(bits, seed=bits)
    mat = _random_cache(bits, L, max_len, 7 + L)
    g = torch.Generator().manual_seed(L)
    q = torch.randn(q_len, H, HD, generator=g).half().float()
    vals, idx = _outliers(L, max_len, L)
    sfx = "opt2" if sparse else "opt"
    name = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_%s" % (bits, sfx)
    ref = torch.zeros(q_len, H, L)
    out = torch.zeros(q_len, H, L).cuda()
    if sparse:
        getattr(orc, name)(q, mat, ref, lut, L, vals, idx, 10000.0, pos_offset)
        getattr(qc, name)(q.cuda(), mat.cuda(), out, lut.cuda(), L, vals.cuda(), idx.cuda(), 10000.0, pos_offset)
    else:
        getattr(orc, name)(q, mat, ref, lut, L, 10000.0, pos_offset)
        getattr(qc, name)(q.cuda(), mat.cuda(), out, lut.cuda(), L, 10000.0, pos_offset)
    err = util.rel_err(out.cpu(), ref)
    assert err < TOL, err
    # accumulate contract: a second call adds on top
    if L <= 700:
        if sparse:
            getattr(qc, name)(q.cuda(), mat.cuda(), out, lut.cuda(), L, vals.cuda(), idx.cuda(), 10000.0, pos_offset)
        else:
            getattr(qc, name)(q.cuda(), mat.cuda(), out, lut.cuda(), L, 10000.0, pos_offset)
        assert util.rel_err(out.cpu(), 2 * ref) < TOL


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("L,q_len,sparse", [(1, 1, True), (63, 1, True), (1000, 2, True), (5001, 1, False),
                                             (8192, 1, True)])
def test_mix_v(qc, orc, bits, L, q_len, sparse):
    n = 2 ** bits
    max_len = L + 2
    mat = _random_cache(bits, L, max_len, 11 + L)
    g = torch.Generator().manual_seed(L + 1)
    rows = torch.zeros(max_len, n)
    sf = torch.rand(L, generator=g) + 0.5
    off = torch.randn(L, generator=g) * 0.1
    rows[:L] = util.centroids(bits).unsqueeze(0) * sf.unsqueeze(1) + off.unsqueeze(1)
    p = torch.softmax(torch.randn(q_len, H, L, generator=g) * 2, dim=-1).half().float().contiguous()
    vals, idx = _outliers(L, max_len, L + 2)
    sfx = "opt2" if sparse else "opt"
    name = "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_%s" % (bits, sfx)
    ref = torch.zeros(q_len, H, HD)
    out = torch.zeros(q_len, H, HD).cuda()
    if sparse:
        getattr(orc, name)(p, mat, ref, rows, L, vals, idx)
        getattr(qc, name)(p.cuda(), mat.cuda(), out, rows.cuda(), L, vals.cuda(), idx.cuda())
    else:
        getattr(orc, name)(p, mat, ref, rows, L)
        getattr(qc, name)(p.cuda(), mat.cuda(), out, rows.cuda(), L)
    err = util.rel_err(out.cpu().reshape(q_len, -1), ref.reshape(q_len, -1))
    assert err < TOL, err


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("L", [700, 40000, 70000])   # (70000: the token-split outlier phase of long 4-bit caches)
@pytest.mark.parametrize("skew", ["low", "high", "33+9", "9+33", "one_channel", "mixed"])
def test_mix_v_skewed_outlier_rows(qc, orc, bits, L, skew):
    """The outlier phase of the streaming p.V kernel (kvq_mix_v.hip, sparse_phase: per-group rows, the straight-line
    single-round form of one-unit-group launches, and -- 4 bit from 64K tokens -- the entries split by tokens between the
    unit groups with extra slabs for the other group's channels; 3 / 2 bit take the one-workgroup-per-CU plan up to 48K
    tokens): rows whose 42 entries sit entirely / mostly in one half of the channels, rows that all name the same
    channels (same-address LDS adds), against the oracle."""
    n = 2 ** bits
    max_len = (L + 64 + 63) // 64 * 64          # aligned rows: the streaming kernel, not the row-per-lane fallback
    mat = _random_cache(bits, L, max_len, 5 + L)
    g = torch.Generator().manual_seed(L + 7)
    rows = torch.zeros(max_len, n)
    rows[:L] = util.centroids(bits).unsqueeze(0) * (torch.rand(L, 1, generator=g) + 0.5) + torch.randn(L, 1, generator=g) * 0.1
    p = torch.softmax(torch.randn(1, H, L, generator=g) * 2, dim=-1).half().float().contiguous()
    vals = torch.zeros(max_len, 42)
    idx = torch.zeros(max_len, 42, dtype=torch.int32)
    half = C // 2
    for t in range(L):
        kind = skew if skew != "mixed" else ["low", "high", "33+9", "9+33", "uniform"][t % 5]
        if kind == "low":
            ch = torch.randperm(half, generator=g)[:42]
        elif kind == "high":
            ch = half + torch.randperm(half, generator=g)[:42]
        elif kind == "33+9":
            ch = torch.cat((torch.randperm(half, generator=g)[:33], half + torch.randperm(half, generator=g)[:9]))
        elif kind == "9+33":
            ch = torch.cat((torch.randperm(half, generator=g)[:9], half + torch.randperm(half, generator=g)[:33]))
        elif kind == "one_channel":
            ch = torch.arange(42) * 97 + 5          # every token names the same 42 channels
        else:
            ch = torch.randperm(C, generator=g)[:42]
        idx[t] = torch.sort(ch).values.int()
    vals[:L] = torch.randn(L, 42, generator=g) * 3
    name = "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt2" % bits
    ref = torch.zeros(1, H, HD)
    out = torch.zeros(1, H, HD).cuda()
    getattr(orc, name)(p, mat, ref, rows, L, vals, idx)
    getattr(qc, name)(p.cuda(), mat.cuda(), out, rows.cuda(), L, vals.cuda(), idx.cuda())
    err = util.rel_err(out.cpu().reshape(1, -1), ref.reshape(1, -1))
    # (one_channel at 70000 tokens sums 70000 signed residuals per channel: the oracle's sequential fp32 sum and the
    #  kernel's exact fixed-point sum differ by 2e-5 of the row maximum; contract 1e-3)
    assert err < 5 * TOL, err


def test_bad_arguments_raise(qc):
    m = torch.zeros(H, 16, 8, dtype=torch.int32).cuda()
    lut = torch.zeros(H, HD, 16).cuda()
    with pytest.raises(ValueError):
        qc.vecquant4appendvecK(m, lut, torch.zeros(C), 0)           # CPU tensor
    with pytest.raises(ValueError):
        qc.vecquant4appendvecK(m, lut.half(), torch.zeros(C).cuda(), 0)  # wrong dtype
    from kvquant_amd._lib import KvqError
    with pytest.raises(KvqError):
        qc.vecquant4appendvecK(m, lut, torch.zeros(C).cuda(), 8)    # column out of range


def test_score_k_unsorted_outlier_rows(qc, orc):
    """the reference's SpMV is one atomic per entry and does not care about the order of a token's entries;
    its glue happens to sort them, which the fast path exploits -- unsorted rows must still be right"""
    bits, L = 4, 700
    max_len = L + 4
    lut, _, _, _, _ = util.k_tables(bits, seed=3)
    mat = _random_cache(bits, L, max_len, 99)
    g = torch.Generator().manual_seed(17)
    q = torch.randn(1, H, HD, generator=g)
    vals, idx = _outliers(L, max_len, 5)
    perm = torch.stack([torch.randperm(42, generator=g) for _ in range(max_len)])
    vals, idx = torch.gather(vals, 1, perm), torch.gather(idx, 1, perm)
    # duplicates of one (token, head) far apart inside a row as well
    idx[10, 0], idx[10, 41] = 5, 6
    vals[10, 0], vals[10, 41] = 1.5, -2.5
    ref = torch.zeros(1, H, L)
    got = torch.zeros(1, H, L).cuda()
    name = "vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2"
    getattr(orc, name)(q, mat, ref, lut, L, vals, idx, 10000.0, 0)
    getattr(qc, name)(q.cuda(), mat.cuda(), got, lut.cuda(), L, vals.cuda(), idx.cuda(), 10000.0, 0)
    assert util.rel_err(got.cpu().reshape(1, -1), ref.reshape(1, -1)) < TOL

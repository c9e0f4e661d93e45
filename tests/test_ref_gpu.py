"""Pins the C oracle AND the HIP kernels to the RUNNING reference.

`oracle/_ref/quant_cuda_ref*.so` is the reference's own extension
(/root/reference/deployment/kvquant/quant_cuda.cpp + quant_cuda_kernel.cu) built
for gfx950 by oracle/build_ref.py (test infrastructure; never imported by the
product).  Every one of its 34 ops is run here on the GPU next to
`kvquant_amd.quant_cuda` (through the C ABI) and next to the CPU oracle on the
same seeded inputs:
  * packed codes, rescaled values, CSR/CSC bookkeeping: bit-exact, three ways;
  * q.K^T / p.V: within 1e-3 relative (the north-star tolerance) -- the reference
    sums with fp32 atomics in an order that changes from run to run, so it is not
    bit-reproducible against itself; measured differences are printed.
Documented reference defects are not hidden: the racy
parallel K pack must agree bit for bit wherever its result does not depend on the race (tests/util.py), the
3-bit parallel V pack (wrong codebook column, KCU:2574-2579) is compared on the
inputs where the defect cannot show (all tokens share one codebook row) and
reported as differing otherwise.
"""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

H, HD, C = util.H, util.HD, util.C
TOL = 1e-3   # BASELINE.json north_star: qK^T / softmax.V within 1e-3 relative


@pytest.fixture(scope="module")
def ref():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import build_ref
    if build_ref.built() is None:
        pytest.skip("oracle/_ref not built (python -m oracle.build_ref needs /root/reference)")
    return build_ref.load()


@pytest.fixture(scope="module")
def qc():
    from kvquant_amd import quant_cuda
    return quant_cuda


@pytest.fixture(scope="module")
def orc():
    from oracle import quant_cuda_ref
    return quant_cuda_ref


def W(bits):
    return HD // 32 * bits


def test_all_34_names_present(ref, qc, orc):
    names = [n for n in dir(ref) if n.startswith("vecquant")]
    assert sorted(names) == orc.NAMES
    for n in names:
        assert hasattr(qc, n)


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_append_k(ref, qc, orc, bits):
    lut, lo, hi, scale, shift = util.k_tables(bits, seed=10 + bits)
    xs = util.k_tokens(6, scale, shift, seed=3)
    max_len = 16
    cols = [0, 3, 8, 1, 15, 5]
    m = {k: torch.zeros(H, W(bits), max_len, dtype=torch.int32) for k in ("o", "o2")}
    g = {k: torch.zeros(H, W(bits), max_len, dtype=torch.int32).cuda() for k in ("r", "r2", "h", "h2")}
    lut_g, lo_g, hi_g = lut.cuda(), lo.cuda(), hi.cuda()
    for i, col in enumerate(cols):
        x = xs[i].contiguous()
        xg = x.cuda()
        getattr(orc, "vecquant%dappendvecK" % bits)(m["o"], lut, x, col)
        getattr(ref, "vecquant%dappendvecK" % bits)(g["r"], lut_g, xg, col)
        getattr(qc, "vecquant%dappendvecK" % bits)(g["h"], lut_g, xg, col)
        ro, rr, rh = torch.zeros(C), torch.zeros(C).cuda(), torch.zeros(C).cuda()
        getattr(orc, "vecquant%dappendvecKsparse" % bits)(m["o2"], lut, x, ro, lo, hi, col)
        getattr(ref, "vecquant%dappendvecKsparse" % bits)(g["r2"], lut_g, xg, rr, lo_g, hi_g, col)
        getattr(qc, "vecquant%dappendvecKsparse" % bits)(g["h2"], lut_g, xg, rh, lo_g, hi_g, col)
        assert torch.equal(rr.view(torch.int32), rh.view(torch.int32)), "rescaled: HIP != reference"
        assert torch.equal(rr.cpu().view(torch.int32), ro.view(torch.int32)), "rescaled: oracle != reference"
    assert torch.equal(g["r"], g["h"]) and torch.equal(g["r2"], g["h2"]), "packed codes: HIP != reference"
    assert torch.equal(g["r"].cpu(), m["o"]) and torch.equal(g["r2"].cpu(), m["o2"]), "packed codes: oracle != reference"


def _v_rows(bits, xs, max_len):
    n = 2 ** bits
    rows = torch.zeros(max_len, n)
    los, his = [], []
    for i in range(xs.shape[0]):
        hi_t = torch.topk(xs[i], 22).values[-1]
        lo_t = torch.topk(xs[i], 22, largest=False).values[-1]
        rows[i] = util.centroids(bits) * ((hi_t - lo_t) / 2) + (hi_t + lo_t) / 2
        los.append(float(lo_t))
        his.append(float(hi_t))
    return rows, los, his


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_append_v(ref, qc, orc, bits):
    xs = util.v_tokens(5, seed=21)
    max_len = 8
    rows, los, his = _v_rows(bits, xs, max_len)
    m = {k: torch.zeros(H, W(bits), max_len, dtype=torch.int32) for k in ("o", "o2")}
    g = {k: torch.zeros(H, W(bits), max_len, dtype=torch.int32).cuda() for k in ("r", "r2", "h", "h2")}
    rows_g = rows.cuda()
    for i in range(5):
        x = xs[i].contiguous()
        xg = x.cuda()
        getattr(orc, "vecquant%dappendvecV" % bits)(m["o"], rows, x, i)
        getattr(ref, "vecquant%dappendvecV" % bits)(g["r"], rows_g, xg, i)
        getattr(qc, "vecquant%dappendvecV" % bits)(g["h"], rows_g, xg, i)
        getattr(orc, "vecquant%dappendvecVsparse" % bits)(m["o2"], rows, x, 0.0, los[i], his[i], i)
        getattr(ref, "vecquant%dappendvecVsparse" % bits)(g["r2"], rows_g, xg, 0.0, los[i], his[i], i)
        getattr(qc, "vecquant%dappendvecVsparse" % bits)(g["h2"], rows_g, xg, 0.0, los[i], his[i], i)
    assert torch.equal(g["r"], g["h"]) and torch.equal(g["r2"], g["h2"]), "HIP != reference"
    assert torch.equal(g["r"].cpu(), m["o"]) and torch.equal(g["r2"].cpu(), m["o2"]), "oracle != reference"


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("S", [1, 130, 300])
def test_pack_k_parallel(ref, qc, orc, bits, S):
    """KCU:1829-1898 / 2275-2366 / 2765-2834 read LDS written by the block's other threads without a
    barrier: compared wherever the reference's result is defined (tests/util.py)."""
    lut, lo, hi, scale, shift = util.k_tables(bits, seed=20 + bits)
    max_len = S + 5
    k = util.k_tokens(S, scale, shift, seed=S).t().contiguous().view(H, HD, S)
    name = "vecquant%dappendvecKsparseParallel" % bits
    mo, ro = torch.zeros(H, W(bits), max_len, dtype=torch.int32), torch.zeros(H, HD, S)
    getattr(orc, name)(mo, lut, k, ro, lo, hi)
    mh, rh = torch.zeros_like(mo).cuda(), torch.zeros(H, HD, S).cuda()
    getattr(qc, name)(mh, lut.cuda(), k.cuda(), rh, lo.cuda(), hi.cuda())
    assert torch.equal(mh.cpu(), mo) and torch.equal(rh.cpu().view(torch.int32), ro.view(torch.int32))
    rw, rv = util.ref_pack_k_agrees_where_defined(ref, name, lut.cuda(), k.cuda(), lo.cuda(), hi.cuda(), mh, rh)
    if sum(rw) + sum(rv):
        print("reference parallel K pack bits=%d S=%d: racy words per run %s, racy rescaled values per run %s"
              % (bits, S, rw, rv))


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("S", [1, 130, 300])
def test_pack_v_parallel(ref, qc, orc, bits, S):
    n = 2 ** bits
    max_len = S + 5
    v = util.v_tokens(S, seed=S + 1).t().contiguous().view(H, HD, S)
    vt = v.reshape(C, S).t()
    hi_t = torch.topk(vt, 22, dim=-1).values[:, -1].contiguous()
    lo_t = torch.topk(vt, 22, dim=-1, largest=False).values[:, -1].contiguous()
    rows = torch.zeros(max_len, n)
    rows[:S] = util.centroids(bits).unsqueeze(0) * ((hi_t - lo_t) / 2).unsqueeze(1) + ((hi_t + lo_t) / 2).unsqueeze(1)
    name = "vecquant%dappendvecVsparseParallel" % bits
    mo = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    getattr(orc, name)(mo, rows, v, lo_t, hi_t)
    mr = torch.zeros_like(mo).cuda()
    getattr(ref, name)(mr, rows.cuda(), v.cuda(), lo_t.cuda(), hi_t.cuda())
    mh = torch.zeros_like(mo).cuda()
    getattr(qc, name)(mh, rows.cuda(), v.cuda(), lo_t.cuda(), hi_t.cuda())
    assert torch.equal(mh.cpu(), mo), "HIP != oracle"
    if bits == 3:
        # KCU:2574-2579: the 3-bit kernel scans token (block_start + k)'s codebook row instead of its own (for
        # k beyond the block's tokens: LDS nobody wrote), whatever S is
        if not torch.equal(mr, mh):
            frac = float((mr != mh).float().mean())
            pytest.xfail("reference 3-bit parallel V pack uses the wrong codebook column (SURVEY App. B-1b): "
                         "%.1f %% of the packed words differ from the intended semantics" % (100 * frac))
    assert torch.equal(mr, mh), "HIP != reference"


def test_pack_v_parallel_3bit_shared_row(ref, qc, orc):
    """the 3-bit defect cannot show when every token has the same codebook row and every 128-token block is
    full: then the reference's own result must equal ours bit for bit"""
    bits, S = 3, 256
    n = 8
    max_len = S + 3
    v = util.v_tokens(S, seed=77).t().contiguous().view(H, HD, S)
    hi_t = torch.full((S,), 2.75)
    lo_t = torch.full((S,), -2.5)
    rows = torch.zeros(max_len, n)
    rows[:] = util.centroids(bits) * ((2.75 + 2.5) / 2) + (2.75 - 2.5) / 2
    name = "vecquant3appendvecVsparseParallel"
    mo = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    getattr(orc, name)(mo, rows, v, lo_t, hi_t)
    mr, mh = torch.zeros_like(mo).cuda(), torch.zeros_like(mo).cuda()
    getattr(ref, name)(mr, rows.cuda(), v.cuda(), lo_t.cuda(), hi_t.cuda())
    getattr(qc, name)(mh, rows.cuda(), v.cuda(), lo_t.cuda(), hi_t.cuda())
    assert torch.equal(mr, mh) and torch.equal(mr.cpu(), mo)


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
    vals[:L][torch.rand(L, n_out, generator=g) < 0.3] = 0.0
    return vals, idx


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("L,q_len,pos_offset,sparse", [
    (1, 1, 0, True), (77, 1, 5, True), (700, 2, 0, False), (4097, 1, 0, True), (4097, 1, 123456, True),
    (300, 1, 1000000, True), (20000, 1, 0, True)])
def test_score_k(ref, qc, orc, bits, L, q_len, pos_offset, sparse):
    max_len = L + 3
    lut, _, _, _, _ = util.k_tables(bits, seed=bits)
    mat = _random_cache(bits, L, max_len, 7 + L)
    g = torch.Generator().manual_seed(L)
    q = torch.randn(q_len, H, HD, generator=g).half().float()
    vals, idx = _outliers(L, max_len, L)
    sfx = "opt2" if sparse else "opt"
    name = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_%s" % (bits, sfx)
    o = torch.zeros(q_len, H, L)
    r = torch.zeros(q_len, H, L).cuda()
    h = torch.zeros(q_len, H, L).cuda()
    if sparse:
        if L <= 5000:
            getattr(orc, name)(q, mat, o, lut, L, vals, idx, 10000.0, pos_offset)
        getattr(ref, name)(q.cuda(), mat.cuda(), r, lut.cuda(), L, vals.cuda(), idx.cuda(), 10000.0, pos_offset)
        getattr(qc, name)(q.cuda(), mat.cuda(), h, lut.cuda(), L, vals.cuda(), idx.cuda(), 10000.0, pos_offset)
    else:
        getattr(orc, name)(q, mat, o, lut, L, 10000.0, pos_offset)
        getattr(ref, name)(q.cuda(), mat.cuda(), r, lut.cuda(), L, 10000.0, pos_offset)
        getattr(qc, name)(q.cuda(), mat.cuda(), h, lut.cuda(), L, 10000.0, pos_offset)
    torch.cuda.synchronize()
    e_h = util.rel_err(h.cpu(), r.cpu())
    print("score_k bits=%d L=%d pos_offset=%d: |HIP - reference| = %.2e" % (bits, L, pos_offset, e_h), end="")
    assert e_h < TOL, e_h
    if L <= 5000:
        e_o = util.rel_err(o, r.cpu())
        print(", |oracle - reference| = %.2e" % e_o)
        assert e_o < TOL, e_o


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_score_k_opt2_shadow_mirror_decode_pattern(ref, qc, bits):
    """`..._rope_mha_batched_fused_opt2` through kvquant_amd.quant_cuda in the reference glue's decode pattern (ML:748-749:
    one in-place row write per array and append, then the score call at a length one larger): the shadow mirror is
    transposed in full on the first call, by one row on an append, in full again whenever older rows may have changed
    (a rewritten row, a shorter cache, an unchanged length) -- and every result equals the reference's own kernel."""
    name = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits
    dev = torch.device("cuda:0")
    L0, max_len = 17000, 20032
    assert L0 >= qc.QC_MIRROR_FROM and qc.QC_MIRROR
    lut = util.k_tables(bits, seed=bits)[0].to(dev)
    mat = _random_cache(bits, max_len - 3, max_len, 99).to(dev)
    vals, idx = [x.to(dev) for x in _outliers(max_len - 3, max_len, 5)]
    g = torch.Generator().manual_seed(3)
    q = torch.randn(1, H, HD, generator=g).half().float().to(dev)

    def both(L):
        r, h = torch.zeros(1, H, L, device=dev), torch.zeros(1, H, L, device=dev)
        getattr(ref, name)(q, mat, r, lut, L, vals, idx, 10000.0, 7)
        before = dict(qc.shadow_stats)
        getattr(qc, name)(q, mat, h, lut, L, vals, idx, 10000.0, 7)
        e = util.rel_err(h.cpu(), r.cpu())
        assert e < TOL, (L, e)
        return {k: qc.shadow_stats[k] - before[k] for k in before}, e

    def glue_append(L):    # what ML:748-749 does for token L
        gg = torch.Generator().manual_seed(1000 + L)
        vals[L] = (torch.randn(42, generator=gg) * 3).to(dev)
        idx[L] = torch.sort(torch.randperm(C, generator=gg)[:42]).values.to(torch.int32).to(dev)

    qc.shadow_invalidate()
    assert both(L0)[0] == {"full": 1, "incremental": 0}
    errs = []
    for L in range(L0, L0 + 5):
        glue_append(L)
        d, e = both(L + 1)
        errs.append(e)
        assert d == {"full": 0, "incremental": 1}, d
    L = L0 + 5
    # an older row rewritten next to the append: the counters moved by two, everything is transposed again
    vals[17] = vals[17] * -2.0
    glue_append(L)
    assert both(L + 1)[0] == {"full": 1, "incremental": 0}
    # the same length again, a shorter cache ("reset"), a jump of many rows written by ONE in-place operation (prefill)
    assert both(L + 1)[0]["full"] == 1
    assert both(16500)[0]["full"] == 1
    vals[16500:16600] = vals[16500:16600] + 1.0
    idx[16500:16600] = idx[16500:16600].clone()
    assert both(16600)[0]["full"] == 1
    # a writer torch's counters do not see (another handle on the same memory) + an ordinary append: the documented
    # contract is shadow_invalidate()
    alias = torch.from_dlpack(torch.utils.dlpack.to_dlpack(vals))
    alias[33] = alias[33] * 0.5 + 1.0
    glue_append(16600)
    qc.shadow_invalidate(vals)
    assert both(16601)[0] == {"full": 1, "incremental": 0}
    # q_len = 2 and the switch: the row kernel
    q2 = torch.cat([q, q * 0.5])
    r, h = torch.zeros(2, H, 16601, device=dev), torch.zeros(2, H, 16601, device=dev)
    before = dict(qc.shadow_stats)
    getattr(ref, name)(q2, mat, r, lut, 16601, vals, idx, 10000.0, 7)
    getattr(qc, name)(q2, mat, h, lut, 16601, vals, idx, 10000.0, 7)
    assert qc.shadow_stats == before and util.rel_err(h.cpu(), r.cpu()) < TOL
    qc.QC_MIRROR = False
    try:
        assert both(16601)[0] == {"full": 0, "incremental": 0}
    finally:
        qc.QC_MIRROR = True
    print("bits=%d: shadow-mirror decode steps |HIP - reference| max %.2e" % (bits, max(errs)))


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("L,q_len,sparse", [(1, 1, True), (63, 1, True), (1000, 2, False), (5001, 1, True),
                                             (20000, 1, True)])
def test_mix_v(ref, qc, orc, bits, L, q_len, sparse):
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
    o = torch.zeros(q_len, H, HD)
    r = torch.zeros(q_len, H, HD).cuda()
    h = torch.zeros(q_len, H, HD).cuda()
    if sparse:
        getattr(orc, name)(p, mat, o, rows, L, vals, idx)
        getattr(ref, name)(p.cuda(), mat.cuda(), r, rows.cuda(), L, vals.cuda(), idx.cuda())
        getattr(qc, name)(p.cuda(), mat.cuda(), h, rows.cuda(), L, vals.cuda(), idx.cuda())
    else:
        getattr(orc, name)(p, mat, o, rows, L)
        getattr(ref, name)(p.cuda(), mat.cuda(), r, rows.cuda(), L)
        getattr(qc, name)(p.cuda(), mat.cuda(), h, rows.cuda(), L)
    torch.cuda.synchronize()
    e_h = util.rel_err(h.cpu().reshape(q_len, -1), r.cpu().reshape(q_len, -1))
    e_o = util.rel_err(o.reshape(q_len, -1), r.cpu().reshape(q_len, -1))
    print("mix_v bits=%d L=%d: |HIP - reference| = %.2e, |oracle - reference| = %.2e" % (bits, L, e_h, e_o))
    assert e_h < TOL and e_o < TOL, (e_h, e_o)


def _cmp_list(a, b, what):
    """outputs of the ...sparseorig ops: [rows, cols, vals, start, num_threads(cpu), outlier_count]"""
    assert len(a) == len(b) == 6
    for i, (x, y) in enumerate(zip(a, b)):
        x, y = x.cpu(), y.cpu()
        assert x.numel() == y.numel(), "%s: element %d has %d vs %d entries" % (what, i, x.numel(), y.numel())
        if x.is_floating_point():
            assert torch.equal(x.float().view(torch.int32), y.float().view(torch.int32)), "%s: element %d" % (what, i)
        else:
            assert torch.equal(x.long(), y.long()), "%s: element %d" % (what, i)


def test_orig_k(ref, qc, orc):
    """uncapped CSR path (KCU:691-931, 5506-5596), token by token from an empty matrix"""
    bits = 4
    lut, lo, hi, scale, shift = util.k_tables(bits, seed=31)
    zp = ((hi.half() + lo.half()) / 2).float()
    xs = util.k_tokens(7, scale, shift, seed=9)
    xs[3] = xs[3].clamp(lo + 1e-3, hi - 1e-3).half().float().clamp(lo, hi)   # a token without outliers
    max_len = 12
    mo = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    mr, mh = mo.clone().cuda(), mo.clone().cuda()
    e = torch.tensor([])
    so = [e, e, e, e]
    sr = [e.cuda()] * 4
    sh = [e.cuda()] * 4
    name = "vecquant4appendvecKsparseorig"
    for t in range(7):
        x = xs[t].contiguous()
        oo = getattr(orc, name)(mo, lut, x, zp, so[0], so[1], so[2], so[3], lo, hi, t)
        rr = getattr(ref, name)(mr, lut.cuda(), x.cuda(), zp.cuda(), sr[0], sr[1], sr[2], sr[3], lo.cuda(), hi.cuda(), t)
        hh = getattr(qc, name)(mh, lut.cuda(), x.cuda(), zp.cuda(), sh[0], sh[1], sh[2], sh[3], lo.cuda(), hi.cuda(), t)
        _cmp_list(hh, rr, "HIP vs reference, token %d" % t)
        _cmp_list(oo, rr, "oracle vs reference, token %d" % t)
        so, sr, sh = oo[:4], rr[:4], hh[:4]
        nt = int(rr[4][0])
    assert torch.equal(mr, mh) and torch.equal(mr.cpu(), mo)
    L = 7
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, H, HD, generator=g).half().float()
    name = "vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig"
    o = torch.zeros(1, H, L)
    r, h = o.clone().cuda(), o.clone().cuda()
    nnz = sr[2].numel()
    getattr(orc, name)(q, mo, o, lut, L, so[0], so[1], so[3], so[2], L, nt, nnz, 10000.0, 2)
    getattr(ref, name)(q.cuda(), mr, r, lut.cuda(), L, sr[0], sr[1], sr[3], sr[2], L, nt, nnz, 10000.0, 2)
    getattr(qc, name)(q.cuda(), mh, h, lut.cuda(), L, sh[0], sh[1], sh[3], sh[2], L, nt, nnz, 10000.0, 2)
    assert util.rel_err(h.cpu(), r.cpu()) < TOL and util.rel_err(o, r.cpu()) < TOL


def test_orig_v(ref, qc, orc):
    """uncapped CSC path (KCU:933-1163, 5599-5668)"""
    bits, n = 4, 16
    xs = util.v_tokens(6, seed=41)
    max_len = 10
    rows, los, his = _v_rows(bits, xs, max_len)
    mo = torch.zeros(H, W(bits), max_len, dtype=torch.int32)
    mr, mh = mo.clone().cuda(), mo.clone().cuda()
    e = torch.tensor([])
    so = [e, e, e, e]
    sr = [e.cuda()] * 4
    sh = [e.cuda()] * 4
    name = "vecquant4appendvecVsparseorig"
    for t in range(6):
        x = xs[t].contiguous()
        zp = float(rows[t, 7])
        oo = getattr(orc, name)(mo, rows, x, zp, so[0], so[1], so[2], so[3], los[t], his[t], t)
        rr = getattr(ref, name)(mr, rows.cuda(), x.cuda(), zp, sr[0], sr[1], sr[2], sr[3], los[t], his[t], t)
        hh = getattr(qc, name)(mh, rows.cuda(), x.cuda(), zp, sh[0], sh[1], sh[2], sh[3], los[t], his[t], t)
        _cmp_list(hh, rr, "HIP vs reference, token %d" % t)
        _cmp_list(oo, rr, "oracle vs reference, token %d" % t)
        so, sr, sh = oo[:4], rr[:4], hh[:4]
        nt = int(rr[4][0])
    assert torch.equal(mr, mh) and torch.equal(mr.cpu(), mo)
    L = 6
    g = torch.Generator().manual_seed(6)
    p = torch.softmax(torch.randn(1, H, L, generator=g), dim=-1).half().float().contiguous()
    name = "vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig"
    o = torch.zeros(1, H, HD)
    r, h = o.clone().cuda(), o.clone().cuda()
    nnz = sr[2].numel()
    getattr(orc, name)(p, mo, o, rows, L, so[0], so[1], so[3], so[2], L, nt, nnz)
    getattr(ref, name)(p.cuda(), mr, r, rows.cuda(), L, sr[0], sr[1], sr[3], sr[2], L, nt, nnz)
    getattr(qc, name)(p.cuda(), mh, h, rows.cuda(), L, sh[0], sh[1], sh[3], sh[2], L, nt, nnz)
    assert util.rel_err(h.cpu().reshape(1, -1), r.cpu().reshape(1, -1)) < TOL
    assert util.rel_err(o.reshape(1, -1), r.cpu().reshape(1, -1)) < TOL


def test_rope_frequency_powf(ref):
    """theta_j = powf(rope_theta, -2j/128) is evaluated ON THE DEVICE by the reference (KCU:3081); libkvq does
    the same (kvq_rope_freqs returns the table its kernels use) and the oracle is handed that table on a GPU
    box (tests/conftest.py).  Quantify what the alternative -- the correctly rounded value -- would cost."""
    from kvquant_amd import ops
    j = torch.arange(64, dtype=torch.float32)
    e = -2.0 * j / 128.0
    ours = ops.rope_freqs(10000.0).cpu()
    dev = torch.pow(torch.tensor(10000.0, device="cuda"), e.cuda()).cpu()
    exact = torch.pow(torch.tensor(10000.0, dtype=torch.float64), e.double()).float()
    ulp = (ours.view(torch.int32) - exact.view(torch.int32)).abs()
    print("device powf vs correctly rounded: %d of 64 frequencies differ, max %d ulp; vs torch.pow on the GPU: %d differ"
          % (int((ulp > 0).sum()), int(ulp.max()), int((ours != dev).sum())))
    assert int(ulp.max()) <= 2

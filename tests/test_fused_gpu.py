"""GPU parity of the GPU-resident decode path (fused append kernels + fused
softmax) against the restated reference glue (oracle.glue, itself pinned to the
reference's classes by tests/golden)."""
import math

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C


def _quantizer(bits, seed=0):
    g = torch.Generator().manual_seed(seed)
    scale = torch.exp(0.5 * torch.randn(C, generator=g))
    shift = 0.3 * torch.randn(C, generator=g)
    upper = (shift + 2.5 * scale).numpy()[None, :]
    lower = (shift - 2.5 * scale).numpy()[None, :]
    cent = util.centroids(bits)[torch.randperm(2 ** bits, generator=g)].numpy().reshape(-1, 1)
    return (upper, lower, [cent]), scale, shift


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("sinks", [0, 5])
def test_fused_decode_matches_oracle(gpu, bits, sinks):
    from kvquant_amd.cache import QuantK, QuantV
    from oracle.glue import OracleQuantK, OracleQuantV
    quant, scale, shift = _quantizer(bits, seed=bits)
    steps, max_len = 6, 16
    ks = util.k_tokens(steps, scale, shift, seed=10 + bits)
    vs = util.v_tokens_no_ties(steps, seed=20 + bits)
    g = torch.Generator().manual_seed(bits)
    qs = torch.randn(steps, H, 1, HD, generator=g).half()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=sinks)
    ok, ov = OracleQuantK(rope_theta=10000.0, **kw), OracleQuantV(**kw)
    gk, gv = QuantK(rope_theta=10000.0, device=gpu, **kw), QuantV(device=gpu, **kw)
    for c in (ok, ov, gk, gv):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    for c in (ok, gk):
        c.klen += sinks
    for c in (ov, gv):
        c.vlen += sinks
    for i in range(steps):
        k16, v16, q16 = ks[i].half(), vs[i].half(), qs[i]
        s_ref = ok.forward_fused_sparse(q16, k16)
        s_gpu = gk.forward_fused_sparse(q16.to(gpu), k16.to(gpu))
        assert util.rel_err(s_gpu.float().cpu(), s_ref.float()) < 2e-3
        p = torch.softmax(s_ref.float() / math.sqrt(HD), dim=-1).half()
        uv, ui, lv, li = OracleQuantV.topk_inputs(v16.float().unsqueeze(0), 0.99, C)
        o_ref = ov.forward_fused_sparse(p, v16, uv[0], ui[0], lv[0], li[0])
        o_gpu = gv.forward_fused_sparse(p.to(gpu), v16.to(gpu))          # selection inside the fused kernel
        assert util.rel_err(o_gpu.float().cpu().reshape(1, -1), o_ref.float().reshape(1, -1)) < 2e-3
    L = steps
    assert torch.equal(ok.kcache[:, :, :L], gk.kcache[:, :, :L].cpu())
    assert torch.equal(ov.vcache[:, :, :L], gv.vcache[:, :, :L].cpu())
    assert torch.equal(ok.outlier_indices[:L], gk.outlier_indices[:L].cpu())
    assert torch.equal(ok.outliers[:L].view(torch.int32), gk.outliers[:L].cpu().view(torch.int32))
    assert torch.equal(ov.outlier_indices[:L], gv.outlier_indices[:L].cpu())
    assert torch.equal(ov.outliers[:L].view(torch.int32), gv.outliers[:L].cpu().view(torch.int32))
    assert torch.equal(ov.lookup_table[:L].view(torch.int32), gv.lookup_table[:L].cpu().view(torch.int32))
    # the token-contiguous mirror the decode kernel reads is the transpose of the reference-layout rows
    assert torch.equal(gk.outlier_indices_t[:, :L].t().cpu(), gk.outlier_indices[:L].cpu())
    assert torch.equal(gk.outliers_t[:, :L].t().cpu().view(torch.int32), gk.outliers[:L].cpu().view(torch.int32))


def test_fused_selection_with_ties(gpu):
    """ties at the selection boundary: torch.topk's choice among equal values is unspecified;
    ours is 'lowest channel first'.  The selected VALUES and thresholds must still agree."""
    from kvquant_amd import ops
    bits, n = 4, 16
    x = torch.zeros(C)
    x[100:140] = 3.0          # 40 equal maxima, only 21 can be kept
    x[2000:2050] = -2.0       # 50 equal minima
    x[7] = 5.0
    lut = util.centroids(bits)
    mat = torch.zeros(H, 16, 4, dtype=torch.int32, device=gpu)
    rows = torch.zeros(4, n, device=gpu)
    outl = torch.zeros(4, 42, device=gpu)
    oidx = torch.zeros(4, 42, dtype=torch.int32, device=gpu)
    ops.append_v_fused(bits, mat, rows, lut.to(gpu), x.to(gpu), outl, oidx, 21, 1)
    idx = oidx[1].cpu()
    assert torch.equal(idx, torch.sort(idx).values)
    assert len(set(idx.tolist())) == 42
    expect_hi = [7] + list(range(100, 120))           # 5.0, then the 20 lowest-index 3.0s
    expect_lo = list(range(2000, 2021))
    assert sorted(idx.tolist()) == sorted(expect_hi + expect_lo)
    # thresholds: 22nd largest = 3.0, 22nd smallest = -2.0
    r = rows[1].cpu()
    assert torch.allclose(r, lut * 2.5 + 0.5)


@pytest.mark.parametrize("L,n_sink", [(1, 0), (1000, 0), (40000, 5), (131073, 0)])
def test_softmax_scale(gpu, L, n_sink):
    from kvquant_amd import ops
    g = torch.Generator().manual_seed(L)
    raw = (torch.randn(H, L, generator=g) * 30).to(gpu)
    inv = 1.0 / math.sqrt(HD)
    sink = (torch.randn(H, n_sink, generator=g) * 3).half().to(gpu) if n_sink else None
    probs, sp = ops.softmax_scale(raw, inv, sink)
    s16 = (raw.half() / math.sqrt(HD))                       # fp16 tensor / python scalar, as ML:1972-1973
    full = s16 if sink is None else torch.cat((sink, s16), dim=-1)
    ref = torch.softmax(full, dim=-1, dtype=torch.float32).half()
    got = probs.half() if sink is None else torch.cat((sp, probs.half()), dim=-1)
    assert torch.equal(probs, probs.half().float())          # values are fp16-representable
    d = (got.float() - ref.float()).abs()
    tol = ref.float().abs() * 2e-3 + 1e-7                    # 1-2 fp16 ulp
    assert bool((d <= tol).all()), float((d - tol).max())
    assert abs(float(got.float().sum(-1).mean()) - 1.0) < 5e-3


@pytest.mark.parametrize("bits,L,n_sink", [(4, 300, 0), (4, 20000, 3), (3, 16500, 0), (2, 1000, 2)])
def test_score_softmax_fused_matches_two_pass(gpu, bits, L, n_sink):
    """the score kernel with the first softmax pass fused in (+ kvq_softmax_finish) against the plain
    score kernel + kvq_softmax_scale, and the raw scores against the C oracle"""
    from kvquant_amd import ops
    from oracle import ckernels as ck
    n = 2 ** bits
    max_len = (L + 64) // 64 * 64
    g = torch.Generator().manual_seed(bits * 1000 + L)
    mat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, HD // 32 * bits, max_len), dtype=torch.int64, generator=g).to(torch.int32)
    lut = torch.randn(H, HD, n, generator=g).sort(dim=-1).values.contiguous()
    q = torch.randn(1, H, HD, generator=g)
    vals = torch.randn(max_len, 42, generator=g)
    idx = torch.sort(torch.randint(0, C, (max_len, 42), generator=g, dtype=torch.int64), dim=-1).values.to(torch.int32)
    inv = 1.0 / math.sqrt(HD)
    sink = (torch.randn(H, n_sink, generator=g) * 3).half().to(gpu) if n_sink else None
    mg, lg, vg, ig = mat.to(gpu), lut.to(gpu), vals.to(gpu), idx.to(gpu)
    # two-pass reference on the GPU
    s2 = torch.zeros(1, H, L, device=gpu)
    ops.score_k(bits, q.to(gpu), mg, s2, lg, L, 10000.0, 0, vg, ig, accumulate=False)
    p2, sp2 = ops.softmax_scale(s2[0], inv, sink)
    # fused: tables through the prologue-free prep inside score_k leave them in the "score" workspace
    ws = ops._workspace(mg.device, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
    s1 = torch.zeros(1, H, L, device=gpu)
    p1, sp1 = ops.score_k_softmax(bits, mg, s1, lg, L, 10000.0, 0, ws, vg, ig, inv, sink)
    assert torch.equal(s1, s2)
    d = (p1 - p2).abs()
    assert bool((d <= p2.abs() * 2e-3 + 1e-7).all()), float(d.max())
    if n_sink:
        assert bool(((sp1.float() - sp2.float()).abs() <= sp2.float().abs() * 2e-3 + 1e-7).all())
    assert abs(float(p1.sum(-1).mean()) + (float(sp1.float().sum(-1).mean()) if n_sink else 0.0) - 1.0) < 5e-3
    # raw scores against the oracle
    ref = torch.zeros(1, H, L)
    ck.score_k(bits, q, mat, ref, lut, L, 10000.0, 0)
    ck.spmv_k_rope(vals, idx, q, ref, L, 10000.0, 0)
    assert util.rel_err(s1.cpu().reshape(1, -1), ref.reshape(1, -1)) < 2e-5
    # the same through the token-contiguous outlier mirror (lane-owns-token variant of the kernel)
    vt, it = vg.t().contiguous(), ig.t().contiguous()
    s3 = torch.zeros(1, H, L, device=gpu)
    p3, sp3 = ops.score_k_softmax(bits, mg, s3, lg, L, 10000.0, 0, ws, vg, ig, inv, sink, vt, it)
    assert util.rel_err(s3.cpu().reshape(1, -1), ref.reshape(1, -1)) < 2e-5
    # its raw scores are summed in another order than s2 (equal to ~1e-6), and one fp16 ulp of a scaled score
    # is up to 1.6 % of its probability (the reference's own half() pipeline, ML:873-874): the probabilities are
    # therefore compared with the two-pass softmax of THIS variant's scores
    p4, sp4 = ops.softmax_scale(s3[0], inv, sink)
    d = (p3 - p4).abs()
    assert bool((d <= p4.abs() * 2e-3 + 1e-7).all()), float(d.max())
    if n_sink:
        assert bool(((sp3.float() - sp4.float()).abs() <= sp4.float().abs() * 2e-3 + 1e-7).all())


@pytest.mark.parametrize("bits,L,max_len,n_sink", [(4, 5000, 5056, 0), (4, 777, 1024, 5), (4, 31, 64, 0), (4, 1, 64, 3),
                                                    (3, 3000, 3072, 0), (3, 100, 128, 5), (2, 2100, 2112, 0),
                                                    (4, 3000, 3003, 0), (4, 40000, 40064, 0)])
def test_mix_v_softmax_matches_two_passes(bits, L, max_len, n_sink):
    """kvq_mix_v_softmax (the second softmax pass inside the p.V kernel) against kvq_softmax_finish + kvq_mix_v on the
    same raw scores and partials: same arithmetic per element, the normaliser merged in another order -- differences
    are last-bit effects on fp16-rounded probabilities.  Covers sinks, ragged ranges, L smaller than a chunk, the
    shape that takes the two-pass route inside the library (max_len % 4 != 0) and every bit width."""
    import math
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    H, HD, C = 32, 128, 4096
    n = 2 ** bits
    W = HD // 32 * bits
    g = torch.Generator(device="cuda").manual_seed(L + bits)
    kmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
    vmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
    rows = (torch.randn(max_len, n, device=dev, generator=g) * 0.5).sort(dim=-1).values.contiguous()
    lut = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    vals = {}
    for nm in ("k", "v"):
        vals[nm] = torch.randn(max_len, 42, device=dev, generator=g) * 0.3
        vals[nm + "i"] = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
    q = torch.randn(1, H, HD, device=dev, generator=g) * 3
    inv = 1.0 / math.sqrt(HD)
    sink = (torch.randn(H, n_sink, device=dev, generator=g) * 2).half() if n_sink else None
    s = torch.zeros(1, H, L, device=dev)
    ops.score_k(bits, q, kmat, torch.zeros(1, H, 1, device=dev), lut, 1, 10000.0, 0, accumulate=False)   # tables -> workspace
    ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    assert n_parts > 0
    parts = ops.score_k_prepared_softmax(bits, kmat, s, lut, L, 10000.0, 0, ws, vals["k"], vals["ki"], inv, n_parts)
    probs, sp_ref = ops.softmax_finish(s[0], parts, n_parts, inv, sink)
    ref = torch.zeros(1, H, HD, device=dev)
    ops.mix_v(bits, probs.unsqueeze(0), vmat, ref, rows, L, vals["v"], vals["vi"], accumulate=False)
    out = torch.full((1, H, HD), 7.0, device=dev)
    sp = ops.score_k_mix_v(bits, kmat, s, lut, L, 10000.0, 0, ws, vals["k"], vals["ki"], inv, vmat, out, rows,
                           vals["v"], vals["vi"], sink)
    torch.cuda.synchronize()
    scale = ref.abs().max().item() + 1e-6
    assert (out - ref).abs().max().item() <= 2e-3 * scale
    if n_sink:
        assert (sp.float() - sp_ref.float()).abs().max().item() <= 2e-3 * (sp_ref.float().abs().max().item() + 1e-6)

"""BASELINE-size (128K cached tokens, ragged) checks of the two decode matvecs through size-independent
properties: windows of the big result against the oracle run on the same window of the cache (the oracle
finishes a few hundred tokens in well under a second), linearity in the query / the probabilities, and
agreement of the kernel variants (reference row layout vs token-contiguous outlier mirror)."""
import math

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C
L_FULL = 131072 + 77          # ragged last 256-token tile, as on every decode step
WINDOWS = [(0, 300), (65536 - 100, 333), (L_FULL - 290, 290)]


@pytest.fixture(scope="module", params=[4, 3, 2])
def big(request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    bits = request.param
    n = 2 ** bits
    max_len = (L_FULL + 64) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(2024)
    W = HD // 32 * bits
    d = dict(bits=bits, dev=dev, max_len=max_len)
    for name in ("kmat", "vmat"):
        d[name] = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
    d["klut"] = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    d["vrows"] = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    for name in ("k", "v"):
        d[name + "vals"] = torch.randn(max_len, 42, device=dev, generator=g) * (torch.rand(max_len, 42, device=dev, generator=g) > 0.2)
        d[name + "idx"] = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
    d["q"] = torch.randn(1, H, HD, device=dev, generator=g)
    return d


def _scores(d, q, mirror):
    from kvquant_amd import ops
    s = torch.zeros(1, H, L_FULL, device=d["dev"])
    if not mirror:
        ops.score_k(d["bits"], q, d["kmat"], s, d["klut"], L_FULL, 10000.0, 3, d["kvals"], d["kidx"], accumulate=False)
        return s
    ops.score_k(d["bits"], q, d["kmat"], torch.zeros(1, H, 1, device=d["dev"]), d["klut"], 1, 10000.0, 3,
                accumulate=False)                                    # (leaves the tables of q in the workspace)
    ws = ops._workspace(d["dev"], ops._L().kvq_score_k_workspace_bytes(d["bits"], 1, H), slot="score")
    n_parts = ops._L().kvq_score_k_softmax_parts(d["bits"], L_FULL, 1)
    s = torch.zeros(1, H, L_FULL, device=d["dev"])
    parts = ops.score_k_prepared_softmax(d["bits"], d["kmat"], s, d["klut"], L_FULL, 10000.0, 3, ws, d["kvals"],
                                         d["kidx"], 1.0 / math.sqrt(HD), n_parts,
                                         d["kvals"].t().contiguous(), d["kidx"].t().contiguous())
    return s, parts, n_parts


def test_score_k_full_size_windows_linearity_and_variants(big):
    from oracle import ckernels as ck
    d = big
    s_rows = _scores(d, d["q"], mirror=False)
    s_mir, parts, n_parts = _scores(d, d["q"], mirror=True)
    assert util.rel_err(s_mir.cpu().reshape(H, -1), s_rows.cpu().reshape(H, -1)) < 2e-5
    for w0, n in WINDOWS:       # the oracle on a window of the cache, positions offset accordingly
        mat = d["kmat"][:, :, w0:w0 + n].contiguous().cpu()
        ref = torch.zeros(1, H, n)
        ck.score_k(d["bits"], d["q"].cpu(), mat, ref, d["klut"].cpu(), n, 10000.0, 3 + w0)
        ck.spmv_k_rope(d["kvals"][w0:w0 + n].contiguous().cpu(), d["kidx"][w0:w0 + n].contiguous().cpu(), d["q"].cpu(),
                       ref, n, 10000.0, 3 + w0)
        for s in (s_rows, s_mir):
            assert util.rel_err(s[0, :, w0:w0 + n].cpu().reshape(1, -1), ref.reshape(1, -1)) < 2e-5, (w0, n)
    # linear in the query
    g = torch.Generator(device=d["dev"]).manual_seed(5)
    q2 = torch.randn(1, H, HD, device=d["dev"], generator=g)
    s2 = _scores(d, q2, mirror=False)
    s12 = _scores(d, d["q"] + q2, mirror=False)
    assert util.rel_err(s12.cpu().reshape(H, -1), (s_rows + s2).cpu().reshape(H, -1)) < 1e-5
    # the fused first softmax pass: partials reproduce max and sum of exp of the scaled scores per row
    from kvquant_amd import ops
    probs, _ = ops.softmax_finish(s_mir[0], parts, n_parts, 1.0 / math.sqrt(HD))
    ref_p = torch.softmax(s_mir[0].half() / math.sqrt(HD), dim=-1, dtype=torch.float32).half().float()
    assert bool(((probs - ref_p).abs() <= ref_p.abs() * 2e-3 + 1e-7).all())
    assert abs(float(probs.sum(-1).mean()) - 1.0) < 5e-3


def test_mix_v_full_size_windows_and_linearity(big):
    from kvquant_amd import ops
    from oracle import ckernels as ck
    d = big
    dev = d["dev"]
    g = torch.Generator(device=dev).manual_seed(11)

    def mix(p):
        out = torch.zeros(1, H, HD, device=dev)
        ops.mix_v(d["bits"], p, d["vmat"], out, d["vrows"], L_FULL, d["vvals"], d["vidx"], accumulate=False)
        return out

    for w0, n in WINDOWS:       # probabilities supported on a window: the result is the oracle's on that window
        p = torch.zeros(1, H, L_FULL, device=dev)
        pw = torch.rand(1, H, n, device=dev, generator=g).half().float()
        p[:, :, w0:w0 + n] = pw
        ref = torch.zeros(1, H, HD)
        ck.mix_v(d["bits"], pw.cpu(), d["vmat"][:, :, w0:w0 + n].contiguous().cpu(), ref, d["vrows"][w0:w0 + n].contiguous().cpu(), n)
        ck.spmv_v(d["vvals"][w0:w0 + n].contiguous().cpu(), d["vidx"][w0:w0 + n].contiguous().cpu(), pw.cpu(), ref, n)
        assert util.rel_err(mix(p).cpu().reshape(1, -1), ref.reshape(1, -1)) < 2e-5, (w0, n)
    # linear in p, and every token counted exactly once: uniform p over the whole range vs two halves
    p1 = torch.softmax(torch.randn(1, H, L_FULL, device=dev, generator=g), dim=-1).half().float()
    p2 = torch.softmax(torch.randn(1, H, L_FULL, device=dev, generator=g), dim=-1).half().float()
    # (131K random-sign terms cancel to a result ~100x smaller than the summands: fp32 rounding alone is ~1e-5
    # of the result; an indexing error would be O(1))
    assert util.rel_err(mix(p1 + p2).cpu().reshape(1, -1), (mix(p1) + mix(p2)).cpu().reshape(1, -1)) < 3e-4
    pa, pb = p1.clone(), p1.clone()
    pa[:, :, 70000:] = 0
    pb[:, :, :70000] = 0
    assert util.rel_err((mix(pa) + mix(pb)).cpu().reshape(1, -1), mix(p1).cpu().reshape(1, -1)) < 3e-4

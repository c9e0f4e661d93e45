"""Tokens with a TIE at the outlier-selection boundary (21st == 22nd largest or smallest fp16 value: one token in
ten) through the GPU-resident V append and the one-call decode step, next to the reference's own kernel + glue.

What is and is not defined there: the reference's kernel clips STRICTLY outside the per-token thresholds (KCU:2084),
so its packed codes do not depend on which of two equal values torch.topk happened to select -- they are compared
bit for bit.  Which CHANNEL among equal values lands in the outlier row is torch.topk's unspecified tie order (the
GPU selection here takes the lowest channel): rows are compared as sets of (value) with the differing indices
required to be ties.  With `reference_tie_quirk = False` (the simulated path's semantics) exactly one code per tie
and side differs, and exactly where predicted."""
import math

import pytest
import torch

from tests import decode_check, util

pytestmark = pytest.mark.gpu

H, HD, C = util.H, util.HD, util.C
K = 21


@pytest.fixture(scope="module")
def ref():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import build_ref
    if build_ref.built() is None:
        pytest.skip("oracle/_ref not built")
    util.sync_oracle_freqs(10000.0)
    return build_ref.load()


def tie_tokens(n, seed):
    """fp16 tokens that ALL have a tie at the upper or the lower selection boundary (alternating), some at both"""
    x = util.v_tokens_no_ties(n, seed=seed).clone()
    kinds = []
    for t in range(n):
        kind = t % 3            # 0: upper tie, 1: lower tie, 2: both
        if kind in (0, 2):
            top = torch.topk(x[t], K + 2)
            x[t, top.indices[K - 1]] = x[t, top.indices[K]]
        if kind in (1, 2):
            bot = torch.topk(x[t], K + 2, largest=False)
            x[t, bot.indices[K - 1]] = x[t, bot.indices[K]]
        kinds.append(kind)
    return x.half().float(), kinds


def ref_append(ref, bits, vcache, rows, x, pos):
    """the reference's decode glue for one token (ML:1086-1176) around ITS kernel: thresholds from torch.topk(22),
    per-token codebook row, vecquant{b}appendvecVsparse, outlier row sorted by channel"""
    from kvquant_amd.cache import ZERO_CODE
    uv, ui = torch.topk(x, K + 1)
    lv, li = torch.topk(x, K + 1, largest=False)
    maxval, minval = uv[-1], lv[-1]
    off, sf = (maxval + minval) / 2, (maxval - minval) / 2
    return uv, ui, lv, li, float(minval), float(maxval), off, sf


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_fused_append_on_tie_tokens_matches_reference_kernel(ref, bits):
    from kvquant_amd.cache import QuantV, ZERO_CODE
    from oracle import ckernels as ck
    dev = torch.device("cuda:0")
    n = 12
    quant, _, _ = decode_check.quantizer(bits, seed=bits)
    xs, kinds = tie_tokens(n, seed=90 + bits)
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=16, include_sparse=True,
              sparsity_threshold=0.99, device=dev)
    vc = QuantV(**kw)                       # default: reference_tie_quirk = True
    assert vc.reference_tie_quirk is True
    vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    sim = QuantV(**kw)
    sim.reference_tie_quirk = False
    sim.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    lut = vc.lut.float()
    rcache = torch.zeros_like(vc.vcache)
    rrows = torch.zeros_like(vc.lookup_table)
    p = torch.ones(H, 1, 1, device=dev)
    n_diff_expected = 0
    for t in range(n):
        x = xs[t].to(dev)
        vc.forward_fused_sparse(p[:, :, :1].expand(H, 1, t + 1).contiguous(), x.half())
        sim.forward_fused_sparse(p[:, :, :1].expand(H, 1, t + 1).contiguous(), x.half())
        uv, ui, lv, li, mn, mx, off, sf = ref_append(ref, bits, rcache, rrows, x, t)
        rrows[t] = lut * sf + off
        getattr(ref, "vecquant%dappendvecVsparse" % bits)(rcache, rrows, x.contiguous(), float(rrows[t, ZERO_CODE[bits]]), mn, mx, t)
        n_diff_expected += 2 if kinds[t] == 2 else 1
        # the outlier row: same values; indices may differ only where values tie
        want_v = torch.cat((uv[:-1], lv[:-1])) - rrows[t, ZERO_CODE[bits]]
        want_i = torch.cat((ui[:-1], li[:-1]))
        got_v, got_i = vc.outliers[t], vc.outlier_indices[t].long()
        assert torch.equal(torch.sort(want_v).values.view(torch.int32), torch.sort(got_v).values.view(torch.int32)), t
        only_ref = set(want_i.tolist()) - set(got_i.tolist())
        only_hip = set(got_i.tolist()) - set(want_i.tolist())
        assert len(only_ref) == len(only_hip) <= 2
        assert sorted(float(x[i]) for i in only_ref) == sorted(float(x[i]) for i in only_hip), "row %d differs beyond a tie" % t
    # codes: bit for bit the reference kernel's on every token, ties included; codebook rows too
    assert torch.equal(vc.vcache[:, :, :n], rcache[:, :, :n])
    assert torch.equal(vc.lookup_table[:n].view(torch.int32), rrows[:n].view(torch.int32))
    # simulated-path semantics: exactly one code per tie and side differs, it is the selected tied channel, and it
    # becomes the zero-point code
    cq = ck.unpack_codes(bits, vc.vcache.cpu(), C, n)
    cs = ck.unpack_codes(bits, sim.vcache.cpu(), C, n)
    diff = (cq != cs)
    assert int(diff.sum()) == n_diff_expected
    for t in range(n):
        ch = torch.nonzero(diff[t]).flatten().tolist()
        sel = set(sim.outlier_indices[t].tolist())
        for c in ch:
            assert c in sel and int(cs[t, c]) == ZERO_CODE[bits] and int(cq[t, c]) != ZERO_CODE[bits]
            assert float(xs[t, c]) in (float(torch.topk(xs[t], K + 1).values[-1]), float(torch.topk(xs[t], K + 1, largest=False).values[-1]))


@pytest.mark.parametrize("bits", [4, 3])
def test_decode_kv_on_tie_tokens(ref, bits):
    """the one-call decode step on tie tokens against the pipeline assembled from the REFERENCE's ops (its score
    kernel, torch softmax in its dtype order, its V append + p.V kernels): attention outputs within 1e-3, packed V
    codes bit for bit.  The reference's OUTPUT on a tie token depends on which of the two equal values torch.topk
    happened to put into the outlier row (that channel is dequantised twice, the other once -- KCU:2084 vs
    ML:1093-1096), so its p.V kernel is run on the outlier rows of the cache under test, whose agreement with the
    reference's rows up to that choice is what the first test establishes."""
    from kvquant_amd.cache import QuantK, QuantV, ZERO_CODE, decode_kv
    dev = torch.device("cuda:0")
    steps, max_len = 6, 64
    quant, scale, shift = decode_check.quantizer(bits, seed=bits)
    ks = util.k_tokens(steps, scale, shift, seed=33 + bits)
    vs, _ = tie_tokens(steps, seed=60 + bits)
    g = torch.Generator().manual_seed(5 + bits)
    qs = torch.randn(steps, H, HD, generator=g).half()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99)
    kc, vc = QuantK(rope_theta=10000.0, device=dev, **kw), QuantV(device=dev, **kw)
    kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    rv = torch.zeros_like(vc.vcache)
    rrows = torch.zeros_like(vc.lookup_table)
    lut = vc.lut.float()
    zc = ZERO_CODE[bits]
    for t in range(steps):
        q, k, v = qs[t].to(dev), ks[t].half().to(dev), vs[t].half().to(dev)
        out, _ = decode_kv(kc, vc, q, k, v)
        L = t + 1
        # reference pipeline on the SAME key cache (the K side is compared against the reference elsewhere)
        s = torch.zeros(1, H, L, device=dev)
        getattr(ref, "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits)(
            q.float().view(1, H, HD).contiguous(), kc.kcache, s, kc.lookup_table, L, kc.outliers, kc.outlier_indices, 10000.0, 0)
        probs = torch.softmax((s[0].half() / math.sqrt(HD)), dim=-1, dtype=torch.float32).half().float()
        x = vs[t].to(dev)
        uv, ui, lv, li, mn, mx, off, sf = ref_append(ref, bits, rv, rrows, x, t)
        rrows[t] = lut * sf + off
        getattr(ref, "vecquant%dappendvecVsparse" % bits)(rv, rrows, x.contiguous(), float(rrows[t, zc]), mn, mx, t)
        o = torch.zeros(1, H, HD, device=dev)
        getattr(ref, "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt2" % bits)(
            probs.view(1, H, L).contiguous(), rv, o, rrows, L, vc.outliers, vc.outlier_indices)
        err = util.rel_err(out.float().cpu().reshape(1, -1), o.half().float().cpu().reshape(1, -1))
        assert err < 1e-3, (t, err)
    assert torch.equal(vc.vcache[:, :, :steps], rv[:, :, :steps])

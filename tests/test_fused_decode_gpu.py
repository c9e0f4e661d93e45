"""kvq_fused_attend (kvq_fused_decode.hip: one kernel for q.K^T + tile-local softmax + p.V per 256-token tile, then a
merge) against the separate kernels on identical caches, and against the pipeline assembled from the REFERENCE's own ops
(oracle/_ref) at the BASELINE sizes.  North-star tolerance of the attention output: 1e-3."""
import math

import pytest
import torch

from tests import decode_check, util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C
TOL = 1e-3


def _layer(bits, ctx, dev, seed, sinks=0):
    import bench
    gen = torch.Generator(device=dev).manual_seed(seed)
    max_len = (ctx + 8 + 63) // 64 * 64
    lay = bench.Layer(bits, max_len, gen, dev, sinks)
    if ctx:
        lay.fill(ctx, gen, dev)
    return lay, gen


def _step(lay, q, k, v, fused):
    from kvquant_amd import cache
    old = cache.FUSED_ATTEND
    cache.FUSED_ATTEND = fused
    try:
        if lay.sinks:
            out, sp = cache.decode_kv(lay.k, lay.v, q, k, v, k_sink=lay.k_sink, v_sink=lay.v_sink)
        else:
            out, sp = cache.decode_kv(lay.k, lay.v, q, k, v)
    finally:
        cache.FUSED_ATTEND = old
    return out, sp


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("ctx", [0, 1, 37, 255, 256, 300, 5000, 66000])
def test_fused_matches_separate_kernels(bits, ctx):
    """two identical caches, the same decode tokens: one through the score / p.V kernel pair, one through the fused
    kernel (incl. the ragged last tile split into head groups, a single token, exactly one full tile)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    dev = torch.device("cuda:0")
    a, gen = _layer(bits, ctx, dev, 1000 + bits + ctx)
    b, _ = _layer(bits, ctx, dev, 1000 + bits + ctx)
    assert torch.equal(a.k.kcache, b.k.kcache)
    k, v = bench.synth_tokens(3, a.scale, a.shift, gen, dev)
    for step in range(3):
        q = torch.randn(H, HD, generator=gen, device=dev).half() * (2.0 if step == 1 else 1.0)
        oa, _ = _step(a, q, k[step], v[step], False)
        ob, _ = _step(b, q, k[step], v[step], True)
        err = util.rel_err(ob.float().reshape(1, -1), oa.float().reshape(1, -1))
        assert err < TOL, (step, err)
    assert torch.equal(a.k.kcache, b.k.kcache) and torch.equal(a.v.vcache, b.v.vcache)


def test_fused_is_opt_in_and_matches_the_separate_kernels_at_1m_tokens():
    """1M cached tokens (config 5's shape, one layer).  Rounds 4 - 5 the fused kernel was decode_kv's own choice from 512K
    tokens on; since round 6 the kernel pair is faster at every length (its p.V evaluates the outlier entries inside the
    dense loop: profiles/r06_k_misc.txt, 1M 10.70 -> 9.86 ms/step), so the default takes the pair (mode 1) and the fused
    kernel (mode 3, KVQ_FUSED_ATTEND=1 / KVQ_FUSED_ATTEND_FROM) gives the same output on the same cache"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from kvquant_amd import cache, ops
    dev = torch.device("cuda:0")
    ctx = 1048576
    assert cache.FUSED_ATTEND is None and ctx < cache.FUSED_ATTEND_FROM
    a, gen = _layer(4, ctx, dev, 4242)
    b, _ = _layer(4, ctx, dev, 4242)
    k, v = bench.synth_tokens(2, a.scale, a.shift, gen, dev)
    modes = []
    real = ops.decode_step
    ops.decode_step = lambda layer, col, q_, k_, v_, out, mode, *rest: (modes.append(mode), real(layer, col, q_, k_, v_, out, mode, *rest))[1]
    try:
        for step in range(2):
            q = torch.randn(H, HD, generator=gen, device=dev).half()
            oa, _ = cache.decode_kv(a.k, a.v, q, k[step], v[step])            # default: the kernel pair
            ob, _ = _step(b, q, k[step], v[step], True)                       # fused split-L kernel
            err = util.rel_err(oa.float().reshape(1, -1), ob.float().reshape(1, -1))
            assert err < TOL, (step, err)
    finally:
        ops.decode_step = real
    assert modes == [1, 3, 1, 3], modes


@pytest.mark.parametrize("bits,ctx", [(3, 700), (4, 40000)])
def test_fused_with_sink_tokens(bits, ctx):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    dev = torch.device("cuda:0")
    a, gen = _layer(bits, ctx, dev, 77 + bits, sinks=5)
    b, _ = _layer(bits, ctx, dev, 77 + bits, sinks=5)
    k, v = bench.synth_tokens(2, a.scale, a.shift, gen, dev)
    for step in range(2):
        q = torch.randn(H, HD, generator=gen, device=dev).half()
        oa, pa = _step(a, q, k[step], v[step], False)
        ob, pb = _step(b, q, k[step], v[step], True)
        assert util.rel_err(ob.float().reshape(1, -1), oa.float().reshape(1, -1)) < TOL
        assert (pa.float() - pb.float()).abs().max().item() <= 2e-3 * max(pa.float().abs().max().item(), 1e-6)


@pytest.mark.parametrize("bits,ctx,sinks", [(4, 131072, 0), (3, 131072, 5), (4, 32768, 0)])
def test_fused_against_reference_pipeline_at_size(bits, ctx, sinks):
    """BASELINE configs 2 / 3 through the fused kernel against the reference's own ops (oracle/_ref) + its glue for the
    fp16 sink tokens (ML:1950-1962 concat of the sink scores, 1987-1995 add of their output)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from oracle import build_ref
    if build_ref.built() is None:
        pytest.skip("oracle/_ref not built")
    ref = build_ref.load()
    dev = torch.device("cuda:0")
    util.sync_oracle_freqs(10000.0)
    lay, gen = _layer(bits, ctx, dev, 4321 + bits, sinks=sinks)
    k, v = bench.synth_tokens(2, lay.scale, lay.shift, gen, dev)
    worst = 0.0
    for step in range(2):
        q = torch.randn(H, HD, generator=gen, device=dev).half()
        out, _ = _step(lay, q, k[step], v[step], True)
        L = ctx + step + 1
        s = torch.zeros(1, H, L, device=dev)
        getattr(ref, "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits)(
            q.float().view(1, H, HD).contiguous(), lay.k.kcache, s, lay.k.lookup_table, L, lay.k.outliers,
            lay.k.outlier_indices, 10000.0, sinks)
        w = s[0].half() / math.sqrt(HD)                                              # [H, L] fp16
        if sinks:
            ws = (torch.matmul(q.view(H, 1, HD), lay.k_sink) / math.sqrt(HD))[:, 0, :]   # fp16 [H, sinks]
            w = torch.cat((ws, w), dim=-1)
        p = torch.softmax(w, dim=-1, dtype=torch.float32).half()
        o = torch.zeros(1, H, HD, device=dev)
        getattr(ref, "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt2" % bits)(
            p[:, sinks:].float().reshape(1, H, L).contiguous(), lay.v.vcache, o, lay.v.lookup_table, L, lay.v.outliers,
            lay.v.outlier_indices)
        o = o[0].half()
        if sinks:
            o = o + torch.matmul(p[:, :sinks].view(H, 1, sinks), lay.v_sink)[:, 0, :]
        err = util.rel_err(out.float().reshape(1, -1), o.float().reshape(1, -1))
        worst = max(worst, err)
    print("fused decode, bits=%d ctx=%d sinks=%d: |out - reference pipeline| %.2e" % (bits, ctx, sinks, worst))
    assert worst < TOL

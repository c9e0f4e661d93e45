"""At-size parity (BASELINE.json configs) against the RUNNING reference: the reference's own extension built
for gfx950 (oracle/_ref, see tests/test_ref_gpu.py) is fast enough to produce whole 128K- and 1M-token results
on the MI355X, so the two decode matvecs are compared element for element over the full cache -- every tile
shape, the ragged last tile, the 8-wave tiles, the token-contiguous outlier mirror and the unaligned-row
fallback -- at 2, 3 and 4 bit, and the fused prefill pack at S = 8192 (config 4) is compared with the reference's
parallel pack kernels (codes, rescaled values) and with the reference-structured glue (outlier rows).
Tolerance for q.K^T / p.V: 1e-3 relative (north star); measured values are printed."""
import math

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C
TOL = 1e-3


@pytest.fixture(scope="module")
def ref():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import build_ref
    if build_ref.built() is None:
        pytest.skip("oracle/_ref not built")
    return build_ref.load()


def _inputs(bits, L, max_len, seed, dev):
    n = 2 ** bits
    g = torch.Generator(device=dev).manual_seed(seed)
    W = HD // 32 * bits
    d = {}
    for name in ("kmat", "vmat"):
        m = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
        m[:, :, L:] = 0
        d[name] = m
    d["klut"] = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    d["vrows"] = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    for name in ("k", "v"):
        d[name + "vals"] = torch.randn(max_len, 42, device=dev, generator=g) * (torch.rand(max_len, 42, device=dev, generator=g) > 0.2)
        d[name + "idx"] = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
    d["q"] = torch.randn(1, H, HD, device=dev, generator=g).half().float()
    d["p"] = torch.softmax(torch.randn(1, H, L, device=dev, generator=g) * 3, dim=-1).half().float().contiguous()
    return d


CASES = [
    # bits, L, max_len, pos_offset
    (4, 131072 + 77, None, 0),           # the headline configuration, ragged last tile
    (3, 131072 + 77, None, 5),           # config 3 (nuq3 + 5 sink tokens: pos_offset 5)
    (2, 131072 + 77, None, 0),
    (4, 4096 + 1, None, 0),              # config 2 sizes
    (4, 32768 + 19, None, 0),
    (4, 131072 + 77, 131072 + 79, 0),    # max_len % 4 != 0: unaligned rows (row-per-lane p.V fallback)
    (4, 1048576 + 33, None, 0),          # config 5: one layer of the 1M-token cache
]


@pytest.mark.parametrize("bits,L,max_len,pos_offset", CASES)
def test_matvecs_at_size_against_reference(ref, bits, L, max_len, pos_offset):
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    if max_len is None:
        max_len = (L + 64) // 64 * 64
    d = _inputs(bits, L, max_len, 1000 + bits + L % 1000, dev)
    kname = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits
    vname = "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt2" % bits
    # ---- q.K^T
    s_ref = torch.zeros(1, H, L, device=dev)
    getattr(ref, kname)(d["q"], d["kmat"], s_ref, d["klut"], L, d["kvals"], d["kidx"], 10000.0, pos_offset)
    s_rows = torch.zeros(1, H, L, device=dev)
    ops.score_k(bits, d["q"], d["kmat"], s_rows, d["klut"], L, 10000.0, pos_offset, d["kvals"], d["kidx"], accumulate=False)
    e_rows = util.rel_err(s_rows[0], s_ref[0])
    # the module swap's call: kvquant_amd.quant_cuda's shadow mirror + the decode kernel's mirror variant
    from kvquant_amd import quant_cuda as qc
    s_qc = torch.zeros(1, H, L, device=dev)
    getattr(qc, kname)(d["q"], d["kmat"], s_qc, d["klut"], L, d["kvals"], d["kidx"], 10000.0, pos_offset)
    e_qc = util.rel_err(s_qc[0], s_ref[0])
    qc.shadow_invalidate()
    assert e_qc < TOL, e_qc
    # the decode path's variant: tables from a prep call, token-contiguous outlier mirror, fused softmax partials
    ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    s_mir = torch.zeros(1, H, L, device=dev)
    kt, it = d["kvals"].t().contiguous(), d["kidx"].t().contiguous()
    parts = ops.score_k_prepared_softmax(bits, d["kmat"], s_mir, d["klut"], L, 10000.0, pos_offset, ws, d["kvals"],
                                         d["kidx"], 1.0 / math.sqrt(HD), n_parts, kt, it)
    e_mir = util.rel_err(s_mir[0], s_ref[0])
    probs, _ = ops.softmax_finish(s_mir[0], parts, n_parts, 1.0 / math.sqrt(HD))
    ref_p = torch.softmax((s_mir[0].half().float() * (1.0 / math.sqrt(HD))).half().float(), dim=-1).half().float()
    assert bool(((probs - ref_p).abs() <= ref_p.abs() * 2e-3 + 1e-7).all())
    # ---- p.V
    o_ref = torch.zeros(1, H, HD, device=dev)
    getattr(ref, vname)(d["p"], d["vmat"], o_ref, d["vrows"], L, d["vvals"], d["vidx"])
    o = torch.zeros(1, H, HD, device=dev)
    ops.mix_v(bits, d["p"], d["vmat"], o, d["vrows"], L, d["vvals"], d["vidx"], accumulate=False)
    e_v = util.rel_err(o.reshape(1, -1), o_ref.reshape(1, -1))
    print("bits=%d L=%d max_len=%d: |score - reference| rows %.2e mirror %.2e, |p.V - reference| %.2e"
          % (bits, L, max_len, e_rows, e_mir, e_v))
    assert e_rows < TOL and e_mir < TOL and e_v < TOL


@pytest.mark.parametrize("L,pos_offset", [(131072 + 77, 5), (4096 + 1, 0), (700, 1000000), (32768 + 19, 123456)])
def test_score_f16_pair_tables_at_size_against_reference(ref, L, pos_offset):
    """3 bit, decode path: q.K^T through the fp16 PAIR-SUM tables (include/kvq.h: KVQ_SCORE_F16_PAIR_TABLES -- one LDS
    look-up and one v_dot2_f32_f16 per rotation pair instead of two look-ups and two packed FMAs) against the reference's
    own kernel over the whole cache, incl. positions beyond 10^6.  The table entries and (cos, sin) are rounded to fp16
    (2^-11 relative), the sums accumulate in fp32: the bar is the north star's 1e-3; the measured error is printed and
    additionally held to 6e-4 (measured 3.5e-4), so that a regression of the rounding scheme (e.g. an fp16 accumulator) is caught."""
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    bits = 3
    max_len = (L + 64) // 64 * 64
    d = _inputs(bits, L, max_len, 2000 + L % 1000, dev)
    kname = "vecquant3matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2"
    s_ref = torch.zeros(1, H, L, device=dev)
    getattr(ref, kname)(d["q"], d["kmat"], s_ref, d["klut"], L, d["kvals"], d["kidx"], 10000.0, pos_offset)
    ws = ops.score_k_tables(bits, d["q"][0], d["klut"], H)
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    kt, it = d["kvals"].t().contiguous(), d["kidx"].t().contiguous()
    inv = 1.0 / math.sqrt(HD)
    s32 = torch.zeros(1, H, L, device=dev)
    ops.score_k_prepared_softmax(bits, d["kmat"], s32, d["klut"], L, 10000.0, pos_offset, ws, d["kvals"], d["kidx"], inv,
                                 n_parts, kt, it)
    s16 = torch.zeros(1, H, L, device=dev)
    parts = ops.score_k_prepared_softmax(bits, d["kmat"], s16, d["klut"], L, 10000.0, pos_offset, ws, d["kvals"], d["kidx"],
                                         inv, n_parts, kt, it, f16_pair=True)
    e32 = util.rel_err(s32[0], s_ref[0])
    e16 = util.rel_err(s16[0], s_ref[0])
    # the error in units of the fp16 rounding the reference applies to every score anyway (ML:873)
    ulp = (s16[0] - s_ref[0]).abs() / (s_ref[0].abs().clamp_min(1e-3) * 2.0 ** -11)
    print("L=%d pos_offset=%d: |score - reference| fp32 tables %.2e, fp16 pair tables %.2e (max %.2f fp16 ulps of the score, mean %.3f)"
          % (L, pos_offset, e32, e16, float(ulp.max()), float(ulp.mean())))
    assert e32 < 1e-5 and e16 < 6e-4 < TOL      # (measured: 3.5e-4 .. 3.7e-4)
    # the softmax partials of the same launch describe the scores it wrote
    probs, _ = ops.softmax_finish(s16[0], parts, n_parts, inv)
    ref_p = torch.softmax((s16[0].half().float() * inv).half().float(), dim=-1).half().float()
    assert bool(((probs - ref_p).abs() <= ref_p.abs() * 2e-3 + 1e-7).all())


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_prefill_pack_8192_against_reference(ref, bits):
    """BASELINE config 4: S = 8192 prompt tokens.  kvq_pack_{k,v}_fused (what QuantK/QuantV.parallel_pack run) vs
    (a) the reference's parallel pack kernels: packed codes and rescaled values, (b) the reference-structured glue
    (pack kernel + torch.topk / gather / sort): outlier rows, mirror, codebook rows -- all bit for bit."""
    from kvquant_amd.cache import QuantK, QuantV
    from tests import decode_check
    dev = torch.device("cuda:0")
    S, max_len = 8192, 8192 + 64
    quant, scale, shift = decode_check.quantizer(bits, seed=17 + bits)
    g = torch.Generator(device=dev).manual_seed(300 + bits)
    k = (torch.randn(C, S, device=dev, generator=g) * scale.to(dev)[:, None] * 1.2 + shift.to(dev)[:, None])
    k[torch.rand(C, S, device=dev, generator=g) < 0.01] *= 4
    k = k.half().float().reshape(H, HD, S).contiguous()
    v = torch.randn(C, S, device=dev, generator=g)
    v[torch.rand(C, S, device=dev, generator=g) < 0.01] *= 5
    # (distinct values per token: no ties at the 21/22 selection boundary, where torch.topk's choice is unspecified)
    v = (v + torch.arange(C, device=dev).unsqueeze(1) * 1e-6).reshape(H, HD, S).contiguous()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, device=dev)
    ka, kb = QuantK(rope_theta=10000.0, **kw), QuantK(rope_theta=10000.0, **kw)
    va, vb = QuantV(**kw), QuantV(**kw)
    for c in (ka, kb, va, vb):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    ka.parallel_pack(k)
    kb.parallel_pack(k, fused=False)
    assert ka.klen == kb.klen == S
    assert torch.equal(ka.kcache, kb.kcache)
    assert torch.equal(ka.outlier_indices, kb.outlier_indices)
    assert torch.equal(ka.outliers.view(torch.int32), kb.outliers.view(torch.int32))
    assert torch.equal(ka.outlier_indices_t, kb.outlier_indices_t)
    assert torch.equal(ka.outliers_t.view(torch.int32), kb.outliers_t.view(torch.int32))
    # the reference's own kernel: agreement wherever its result is defined (its race: tests/util.py)
    name = "vecquant%dappendvecKsparseParallel" % bits
    resc = torch.zeros(H, HD, S, device=dev)
    mine = torch.zeros_like(ka.kcache)
    from kvquant_amd import ops
    ops.pack_k_sparse_parallel(bits, mine, ka.lookup_table, k, resc, ka.outlier_threshold_lower,
                               ka.outlier_threshold_upper, 0)
    assert torch.equal(mine, ka.kcache)
    rw, rv = util.ref_pack_k_agrees_where_defined(ref, name, ka.lookup_table, k, ka.outlier_threshold_lower,
                                                  ka.outlier_threshold_upper, mine, resc)
    print("reference parallel K pack bits=%d S=%d: racy words per run %s of %d (outside the defined region only)"
          % (bits, S, rw, mine[:, :, :S].numel()))
    va.parallel_pack(v)
    vt = v.reshape(-1, S).t().contiguous()
    tk = vb.topk_inputs(vt)
    vb.parallel_pack(v, *tk)
    assert torch.equal(va.vcache, vb.vcache)
    assert torch.equal(va.lookup_table.view(torch.int32), vb.lookup_table.view(torch.int32))
    assert torch.equal(va.outlier_indices, vb.outlier_indices)
    assert torch.equal(va.outliers.view(torch.int32), vb.outliers.view(torch.int32))
    if bits != 3:    # (the reference's 3-bit parallel V kernel reads the wrong codebook column, KCU:2574-2579)
        mr = torch.zeros_like(va.vcache)
        getattr(ref, "vecquant%dappendvecVsparseParallel" % bits)(
            mr, va.lookup_table, v, tk[2][:, -1].contiguous(), tk[0][:, -1].contiguous())
        assert torch.equal(mr, va.vcache)


@pytest.mark.parametrize("bits,ctx,sinks,f16_pair", [(4, 131072, 0, False), (3, 131072, 0, False), (3, 131072, 5, False),
                                                     (4, 32768, 0, False), (3, 131072, 5, True), (3, 131072, 5, "f32"),
                                                     (3, 40000, 0, "f32")])
def test_decode_kv_end_to_end_against_reference_pipeline(ref, bits, ctx, sinks, f16_pair):
    """The one-call decode step at the BASELINE sizes against the pipeline assembled from the REFERENCE's own ops on the
    same cache: its q.K^T(+RoPE, +sparse) kernel -> half(score) / sqrt(d) in fp16 -> [fp16 sink scores in front,
    ML:1950-1962] -> fp32 softmax -> fp16 probabilities (ML:1972-1976) -> its p.V(+sparse) kernel -> half (ML:1290)
    [+ the sink tokens' fp16 matmul, ML:1987-1995].  (3, 131072, 5) is BASELINE config 3: nuq3 + 1 % + 5 fp16 sink
    tokens through decode_kv(k_sink=, v_sink=), i.e. the fused-sink launches at size.  ctx cached tokens filled by the
    fused prefill pack (the cache-fill of bench.py), then two tokens through decode_kv.  North-star tolerance 1e-3.
    f16_pair: the OPT-IN fp16 pair-sum score tables of 3-bit caches (QuantK.score_f16_pair): the scores themselves stay
    within 4e-4 of the reference's (test_score_f16_pair_tables_at_size_against_reference), i.e. within one ulp of the fp16
    value the reference rounds them to -- but where that rounding lands on the neighbouring fp16 value the probability
    moves by 2.8e-3, so the OUTPUT is held to 4e-3 here, not to 1e-3: which is why the mode is not the default."""
    import bench
    from kvquant_amd.cache import decode_kv
    dev = torch.device("cuda:0")
    util.sync_oracle_freqs(10000.0)
    gen = torch.Generator(device=dev).manual_seed(4321 + bits)
    max_len = (ctx + 8 + 63) // 64 * 64
    lay = bench.Layer(bits, max_len, gen, dev, sinks)
    # ("f32": the opt-in EXACT fp32 pair-sum tables -- 1024-lane score workgroups -- held to the default tolerance)
    lay.k.score_f32_pair = f16_pair == "f32"
    f16_pair = f16_pair is True
    lay.k.score_f16_pair = bool(f16_pair)
    lay.fill(ctx, gen, dev)
    k, v = bench.synth_tokens(2, lay.scale, lay.shift, gen, dev)
    worst = 0.0
    for step in range(2):
        q = torch.randn(H, HD, generator=gen, device=dev).half()
        if sinks:
            out, _ = decode_kv(lay.k, lay.v, q, k[step], v[step], k_sink=lay.k_sink, v_sink=lay.v_sink)
        else:
            out, _ = decode_kv(lay.k, lay.v, q, k[step], v[step])
        L = ctx + step + 1
        assert lay.k.klen == L + sinks and lay.v.vlen == L + sinks
        s = torch.zeros(1, H, L, device=dev)
        getattr(ref, "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits)(
            q.float().view(1, H, HD).contiguous(), lay.k.kcache, s, lay.k.lookup_table, L, lay.k.outliers,
            lay.k.outlier_indices, 10000.0, sinks)
        w = s[0].half() / math.sqrt(HD)
        if sinks:
            w = torch.cat(((torch.matmul(q.view(H, 1, HD), lay.k_sink) / math.sqrt(HD))[:, 0, :], w), dim=-1)
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
    print("decode_kv end to end, bits=%d ctx=%d sinks=%d%s: |out - reference pipeline| %.2e"
          % (bits, ctx, sinks, " fp16 pair tables" if f16_pair else "", worst))
    assert worst < (4e-3 if f16_pair else TOL)

import pytest
import torch

from tests import decode_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,prefill,steps,max_len", [(4, 40, 4, 64), (4, 0, 5, 16), (3, 30, 3, 64), (2, 20, 3, 64)])
def test_decode_kv_matches_reference_pipeline(bits, prefill, steps, max_len):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    err = decode_check.run(torch.device("cuda:0"), bits=bits, prefill=prefill, steps=steps, max_len=max_len)
    assert err < 1e-3


@pytest.mark.parametrize("bits,prefill", [(2, 20), (2, 0), (3, 24), (4, 24)])
def test_decode_kv_qnorm(bits, prefill):
    """K and V Q-Norm through the one-call decode step and the fused prefill packs (ML:486-497, 811-815,
    1116-1118, 1153-1156, 1237-1240, 1369-1375): at 2 bit both matvecs dequantise with the Q-Norm tables and the V
    residuals refer to the Q-Norm row; lookup_table2 rows are compared bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    err = decode_check.run(torch.device("cuda:0"), bits=bits, prefill=prefill, steps=4, max_len=64, norm=True)
    assert err < 1e-3


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("fused", [True, False])
def test_qnorm_prompt_outliers_reconstruct_exactly(bits, fused):
    """V Q-Norm prefill: the residual of a prompt token's outlier must refer to the table the p.V kernel dequantises
    with (QuantV.mix_table: lookup_table2 at 2 bit only), so that code + residual gives the value back.  Checked
    through the kernel itself: p.V on a one-hot probability row returns the token's dequantised vector."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    from kvquant_amd.cache import QuantV
    from oracle.glue import OracleQuantV
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    S, max_len = 24, 64
    quant, _, _ = decode_check.quantizer(bits, seed=bits)
    quant = tuple(quant) + (torch.tensor(1.07), torch.tensor(-0.02))
    vs = decode_check.util.v_tokens_no_ties(S, Cn=C, seed=70 + bits, k=21).half().float()
    gv = QuantV(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
                sparsity_threshold=0.99, device=dev)
    gv.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=True)
    vp = vs.t().reshape(H, HD, S).contiguous().to(dev)
    if fused:
        gv.parallel_pack(vp)
    else:
        uv, ui, lv, li = OracleQuantV.topk_inputs(vs, 0.99, C)
        gv.parallel_pack(vp, uv.to(dev), ui.to(dev), lv.to(dev), li.to(dev))
    for t in (0, 7, S - 1):
        onehot = torch.zeros(1, H, S, device=dev)
        onehot[:, :, t] = 1.0
        deq = torch.empty(1, H, HD, device=dev)
        ops.mix_v(bits, onehot, gv.vcache, deq, gv.mix_table(), S, gv.outliers, gv.outlier_indices, accumulate=False)
        deq = deq.reshape(-1).cpu()
        idx = gv.outlier_indices[t].long().cpu()
        x = vs[t]
        assert torch.allclose(deq[idx], x[idx], rtol=2e-6, atol=1e-6), (t, (deq[idx] - x[idx]).abs().max())
        # and the dense channels are codebook values of the same table
        rest = torch.ones(C, dtype=torch.bool)
        rest[idx] = False
        row = gv.mix_table()[t].cpu()
        assert bool(torch.isin(deq[rest], row).all())


@pytest.mark.parametrize("bits", [4, 3, 2])
def test_fused_prefill_pack_matches_reference_structure(bits):
    """kvq_pack_{k,v}_fused (one launch for the whole prompt) against the reference-structured prefill
    (pack kernel + torch.topk / gather / sort on the GPU): bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import QuantK, QuantV
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    S, max_len = 200, 256
    quant, scale, shift = decode_check.quantizer(bits, seed=7 + bits)
    g = torch.Generator().manual_seed(100 + bits)
    k = (torch.randn(C, S, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, S).to(dev)
    v = (torch.randn(C, S, generator=g) * 1.7).reshape(H, HD, S).to(dev)
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, device=dev)
    ka, kb = QuantK(rope_theta=10000.0, **kw), QuantK(rope_theta=10000.0, **kw)
    va, vb = QuantV(**kw), QuantV(**kw)
    for c in (ka, kb, va, vb):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    ka.parallel_pack(k)                      # fused
    kb.parallel_pack(k, fused=False)         # pack kernel + torch glue
    assert torch.equal(ka.kcache, kb.kcache)
    assert torch.equal(ka.outlier_indices, kb.outlier_indices)
    assert torch.equal(ka.outliers.view(torch.int32), kb.outliers.view(torch.int32))
    assert torch.equal(ka.outlier_indices_t, kb.outlier_indices_t)
    assert torch.equal(ka.outliers_t.view(torch.int32), kb.outliers_t.view(torch.int32))
    va.parallel_pack(v)                      # fused
    vt = v.reshape(-1, S).t().contiguous()
    vb.parallel_pack(v, *vb.topk_inputs(vt))
    assert torch.equal(va.vcache, vb.vcache)
    assert torch.equal(va.lookup_table.view(torch.int32), vb.lookup_table.view(torch.int32))
    assert torch.equal(va.outlier_indices, vb.outlier_indices)
    assert torch.equal(va.outliers.view(torch.int32), vb.outliers.view(torch.int32))


@pytest.mark.parametrize("bits", [2, 4])
def test_k_qnorm_paths_match_oracle(bits):
    """Q-Norm (load_lookup_table(norm=True), ML:486-497): the residuals refer to lookup_table2, and at 2 bit the
    score kernel dequantises with it (ML:811-815).  Decode (fused append + score) and prefill pack vs the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import QuantK
    from oracle.glue import OracleQuantK
    from tests import util
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    quant, scale, shift = decode_check.quantizer(bits, seed=11 + bits)
    quant = tuple(quant) + (1.07, -0.02)          # (upper, lower, [centroids], normscale, normoffset)
    prefill, steps, max_len = 24, 3, 64
    ks = util.k_tokens(prefill + steps, scale, shift, seed=60 + bits)
    g = torch.Generator().manual_seed(70 + bits)
    qs = torch.randn(steps, H, 1, HD, generator=g).half()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, rope_theta=10000.0)
    ok, gk = OracleQuantK(**kw), QuantK(device=dev, **kw)
    for c in (ok, gk):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=True)
    kp = ks[:prefill].half().float().t().reshape(H, HD, prefill).contiguous()
    ok.parallel_pack(kp)
    gk.parallel_pack(kp.to(dev))
    for i in range(steps):
        k16, q16 = ks[prefill + i].half(), qs[i]
        s_ref = ok.forward_fused_sparse(q16, k16)
        s_gpu = gk.forward_fused_sparse(q16.to(dev), k16.to(dev))
        assert util.rel_err(s_gpu.float().cpu(), s_ref.float()) < 2e-3
    L = prefill + steps
    assert torch.equal(ok.kcache[:, :, :L], gk.kcache[:, :, :L].cpu())
    assert torch.equal(ok.outlier_indices[:L], gk.outlier_indices[:L].cpu())
    assert torch.equal(ok.outliers[:L].view(torch.int32), gk.outliers[:L].cpu().view(torch.int32))


@pytest.mark.parametrize("bits", [4, 2])
def test_v_tie_at_clip_threshold(bits):
    """A token whose 21st and 22nd largest values are EQUAL (one fp16 token in ten).  The reference clips strictly
    outside the thresholds (KCU:2084) but stores the 21 largest as residuals (ML:1093-1096): the tied value is counted
    twice.  Default here: clipped iff stored (its reconstruction is exact, as in the reference's simulated path);
    reference_tie_quirk=True reproduces the reference-structured path bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import QuantV, ZERO_CODE
    from oracle import ckernels as ck
    from tests import util
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    quant, _, _ = decode_check.quantizer(bits, seed=3)
    x = util.v_tokens_no_ties(1, seed=8)[0].clone()
    top = torch.topk(x, 23)
    x[top.indices[20]] = x[top.indices[21]]          # 21st largest := 22nd largest
    pair = (int(top.indices[20]), int(top.indices[21]))     # two equal values: one is selected (lowest channel), one is the threshold
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=8, include_sparse=True,
              sparsity_threshold=0.99, device=dev)
    deq = {}
    for name, quirk in (("fixed", False), ("quirk", True)):
        vc = QuantV(**kw)
        vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        vc.reference_tie_quirk = quirk
        p = torch.ones(H, 1, 1, device=dev)
        vc.forward_fused_sparse(p, x.half().to(dev))                       # fused GPU-resident append
        codes = ck.unpack_codes(bits, vc.vcache.cpu(), C, 1)[0].long()
        d = vc.lookup_table[0].cpu()[codes]
        d.scatter_add_(0, vc.outlier_indices[0].cpu().long(), vc.outliers[0].cpu())
        sel = [c for c in pair if c in vc.outlier_indices[0].tolist()]
        assert len(sel) == 1
        deq[name] = (d, int(codes[sel[0]]), vc, sel[0])
    xf = x.half().float()
    tied = deq["fixed"][3]
    assert deq["fixed"][1] == ZERO_CODE[bits] and abs(float(deq["fixed"][0][tied] - xf[tied])) < 1e-5
    tied = deq["quirk"][3]
    assert deq["quirk"][1] != ZERO_CODE[bits] and abs(float(deq["quirk"][0][tied] - xf[tied])) > 0.5
    # the quirk variant equals the reference-structured path (legacy kernel + torch glue) bit for bit
    ref = QuantV(**kw)
    ref.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    uv, ui, lv, li = ref.topk_inputs(xf.unsqueeze(0).to(dev))
    ref.forward_fused_sparse(torch.ones(H, 1, 1, device=dev), x.half().to(dev), uv[0], ui[0], lv[0], li[0])
    q = deq["quirk"][2]
    # (which of the two equal values torch.topk calls "the 22nd" is unspecified; everything else must agree, and the
    #  dequantised token must be identical wherever the two paths stored the same channels)
    same = sorted(ref.outlier_indices[0].tolist()) == sorted(q.outlier_indices[0].tolist())
    if same:
        assert torch.equal(ref.vcache, q.vcache)
        assert torch.equal(ref.outliers[0].view(torch.int32), q.outliers[0].view(torch.int32))
    assert torch.equal(ref.lookup_table[0], q.lookup_table[0])


@pytest.mark.parametrize("bits,heads,prefill,max_len", [(4, 40, 150, 256), (4, 8, 300, 512), (3, 40, 0, 64), (2, 16, 70, 128)])
def test_decode_kv_other_model_widths(bits, heads, prefill, max_len):
    """the one-call decode step at head counts other than 32 (LLaMA-13B: 40 heads, hidden 5120, 52 outlier slots;
    narrow models): partial unit groups of the p.V kernel, head groups of the score kernel, other selection sizes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    err = decode_check.run(torch.device("cuda:0"), bits=bits, prefill=prefill, steps=3, max_len=max_len, heads=heads)
    assert err < 1e-3


@pytest.mark.parametrize("bits,prefill,max_len", [(4, 200, 256), (3, 1000, 1024), (4, 40000, 40064)])
def test_fused_sinks_match_torch_glue(bits, prefill, max_len):
    """decode_kv with the fp16 sink caches handed in (scores in the prologue launch, output share in the softmax launch)
    against the reference's glue around it (torch.matmul / division before, torch.matmul + add after, ML:1950-1962,
    1987-1995) on identical caches: sink probabilities to an fp16 ulp, outputs to the decode tolerance.  The short
    caches take the p.V kernel that normalises the scores itself, the long one the softmax-finish launch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import math
    from kvquant_amd.cache import QuantK, QuantV, decode_kv
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    n_sink = 5
    quant, scale, shift = decode_check.quantizer(bits, seed=9 + bits)
    g = torch.Generator().manual_seed(prefill + 17 * bits)
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=n_sink, device=dev)
    pairs = []
    k = (torch.randn(C, prefill, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, prefill).to(dev)
    v = (torch.randn(C, prefill, generator=g) * 1.7).reshape(H, HD, prefill).to(dev)
    for _ in range(2):
        kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
        kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        kc.klen = vc.vlen = n_sink
        kc.parallel_pack(k)
        vc.parallel_pack(v)
        pairs.append((kc, vc))
    k_sink = (torch.randn(H, HD, n_sink, generator=g) * 0.6).half().to(dev)
    v_sink = torch.randn(H, n_sink, HD, generator=g).half().to(dev)
    for step in range(2):
        q = torch.randn(H, HD, generator=g).half().to(dev)
        kn = (torch.randn(C, generator=g) * scale * 1.3 + shift).half().to(dev)
        vn = (torch.randn(C, generator=g) * 1.7).half().to(dev)
        sink_scores = (torch.bmm(q.unsqueeze(1), k_sink) / math.sqrt(HD)).squeeze(1)
        out_ref, sp_ref = decode_kv(pairs[0][0], pairs[0][1], q, kn, vn, sink_scores)
        out_ref = out_ref + torch.bmm(sp_ref.unsqueeze(1), v_sink).transpose(0, 1).float()
        out, sp = decode_kv(pairs[1][0], pairs[1][1], q, kn, vn, k_sink=k_sink, v_sink=v_sink)
        torch.cuda.synchronize()
        assert (sp.float() - sp_ref.float()).abs().max().item() <= 2e-3 * sp_ref.float().abs().max().item() + 1e-7
        scale_o = out_ref.abs().max().item() + 1e-6
        assert (out - out_ref).abs().max().item() <= 2e-3 * scale_o, step


@pytest.mark.parametrize("bits,S,split", [(4, 700, 300), (4, 9000, 4096), (3, 1200, 640), (2, 40, 10)])
def test_token_sharded_attention_matches_unsharded(bits, S, split):
    """cache.shard_attention over the shards of a context split along the token axis (positions of the second shard
    start at `split`; the new token is appended to it; a third shard is EMPTY) merged by kvq_combine_shards on the
    concatenated records -- what one all-gather per layer delivers -- against decode_kv on the unsharded cache: the
    same attention output to the decode tolerance (probabilities are rounded to fp16 per shard).  The torch formula
    (cache.combine_shards, used by the CPU tests) must agree with the kernel."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    from kvquant_amd.cache import QuantK, QuantV, decode_kv, shard_attention, combine_shards, shard_record_floats
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    quant, scale, shift = decode_check.quantizer(bits, seed=21 + bits)
    g = torch.Generator().manual_seed(S + bits)
    k = (torch.randn(C, S, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, S).to(dev)
    v = (torch.randn(C, S, generator=g) * 1.7).reshape(H, HD, S).to(dev)

    def make(ks, vs, max_len):
        kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
                  sparsity_threshold=0.99, first_few_fp16=0, device=dev)
        kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
        kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        if ks.shape[-1]:
            kc.parallel_pack(ks.contiguous())
            vc.parallel_pack(vs.contiguous())
        return kc, vc

    full = make(k, v, (S + 64 + 63) // 64 * 64)
    s0 = make(k[:, :, :split], v[:, :, :split], (split + 63) // 64 * 64)
    s1 = make(k[:, :, split:], v[:, :, split:], (S - split + 64 + 63) // 64 * 64)
    s2 = make(k[:, :, :0], v[:, :, :0], 64)
    nrec = shard_record_floats(H, HD)
    for step in range(2):
        q = torch.randn(H, HD, generator=g).half().to(dev)
        kn = (torch.randn(C, generator=g) * scale * 1.3 + shift).half().to(dev)
        vn = (torch.randn(C, generator=g) * 1.7).half().to(dev)
        ref, _ = decode_kv(full[0], full[1], q, kn, vn)
        gathered = torch.empty(3 * nrec, device=dev)
        o0, m0, z0 = shard_attention(s0[0], s0[1], q, pos_base=0, record=gathered[:nrec])
        o1, m1, z1 = shard_attention(s1[0], s1[1], q, kn, vn, pos_base=split, record=gathered[nrec:2 * nrec])
        o2, m2, z2 = shard_attention(s2[0], s2[1], q, pos_base=S + 100, record=gathered[2 * nrec:])
        assert bool(torch.isinf(m2).all()) and float(z2.abs().max()) == 0.0
        out = torch.empty(1, H, HD, device=dev)
        ops.combine_shards(gathered, 3, H, HD, out)
        out_t = combine_shards(torch.stack((o0, o1, o2)), torch.stack((m0, m1, m2)), torch.stack((z0, z1, z2)))
        torch.cuda.synchronize()
        scale_o = ref.abs().max().item() + 1e-6
        assert (out - ref).abs().max().item() <= 2e-3 * scale_o, step
        assert (out - out_t).abs().max().item() <= 1e-5 * scale_o, step
    with pytest.raises(ValueError):          # a cache that counts sink tokens it was never given (klen < first_few_fp16)
        kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=64, include_sparse=True,
                  sparsity_threshold=0.99, first_few_fp16=5, device=dev)
        shard_attention(QuantK(rope_theta=10000.0, **kw), QuantV(**kw), q)


@pytest.mark.parametrize("bits,S,split,sinks", [(3, 900, 400, 5), (4, 300, 0, 3)])
def test_token_sharded_attention_with_sink_tokens(bits, S, split, sinks):
    """BASELINE config 3 on a token-sharded context (round 5): the shard that holds the START of the context also gets the
    fp16 attention-sink tokens (first_few_fp16, ML:1464-1466) -- their scores enter its softmax and its (max, normaliser),
    their values its output -- and the merged result equals decode_kv(k_sink=, v_sink=) on the unsharded cache.
    split = 0: the first shard holds ONLY the sink tokens."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    from kvquant_amd.cache import QuantK, QuantV, decode_kv, shard_attention, shard_record_floats
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    quant, scale, shift = decode_check.quantizer(bits, seed=5 + bits)
    g = torch.Generator().manual_seed(S + bits)
    k = (torch.randn(C, S, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, S).to(dev)
    v = (torch.randn(C, S, generator=g) * 1.7).reshape(H, HD, S).to(dev)
    k_sink = (torch.randn(H, HD, sinks, generator=g) * 0.5).half().to(dev)
    v_sink = torch.randn(H, sinks, HD, generator=g).half().to(dev)

    def make(ks, vs, max_len, nf):
        kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
                  sparsity_threshold=0.99, first_few_fp16=nf, device=dev)
        kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
        kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        kc.klen += nf
        vc.vlen += nf
        if ks.shape[-1]:
            kc.parallel_pack(ks.contiguous())
            vc.parallel_pack(vs.contiguous())
        return kc, vc

    full = make(k, v, (S + 64 + 63) // 64 * 64, sinks)
    s0 = make(k[:, :, :split], v[:, :, :split], (split + 64 + 63) // 64 * 64, 0)          # plain shard caches; the sinks come per call
    s1 = make(k[:, :, split:], v[:, :, split:], (S - split + 64 + 63) // 64 * 64, 0)
    nrec = shard_record_floats(H, HD)
    for step in range(2):
        q = torch.randn(H, HD, generator=g).half().to(dev)
        kn = (torch.randn(C, generator=g) * scale * 1.3 + shift).half().to(dev)
        vn = (torch.randn(C, generator=g) * 1.7).half().to(dev)
        ref, _ = decode_kv(full[0], full[1], q, kn, vn, k_sink=k_sink, v_sink=v_sink)
        gathered = torch.empty(2 * nrec, device=dev)
        shard_attention(s0[0], s0[1], q, pos_base=sinks, record=gathered[:nrec], k_sink=k_sink, v_sink=v_sink)
        shard_attention(s1[0], s1[1], q, kn, vn, pos_base=sinks + split, record=gathered[nrec:])
        out = torch.empty(1, H, HD, device=dev)
        ops.combine_shards(gathered, 2, H, HD, out)
        torch.cuda.synchronize()
        scale_o = ref.abs().max().item() + 1e-6
        assert (out - ref).abs().max().item() <= 2e-3 * scale_o, (step, (out - ref).abs().max().item() / scale_o)
    # a cache that COUNTS sink tokens must be handed them: no silent drop from the softmax
    q = torch.randn(H, HD, generator=g).half().to(dev)
    with pytest.raises(ValueError, match="sink"):
        shard_attention(full[0], full[1], q, pos_base=sinks)
    with pytest.raises(ValueError, match="sink"):
        shard_attention(full[0], full[1], q, pos_base=sinks, k_sink=k_sink[:, :, :sinks - 1].contiguous(),
                        v_sink=v_sink[:, :sinks - 1].contiguous())


@pytest.mark.parametrize("bits,sinks,L0", [(4, 0, 40), (3, 5, 300), (2, 0, 40)])
def test_one_call_decode_step_equals_the_five_call_path(bits, sinks, L0):
    """kvq_decode_step (prologue, q.K^T, softmax, p.V, reduce from ONE library call) against the same launches issued
    one by one from Python: identical kernels on identical inputs -> bit-identical outputs and cache state"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import cache as kcache
    from kvquant_amd.cache import QuantK, QuantV, decode_kv
    dev = torch.device("cuda:0")
    H, HD, C = decode_check.H, decode_check.HD, decode_check.C
    quant, scale, shift = decode_check.quantizer(bits, seed=bits)
    steps = 3
    ks = decode_check.util.k_tokens(L0 + steps, scale, shift, seed=11).half()
    vs = decode_check.util.v_tokens(L0 + steps, seed=12).half()
    g = torch.Generator().manual_seed(13)
    qs = torch.randn(steps, H, HD, generator=g).half()
    k_sink = (torch.randn(H, HD, sinks, generator=g) * 0.5).half().to(dev) if sinks else None
    v_sink = torch.randn(H, sinks, HD, generator=g).half().to(dev) if sinks else None
    outs = {}
    for one_call in (False, True):
        kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=L0 + 64, include_sparse=True,
                  sparsity_threshold=0.99, first_few_fp16=sinks)
        kc, vc = QuantK(rope_theta=10000.0, device=dev, **kw), QuantV(device=dev, **kw)
        kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        kc.klen += sinks
        vc.vlen += sinks
        kc.parallel_pack(ks[:L0].float().t().reshape(H, HD, L0).contiguous().to(dev))
        vc.parallel_pack(vs[:L0].float().t().reshape(H, HD, L0).contiguous().to(dev))
        kcache.ONE_CALL_PER_LAYER = one_call
        try:
            res = []
            for i in range(steps):
                o, sp = decode_kv(kc, vc, qs[i].to(dev), ks[L0 + i].to(dev), vs[L0 + i].to(dev), k_sink=k_sink, v_sink=v_sink)
                res.append((o.clone(), None if sp is None else sp.clone()))
        finally:
            kcache.ONE_CALL_PER_LAYER = True
        outs[one_call] = (res, kc.kcache.clone(), vc.vcache.clone(), kc.outliers_t.clone(), vc.lookup_table.clone(), kc.klen)
    a, b = outs[False], outs[True]
    for (o1, s1), (o2, s2) in zip(a[0], b[0]):
        assert torch.equal(o1, o2)
        assert (s1 is None and s2 is None) or torch.equal(s1, s2)
    for x, y in zip(a[1:5], b[1:5]):
        assert torch.equal(x, y)
    assert a[5] == b[5] == L0 + steps + sinks

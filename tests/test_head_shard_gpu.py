"""Head-sharded cache (SURVEY 8e "by head"): kvquant_amd.cache.HeadShard + kvq_extract_heads + sharding.head_sharded_step.
A rank's shard must hold EXACTLY what the unsharded cache holds for its heads -- packed words, V codebook rows, and its
share of every token's 42 outlier entries, selected over the whole token -- and its attention output must be the
unsharded output's rows for its heads.  One process builds all shards next to the unsharded layer (the reference's
placement is by layer; there is no reference form of this cut to compare with, the unsharded path is the oracle-pinned
one); a world-2 gloo run on the one visible GPU covers the collective."""
import os

import pytest
import torch

from tests.test_sharding_gpu import C, H, HD, _free_port, _make_layer, _prompt, _tokens

pytestmark = pytest.mark.gpu


def _shards(bits, seed, max_len, dev, prompt, splits, stage_len=64):
    from kvquant_amd.cache import HeadShard
    from tests import decode_check
    quant, _, _ = decode_check.quantizer(bits, seed=seed)
    out = []
    for h0, n in splits:
        hs = HeadShard(bits, C, H, (h0, n), max_len, device=dev, stage_len=stage_len)
        hs.load_lookup_table(quant, quant)
        if prompt is not None:
            hs.pack(prompt[0].to(dev), prompt[1].to(dev))
        out.append(hs)
    return out


def _check_state(kc, vc, shards, L, sinks=0):
    """the shards' caches against the head slices of the unsharded ones as the ORACLE cuts them (oracle.glue.slice_heads: the
    checker's own statement of the extract, evaluated on CPU copies), columns [0, L)"""
    from oracle import glue
    full = [t[..., :L].cpu() if t.dim() == 3 else t[:L].cpu() for t in
            (kc.kcache, vc.vcache, vc.lookup_table, kc.outliers, kc.outlier_indices, vc.outliers, vc.outlier_indices)]
    for hs in shards:
        h0, n = hs.h0, hs.n_heads
        assert hs.k.klen == L + sinks and hs.v.vlen == L + sinks
        kw, vw, rows, (kv, ki), (vv, vi) = glue.slice_heads(*full, h0, n, HD)
        assert torch.equal(hs.k.kcache[:, :, :L].cpu(), kw)
        assert torch.equal(hs.v.vcache[:, :, :L].cpu(), vw)
        assert torch.equal(hs.v.lookup_table[:L].cpu(), rows)
        if vc.lookup_table2 is not None:
            assert torch.equal(hs.v.lookup_table2[:L], vc.lookup_table2[:L])
        for want_v, want_i, sh_val, sh_idx in ((kv, ki, hs.k.outliers, hs.k.outlier_indices), (vv, vi, hs.v.outliers, hs.v.outlier_indices)):
            assert torch.equal(sh_val[:L].cpu().view(torch.int32), want_v.view(torch.int32))
            assert torch.equal(sh_idx[:L].cpu(), want_i)
            assert bool((sh_idx[:L, 1:] >= sh_idx[:L, :-1]).all())          # rows stay sorted by channel
        assert torch.equal(hs.k.outliers_t[:, :L], hs.k.outliers[:L].t())
        assert torch.equal(hs.k.outlier_indices_t[:, :L], hs.k.outlier_indices[:L].t())
    # every outlier of every token lives in exactly one shard
    total = sum((hs.k.outliers[:L] != 0).sum().item() for hs in shards)
    assert total == (kc.outliers[:L] != 0).sum().item()


@pytest.mark.parametrize("bits,splits,S", [(4, [(0, 16), (16, 16)], 300), (3, [(0, 8), (8, 8), (16, 8), (24, 8)], 130),
                                           (4, [(0, 11), (11, 11), (22, 10)], 70), (2, [(0, 4), (4, 28)], 65)])
def test_head_shards_hold_the_unsharded_cache_and_attend_like_it(bits, splits, S):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import decode_kv
    from tests import decode_check, util
    util.sync_oracle_freqs(10000.0)
    dev = torch.device("cuda:0")
    steps, max_len = 3, 512
    quant, scale, shift = decode_check.quantizer(bits, seed=7)
    prompt = _prompt(21, S, scale, shift)
    kc, vc, _, _ = _make_layer(bits, 7, max_len, dev, prompt)
    shards = _shards(bits, 7, max_len, dev, prompt, splits)          # (S > stage_len: several staging pieces)
    _check_state(kc, vc, shards, S)
    q, k, v = _tokens(99, steps, scale, shift)
    for st in range(steps):
        ref, _ = decode_kv(kc, vc, q[st].to(dev), k[st].to(dev), v[st].to(dev))
        scale_ref = ref.abs().max().item() + 1e-6
        for hs in shards:
            out = hs.attend(q[st].to(dev), k[st].to(dev), v[st].to(dev))
            want = ref[:, hs.h0:hs.h0 + hs.n_heads]
            assert out.shape == want.shape
            assert (out - want).abs().max().item() <= 1e-3 * scale_ref, (bits, st, hs.h0)
        _check_state(kc, vc, shards, S + st + 1)


def test_head_shard_rejects_what_it_does_not_carry():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import HeadShard
    dev = torch.device("cuda:0")
    with pytest.raises(ValueError):
        HeadShard(4, C, H, (30, 4), 128, device=dev)
    from tests import decode_check
    quant, _, _ = decode_check.quantizer(4, seed=1)
    hs = HeadShard(4, C, H, (0, 8), 128, device=dev)
    with pytest.raises(ValueError):
        hs.load_lookup_table(quant, quant, include_sparse=False)


def _head_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kvquant_amd import sharding
        from tests import decode_check
        dev = torch.device("cuda:0")
        bits, S, steps = 4, 200, 2
        quant, scale, shift = decode_check.quantizer(bits, seed=3)
        split = sharding.head_assignment(H, world)
        hs = _shards(bits, 3, 512, dev, _prompt(77, S, scale, shift), [split[rank]])[0]
        q, k, v = _tokens(555, steps, scale, shift)
        outs = []
        for st in range(steps):
            # (the one visible GPU is shared by both ranks: the outputs cross the process boundary as CPU tensors; with
            #  RCCL on a multi-GPU node the same call gathers device tensors)
            outs.append(sharding.head_sharded_step(lambda: hs.attend(q[st].to(dev), k[st].to(dev), v[st].to(dev)).cpu(),
                                                   H, HD).clone())
        ret[rank] = outs
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_head_sharded_step_over_real_shards():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from kvquant_amd.cache import decode_kv
    from tests import decode_check
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_head_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    bits, S, steps = 4, 200, 2
    quant, scale, shift = decode_check.quantizer(bits, seed=3)
    kc, vc, _, _ = _make_layer(bits, 3, 512, dev, _prompt(77, S, scale, shift))
    q, k, v = _tokens(555, steps, scale, shift)
    for st in range(steps):
        ref, _ = decode_kv(kc, vc, q[st].to(dev), k[st].to(dev), v[st].to(dev))
        ref = ref.cpu()
        for r in range(world):
            assert (ret[r][st] - ref).abs().max().item() <= 1e-3 * (ref.abs().max().item() + 1e-6), (st, r)
        assert torch.equal(ret[0][st], ret[1][st])


def test_extract_heads_argument_checks():
    """violated preconditions come back as KVQ_EINVAL before any launch; n = 0 is a no-op"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import _lib, ops
    from kvquant_amd.cache import HeadShard
    from tests import decode_check
    dev = torch.device("cuda:0")
    quant, _, _ = decode_check.quantizer(4, seed=1)
    hs = HeadShard(4, C, H, (8, 8), 128, device=dev).load_lookup_table(quant, quant)
    ops.extract_heads(4, 8, 8, hs.full_k, hs.full_v, hs.k, hs.v, 0, 0, 0)                    # nothing to move
    with pytest.raises(_lib.KvqError):
        ops.extract_heads(4, 28, 8, hs.full_k, hs.full_v, hs.k, hs.v, 0, 0, 1)               # heads 28..35 of 32
    with pytest.raises(_lib.KvqError):
        ops.extract_heads(4, 8, 8, hs.full_k, hs.full_v, hs.k, hs.v, 60, 0, 8)               # source columns past the end
    with pytest.raises(_lib.KvqError):
        ops.extract_heads(4, 8, 8, hs.full_k, hs.full_v, hs.k, hs.v, 0, 125, 8)              # destination columns past the end
    with pytest.raises(ValueError):
        ops.extract_heads(4, 8, 4, hs.full_k, hs.full_v, hs.k, hs.v, 0, 0, 1)                # the shard holds 8 heads, not 4
    assert hs.k.klen == 0 and not bool(hs.k.kcache.any())


@pytest.mark.parametrize("bits,norm,sinks,shared_stage", [(3, False, 5, False), (3, False, 5, True), (2, True, 0, False), (4, True, 3, True)])
def test_head_shards_carry_sinks_and_qnorm(bits, norm, sinks, shared_stage):
    """BASELINE config 3 on a head-sharded layer (round 5): nuq3 + 5 fp16 attention-sink tokens -- every shard gets its heads'
    slices of the fp16 sink caches and its output must be the unsharded decode_kv rows of its heads -- and Q-Norm
    quantizers (the normalised V codebook rows travel with the extract); optionally all shards stage through ONE shared
    set of staging buffers.  One library call per shard and token (kvq_head_shard_step)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import HeadShard, QuantK, QuantV, decode_kv
    from tests import decode_check, util
    util.sync_oracle_freqs(10000.0)
    dev = torch.device("cuda:0")
    S, steps, max_len = 150, 3, 512
    quant, scale, shift = decode_check.quantizer(bits, seed=11)
    if norm:      # (upper, lower, [centroids], normscale, normoffset), SQ:550-555
        quant = tuple(quant) + (torch.tensor(1.07), torch.tensor(-0.02))
    prompt = _prompt(31, S, scale, shift)
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=sinks, device=dev)
    kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
    kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=norm)
    vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=norm)
    kc.klen += sinks
    vc.vlen += sinks
    kc.parallel_pack(prompt[0].to(dev).contiguous())
    vc.parallel_pack(prompt[1].to(dev).contiguous())
    g = torch.Generator().manual_seed(5)
    k_sink = (torch.randn(H, HD, sinks, generator=g) * 0.5).half().to(dev) if sinks else None
    v_sink = torch.randn(H, sinks, HD, generator=g).half().to(dev) if sinks else None
    splits = [(0, 8), (8, 16), (24, 8)]
    shards, stage = [], None
    for h0, n in splits:
        hs = HeadShard(bits, C, H, (h0, n), max_len, device=dev, stage_len=64, first_few_fp16=sinks,
                       staging=stage if shared_stage else None)
        if shared_stage and stage is None:
            stage = hs
        hs.load_lookup_table(quant, quant, norm=norm)
        hs.pack(prompt[0].to(dev), prompt[1].to(dev))
        shards.append(hs)
    _check_state(kc, vc, shards, S, sinks)
    q, k, v = _tokens(77, steps, scale, shift)
    for st in range(steps):
        ref, _ = decode_kv(kc, vc, q[st].to(dev), k[st].to(dev), v[st].to(dev), k_sink=k_sink, v_sink=v_sink)
        scale_ref = ref.abs().max().item() + 1e-6
        for hs in shards:
            sl = slice(hs.h0, hs.h0 + hs.n_heads)
            out = hs.attend(q[st].to(dev), k[st].to(dev), v[st].to(dev),
                            k_sink=None if not sinks else k_sink[sl].contiguous(),
                            v_sink=None if not sinks else v_sink[sl].contiguous())
            assert (out - ref[:, sl]).abs().max().item() <= 1e-3 * scale_ref, (bits, st, hs.h0)
        _check_state(kc, vc, shards, S + st + 1, sinks)

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


def _check_state(kc, vc, shards, L):
    """the shards' caches against the unsharded ones, columns [0, L)"""
    W = kc.kcache.shape[1]
    for hs in shards:
        h0, n = hs.h0, hs.n_heads
        assert hs.k.klen == L and hs.v.vlen == L
        assert torch.equal(hs.k.kcache[:, :, :L], kc.kcache[h0:h0 + n, :, :L])
        assert torch.equal(hs.v.vcache[:, :, :L], vc.vcache[h0:h0 + n, :, :L])
        assert torch.equal(hs.v.lookup_table[:L], vc.lookup_table[:L])
        c0, c1 = h0 * HD, (h0 + n) * HD
        for full_val, full_idx, sh_val, sh_idx in ((kc.outliers, kc.outlier_indices, hs.k.outliers, hs.k.outlier_indices),
                                                   (vc.outliers, vc.outlier_indices, hs.v.outliers, hs.v.outlier_indices)):
            fv, fi = full_val[:L], full_idx[:L]
            own = (fi >= c0) & (fi < c1)
            assert torch.equal(sh_val[:L], torch.where(own, fv, torch.zeros_like(fv)))
            exp_idx = torch.where(own, fi - c0, torch.where(fi < c0, torch.zeros_like(fi), torch.full_like(fi, c1 - c0 - 1)))
            assert torch.equal(sh_idx[:L], exp_idx)
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

"""The two multi-GPU cuts over REAL compressed-cache layers, world_size 2 over gloo, both ranks computing on the one
visible GPU (activations / shard records cross the process boundary as CPU tensors; with RCCL on a multi-GPU node the
same code keeps them on the devices):
  * sharding.StreamPipeline (layer placement of modeling_llama.py:2428-2453): rank r owns layer r, two decode streams in
    flight, every stage is kvquant_amd.cache.decode_kv on that rank's QuantK / QuantV;
  * sharding.token_sharded_step (context split along the token axis): rank r holds its slice of the cached tokens of the
    one layer, cache.shard_attention per shard, one all-gather, exact merge.
Both are compared with the same computation done in one process."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

H, HD, C = 32, 128, 4096


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_layer(bits, seed, max_len, dev, prompt):
    from kvquant_amd.cache import QuantK, QuantV
    from tests import decode_check
    quant, scale, shift = decode_check.quantizer(bits, seed=seed)
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, device=dev)
    kc, vc = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
    kc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    if prompt is not None and prompt[0].shape[-1]:
        kc.parallel_pack(prompt[0].to(dev).contiguous())
        vc.parallel_pack(prompt[1].to(dev).contiguous())
    return kc, vc, scale, shift


def _tokens(seed, n, scale, shift):
    g = torch.Generator().manual_seed(seed)
    k = (torch.randn(n, C, generator=g) * scale * 1.3 + shift).half()
    v = (torch.randn(n, C, generator=g) * 1.7).half()
    q = torch.randn(n, H, HD, generator=g).half()
    return q, k, v


def _prompt(seed, S, scale, shift):
    g = torch.Generator().manual_seed(seed)
    k = (torch.randn(C, S, generator=g) * scale[:, None] * 1.3 + shift[:, None]).reshape(H, HD, S)
    v = (torch.randn(C, S, generator=g) * 1.7).reshape(H, HD, S)
    return k, v


def _pipeline_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kvquant_amd import sharding
        from kvquant_amd.cache import decode_kv
        from tests import decode_check
        dev = torch.device("cuda:0")
        bits, S, steps, streams = 4, 48, 3, 2
        # one layer per rank and stream, every rank builds only its own
        layers = {}
        for s in range(streams):
            quant, scale, shift = decode_check.quantizer(bits, seed=10 + rank)
            layers[s] = _make_layer(bits, 10 + rank, 128, dev, _prompt(100 * s + rank, S, scale, shift))
        toks = {s: _tokens(1000 + 10 * s + rank, steps, layers[s][2], layers[s][3]) for s in range(streams)}

        def stage(s, st, x):
            kc, vc = layers[s][0], layers[s][1]
            q, k, v = toks[s]
            qq = (q[st].float() + x.view(H, HD)[:, :HD].float() * 1e-3).half().to(dev)      # true dependency on the hand-over
            out, _ = decode_kv(kc, vc, qq, k[st].to(dev), v[st].to(dev))
            return out.reshape(1, 1, C).half().cpu()
        template = torch.zeros(1, 1, C, dtype=torch.float16)
        pipe = sharding.StreamPipeline(stage, streams, rank=rank, world=world)
        finals = pipe.run(steps, lambda s, st: template, template)
        ret[rank] = [f.clone() for f in finals]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_stream_pipeline_over_real_layers():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from kvquant_amd.cache import decode_kv
    from tests import decode_check
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipeline_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    finals = ret[0]
    assert len(finals) == 3 * 2 and ret[1] == []
    # the same two layers per stream, one process, in sequence
    dev = torch.device("cuda:0")
    bits, S, steps, streams = 4, 48, 3, 2
    ref = []
    layers = {}
    for s in range(streams):
        for r in range(world):
            quant, scale, shift = decode_check.quantizer(bits, seed=10 + r)
            layers[(s, r)] = _make_layer(bits, 10 + r, 128, dev, _prompt(100 * s + r, S, scale, shift))
    for st in range(steps):
        for s in range(streams):
            x = torch.zeros(1, 1, C, dtype=torch.float16)
            for r in range(world):
                kc, vc, scale, shift = layers[(s, r)]
                q, k, v = _tokens(1000 + 10 * s + r, steps, scale, shift)
                qq = (q[st].float() + x.view(H, HD).float() * 1e-3).half().to(dev)
                out, _ = decode_kv(kc, vc, qq, k[st].to(dev), v[st].to(dev))
                x = out.reshape(1, 1, C).half().cpu()
            ref.append(x)
    for a, b in zip(finals, ref):
        assert torch.equal(a, b)


def _token_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kvquant_amd import sharding
        from kvquant_amd.cache import shard_attention
        from tests import decode_check
        dev = torch.device("cuda:0")
        bits, S, steps = 4, 600, 2
        quant, scale, shift = decode_check.quantizer(bits, seed=3)
        kp, vp = _prompt(77, S, scale, shift)
        per = S // world
        lo, hi = rank * per, (S if rank == world - 1 else (rank + 1) * per)
        kc, vc, _, _ = _make_layer(bits, 3, 1024, dev, (kp[:, :, lo:hi], vp[:, :, lo:hi]))
        q, k, v = _tokens(555, steps, scale, shift)
        last = rank == world - 1
        outs = []
        for st in range(steps):
            def shard_fn(record):
                if last:
                    o, M, Z = shard_attention(kc, vc, q[st].to(dev), k[st].to(dev), v[st].to(dev), pos_base=lo)
                else:
                    o, M, Z = shard_attention(kc, vc, q[st].to(dev), pos_base=lo)
                return o.cpu(), M.cpu(), Z.cpu()
            outs.append(sharding.token_sharded_step(shard_fn).clone())
        ret[rank] = outs
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_token_sharded_step_over_real_shards():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from kvquant_amd.cache import decode_kv
    from tests import decode_check
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_token_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    bits, S, steps = 4, 600, 2
    quant, scale, shift = decode_check.quantizer(bits, seed=3)
    kc, vc, _, _ = _make_layer(bits, 3, 1024, dev, _prompt(77, S, scale, shift))
    q, k, v = _tokens(555, steps, scale, shift)
    for st in range(steps):
        ref, _ = decode_kv(kc, vc, q[st].to(dev), k[st].to(dev), v[st].to(dev))
        ref = ref.cpu()
        for r in range(world):
            got = ret[r][st]
            assert (got - ref).abs().max().item() <= 2e-3 * (ref.abs().max().item() + 1e-6), (st, r)
        assert torch.equal(ret[0][st], ret[1][st])          # every rank forms the same merged output

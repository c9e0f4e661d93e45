"""CPU (gloo, world_size 2 and 3): the layer-sharded pipeline routes activations exactly like
the reference's set_devices placement (modeling_llama.py:2428-2453, 2552-2556, 2583-2585)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kvquant_amd import sharding


def test_assignment_matches_reference_rule():
    # 32 layers on 8 GPUs: 4 contiguous layers each (DL:171 -> ML:2428)
    a = sharding.layer_assignment(32, 8)
    assert a == [list(range(4 * r, 4 * r + 4)) for r in range(8)]
    assert sharding.split_indices(32, 8) == [4, 8, 12, 16, 20, 24, 28]
    # remainder layers pile up on the last device, as `min(n-1, i // nums)` does
    a = sharding.layer_assignment(10, 4)
    assert a == [[0, 1], [2, 3], [4, 5], [6, 7, 8, 9]]
    assert sharding.layer_assignment(5, 1) == [[0, 1, 2, 3, 4]]
    # 1M tokens nuq4 + 1 %: 5.07 GB per layer
    assert abs(sharding.kv_bytes_per_layer(4, 1 << 20) / 1e9 - 5.07) < 0.03


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_layers, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    called = []

    def make(i):
        def f(x):
            called.append(i)
            return x * 1.5 + float(i)        # order-sensitive
        return f
    layers = [make(i) for i in range(n_layers)]
    pipe = sharding.LayerShardedPipeline(layers)
    x0 = torch.arange(8, dtype=torch.float32).view(1, 1, 8)
    outs = []
    for step in range(3):
        y = pipe.step(x0 + step, template=x0)
        outs.append(None if y is None else y.clone())
    ret[rank] = (called, outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_layers", [(2, 6), (3, 7)])
def test_pipeline_over_gloo(world, n_layers):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_layers, ret), nprocs=world, join=True)
    x0 = torch.arange(8, dtype=torch.float32).view(1, 1, 8)
    owned = sharding.layer_assignment(n_layers, world)
    for r in range(world):
        called, outs = ret[r]
        assert called == owned[r] * 3, (r, called)          # only its own layers, in order, every step
        for step in range(3):
            if r == 0:
                ref = x0 + step
                for i in range(n_layers):
                    ref = ref * 1.5 + float(i)
                assert torch.equal(outs[step], ref)
            else:
                assert outs[step] is None


def _stream_worker(rank, world, port, n_layers, streams, steps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned = sharding.layer_assignment(n_layers, world)[rank]
    log = []
    # a "layer" with per-stream state (like a KV cache that grows by one token per step): order-sensitive
    state = {(s, i): 0.0 for s in range(streams) for i in owned}

    def stage(s, st, x):
        for i in owned:
            state[(s, i)] += 1.0
            x = x * 1.25 + float(i) + 0.01 * state[(s, i)] + 100.0 * s
        log.append((st, s))
        return x
    pipe = sharding.StreamPipeline(stage, streams)
    x0 = torch.arange(8, dtype=torch.float32).view(1, 1, 8)
    outs = pipe.run(steps, lambda s, st: x0 + s + 10 * st, x0)
    ret[rank] = (log, [o.clone() for o in outs])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_layers,streams", [(2, 6, 2), (3, 7, 3), (3, 6, 1), (2, 4, 5), (4, 8, 4), (8, 32, 8)])
def test_stream_pipeline_over_gloo(world, n_layers, streams):
    """real layer outputs (stateful, order-sensitive) move through the stages; every stream's result equals the
    single-process evaluation and every rank ran every (step, stream) item exactly once, in order"""
    steps = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stream_worker, args=(world, _free_port(), n_layers, streams, steps, ret), nprocs=world, join=True)
    x0 = torch.arange(8, dtype=torch.float32).view(1, 1, 8)
    items = [(st, s) for st in range(steps) for s in range(streams)]
    for r in range(world):
        assert ret[r][0] == items
    outs = ret[0][1]
    assert len(outs) == len(items) and all(len(ret[r][1]) == 0 for r in range(1, world))
    for k, (st, s) in enumerate(items):
        ref = x0 + s + 10 * st
        for i in range(n_layers):
            ref = ref * 1.25 + float(i) + 0.01 * (st + 1) + 100.0 * s
        assert torch.allclose(outs[k], ref, rtol=0, atol=0), (st, s)


def _token_shard_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kvquant_amd import sharding
        H, hd, L = 4, 8, 96
        g = torch.Generator().manual_seed(5)
        s = torch.randn(H, L, generator=g) * 3                 # scaled scores of the whole row
        v = torch.randn(H, L, hd, generator=g)
        per = L // world
        sl = slice(rank * per, (rank + 1) * per if rank < world - 1 else L)

        def shard_fn(record=None):                             # plain-torch stand-in for cache.shard_attention
            ss, vv = s[:, sl], v[:, sl]
            M = ss.max(dim=-1).values
            e = torch.exp(ss - M[:, None])
            Z = e.sum(dim=-1)
            return torch.einsum("hl,hld->hd", e / Z[:, None], vv)[None], M, Z

        out = sharding.token_sharded_step(shard_fn)
        ref = torch.einsum("hl,hld->hd", torch.softmax(s, dim=-1), v)[None]
        q.put((rank, float((out - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_token_sharded_step_merges_exactly(world):
    """the all-gather + combine of sharding.token_sharded_step over gloo: the merged attention of a row split over the
    ranks equals the unsharded softmax . V"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_token_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(err < 1e-5 for _, err in res), res


def _head_shard_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kvquant_amd import sharding
        H, hd = 7, 8                                             # (uneven over 2 and 3 ranks: the gather is padded)
        full = torch.randn(1, H, hd, generator=torch.Generator().manual_seed(11))
        h0, n = sharding.head_assignment(H, world)[rank]
        out = sharding.head_sharded_step(lambda: full[:, h0:h0 + n].clone(), H, hd)
        q.put((rank, bool(torch.equal(out, full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_head_sharded_step_assembles_the_layer_output(world):
    """sharding.head_sharded_step over gloo: every rank's heads land in their rows of the layer's [1, H, hd], uneven
    splits included; head_assignment covers every head once, contiguously"""
    import torch.multiprocessing as mp
    from kvquant_amd import sharding
    for Hn in (32, 7, 8):
        for W in (1, 2, 3, 4, 8):
            sp = sharding.head_assignment(Hn, W)
            assert len(sp) == W and sp[0][0] == 0 and sum(n for _, n in sp) == Hn
            assert all(sp[i][0] + sp[i][1] == sp[i + 1][0] for i in range(W - 1))
            assert max(n for _, n in sp) - min(n for _, n in sp) <= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_head_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


# ---- the op order of StreamPipeline under the strictest p2p semantics --------------------------------
class _RecordingDist:
    """stands in for torch.distributed inside kvquant_amd.sharding: records the point-to-point calls of ONE rank in
    issue order, moves no data"""

    class _Req:
        def wait(self):
            return True

    def __init__(self, rank):
        self.rank, self.ops = rank, []

    def isend(self, t, dst, group=None):
        self.ops.append(("send", dst))
        return self._Req()

    def irecv(self, t, src, group=None):
        self.ops.append(("recv", src))
        return self._Req()

    def send(self, t, dst, group=None):
        self.ops.append(("send", dst))

    def recv(self, t, src, group=None):
        self.ops.append(("recv", src))


def _pipeline_op_order(W, streams, steps, monkeypatch):
    import torch
    from kvquant_amd import sharding
    template = torch.zeros(1, 1, 8)
    per_rank = []
    for r in range(W):
        rec = _RecordingDist(r)
        monkeypatch.setattr(sharding, "dist", rec)
        pipe = sharding.StreamPipeline(lambda s, st, x: x, streams, rank=r, world=W)
        pipe.run(steps, lambda s, st: template.clone(), template)
        per_rank.append(rec.ops)
    return per_rank


def _drains_under_rendezvous(per_rank):
    """RCCL / NCCL run the p2p operations of a communicator in issue order on one stream, and a send finishes only
    against its matching receive.  Strictest model: an operation completes only while it AND its match are at the head of
    their ranks' queues (no buffering at all, the k-th send a -> b matches the k-th receive on b from a).  Returns True
    when every queue drains, i.e. the wait-for graph never has a cycle."""
    heads = [0] * len(per_rank)
    progressed = True
    while progressed:
        progressed = False
        for a, ops in enumerate(per_rank):
            if heads[a] >= len(ops):
                continue
            kind, b = ops[heads[a]]
            if heads[b] >= len(per_rank[b]):
                continue
            kb, peer = per_rank[b][heads[b]]
            if peer == a and kb != kind:       # send meets receive: both complete (in-order per pair by construction)
                heads[a] += 1
                heads[b] += 1
                progressed = True
    return all(h == len(o) for h, o in zip(heads, per_rank))


@pytest.mark.parametrize("W", [2, 3, 4, 8])
def test_stream_pipeline_op_order_cannot_deadlock(W, monkeypatch):
    """The exact isend / irecv / recv sequence that StreamPipeline.run issues on every rank (recorded by running it),
    replayed under zero-buffer rendezvous semantics: it must drain for 1, W and 2W+5 streams (the last exercises the
    bounded send window).  A control with rank 0's final receive of an item posted BEFORE that item's first send (lag 0)
    must NOT drain -- the model can tell the difference."""
    for streams in (1, W, 2 * W + 5):
        per_rank = _pipeline_op_order(W, streams, 3, monkeypatch)
        n_items = 3 * streams
        assert sum(1 for k, _ in per_rank[0] if k == "send") == n_items
        assert _drains_under_rendezvous(per_rank), (W, streams)
    per_rank = _pipeline_op_order(W, W, 3, monkeypatch)
    recvs = [o for o in per_rank[0] if o[0] == "recv"]
    sends = [o for o in per_rank[0] if o[0] == "send"]
    per_rank[0] = [o for pair in zip(recvs, sends) for o in pair]      # receive the final of item i, THEN send item i
    assert not _drains_under_rendezvous(per_rank)

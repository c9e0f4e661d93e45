"""Layer-sharded KV cache across the GPUs of a node.

The reference scales context by placing contiguous chunks of decoder layers --
and therefore their compressed K/V caches -- on different GPUs
(`LlamaModel.set_devices`, ML = modeling_llama.py:2428-2453) and moving the
[1, q_len, hidden] activation with `.to("cuda:k")` at the split points
(ML:2552-2556, 2583-2585) from ONE process.  Here: one process per GPU
(`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm), rank r owns
the layers the reference would put on cuda:r, and the activation (8 KB per hop
at decode) travels by point-to-point send/recv over the direct xGMI link --
there is no collective on the data path; a ring all-reduce of such a payload
would only add latency.  This gives capacity scaling (1M tokens at nuq4 + 1 %
= 5.07 GB per layer: 4 layers = 20 GB per GPU on 8 GPUs), not single-stream
speed-up: the layers of one token are sequential.

The per-layer work is injected as callables, so the routing logic is testable
with the gloo backend on CPU (tests/test_sharding_cpu.py); on GPUs the callables
are the MI355X layers built on `kvquant_amd.attention.KVQuantAttention`.
"""
import torch
import torch.distributed as dist


def layer_device(i, num_layers, world):
    """device of layer i, exactly as ML:2444-2447: `min(world-1, i // (num_layers // world))`."""
    if world <= 1:
        return 0
    nums = max(num_layers // world, 1)
    return min(world - 1, i // nums)


def layer_assignment(num_layers, world):
    """list of owned layer indices per rank."""
    out = [[] for _ in range(max(world, 1))]
    for i in range(num_layers):
        out[layer_device(i, num_layers, world)].append(i)
    return out


def split_indices(num_layers, world):
    """layer indices at which the activation changes device (ML:2443-2449)."""
    return [i for i in range(1, num_layers) if layer_device(i, num_layers, world) != layer_device(i - 1, num_layers, world)]


def kv_bytes_per_layer(bits, tokens, hidden=4096, n_out=42):
    """resident bytes of one layer's K+V cache at `tokens` cached tokens."""
    dense = hidden * bits // 8
    return tokens * (2 * dense + 2 * n_out * 8 + 4 * 2 ** bits)


class LayerShardedPipeline:
    """Runs `layers` (callables hidden -> hidden; only the ones this rank owns are called) as the
    reference's sequential layer pipeline over `world` ranks.

    step(hidden): rank 0 passes the input; every rank returns the final hidden state of the LAST
    layer on rank 0 (ML:2583-2585 moves it back for norm / lm_head) and None elsewhere.
    """

    def __init__(self, layers, rank=None, world=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.num_layers = len(layers)
        self.owned = layer_assignment(self.num_layers, self.world)[self.rank]
        self.layers = layers
        # ranks that actually own layers, in pipeline order
        self.active = [r for r, ls in enumerate(layer_assignment(self.num_layers, self.world)) if ls]

    def _prev_next(self):
        k = self.active.index(self.rank) if self.rank in self.active else -1
        prev_r = self.active[k - 1] if k > 0 else None
        next_r = self.active[k + 1] if 0 <= k < len(self.active) - 1 else None
        return k, prev_r, next_r

    def step(self, hidden, template=None):
        """hidden: the input activation on the first active rank (ignored elsewhere);
        template: a tensor giving shape/dtype/device for receives on the other ranks."""
        if self.world == 1:
            for i in self.owned:
                hidden = self.layers[i](hidden)
            return hidden
        k, prev_r, next_r = self._prev_next()
        first, last = self.active[0], self.active[-1]
        buf = None
        if k >= 0:
            if prev_r is None:
                buf = hidden
            else:
                buf = torch.empty_like(template if template is not None else hidden)
                dist.recv(buf, src=prev_r, group=self.group)
            for i in self.owned:
                buf = self.layers[i](buf)
            if next_r is not None:
                dist.send(buf.contiguous(), dst=next_r, group=self.group)
        # final activation back to the first rank
        if last != first:
            if self.rank == last:
                dist.send(buf.contiguous(), dst=first, group=self.group)
            elif self.rank == first:
                out = torch.empty_like(template if template is not None else hidden)
                dist.recv(out, src=last, group=self.group)
                return out
            return None
        return buf if self.rank == first else None


class StreamPipeline:
    """`streams` independent decode streams kept in flight through the layer-sharded stages, so that every rank
    works on every tick: while rank r runs its layers for stream s, rank r+1 runs its layers for stream s-1
    (classic pipeline parallelism over the reference's layer placement; with streams = 1 it degenerates to the
    reference's sequential hand-over, which is the capacity configuration: one 1M-token stream on 8 GPUs).

    stage(stream, step, x) -> y runs THIS rank's layers of `stream` for decode step `step` on activation x
    ([1, q_len, hidden], 8 KB at decode); the hand-over between ranks is point-to-point (isend / irecv: RCCL over
    the direct xGMI link with the nccl backend, gloo in the CPU tests), there is no collective on the data path.
    The last rank returns each stream's final activation to rank 0 (norm / lm_head live there, ML:2583-2585);
    rank 0 posts those receives `world - 1` items late, when the data is about to arrive, so that the last rank
    never waits for it.
    """

    def __init__(self, stage, streams, rank=None, world=None, group=None):
        self.stage = stage
        self.streams = streams
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world

    def run(self, steps, first_input, template, step0=0):
        """first_input(stream, step) -> activation entering layer 0 (called on rank 0 only).
        Returns on rank 0 the list of final activations in (step, stream) order, elsewhere []."""
        W, r = self.world, self.rank
        items = [(st, s) for st in range(step0, step0 + steps) for s in range(self.streams)]
        if W == 1:
            return [self.stage(s, st, first_input(s, st)) for st, s in items]
        finals, final_reqs, sends = [], [], []
        lag = W - 1

        def post_final():
            buf = torch.empty_like(template)
            finals.append(buf)
            final_reqs.append(dist.irecv(buf, src=W - 1, group=self.group))

        for i, (st, s) in enumerate(items):
            if r == 0:
                if i >= lag:
                    post_final()
                x = first_input(s, st)
            else:
                x = torch.empty_like(template)
                dist.recv(x, src=r - 1, group=self.group)
            y = self.stage(s, st, x).contiguous()
            sends.append((dist.isend(y, dst=(r + 1) % W, group=self.group), y))   # (last rank: back to rank 0)
            if len(sends) > 2 * W + 4:      # keep a bounded number of requests (and their tensors) alive
                sends.pop(0)[0].wait()
        if r == 0:
            while len(finals) < len(items):
                post_final()
            for q in final_reqs:
                q.wait()
        for q, _ in sends:
            q.wait()
        return finals if r == 0 else []


def token_sharded_step(shard_fn, record=None, group=None):
    """One decode step of ONE stream whose context is split along the token axis over the ranks of `group`
    (`kvquant_amd.cache.shard_attention` on each rank's shard): every rank streams only its L / N cached tokens, then
    ONE all-gather of the shard records per layer ([H*hd + 2H] floats = 16.6 KB at the 7B shape; flat over the direct
    xGMI links) carries the shards' outputs and softmax statistics, and every rank forms the exact merged output.
    shard_fn(record) -> (out [1, H, hd], M [H], Z [H]) for this rank's shard.  GPU path: pass `record` (f32
    [cache.shard_record_floats(H, hd)] on the GPU) -- shard_attention writes into it, it is gathered as it is and the
    merge is one library launch (kvq_combine_shards): no torch arithmetic on the data path.  record = None (CPU
    tensors, the gloo tests): the torch formula of cache.combine_shards.  Returns the merged [1, H, hd] on every rank."""
    from . import ops
    from .cache import combine_shards
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    out, M, Z = shard_fn(record)
    H, hd = out.shape[1], out.shape[2]
    if world == 1:
        return out
    if record is not None and record.is_cuda:
        gathered = torch.empty(world * record.numel(), dtype=torch.float32, device=record.device)
        dist.all_gather_into_tensor(gathered, record, group=group)
        merged = torch.empty((1, H, hd), dtype=torch.float32, device=record.device)
        ops.combine_shards(gathered, world, H, hd, merged)
        return merged
    packed = torch.cat((out[0], M[:, None], Z[:, None]), dim=-1).contiguous()          # [H, hd + 2]
    bufs = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(bufs, packed, group=group)
    allp = torch.stack(bufs)                                                            # [R, H, hd + 2]
    return combine_shards(allp[:, None, :, :hd], allp[:, :, hd], allp[:, :, hd + 1])


def head_assignment(num_heads, world):
    """heads of every rank under the head-sharded placement: contiguous, as even as possible (the first num_heads % world
    ranks hold one more) -> [(h0, n)] per rank"""
    base, rem = divmod(num_heads, world)
    out, h = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((h, n))
        h += n
    return out


def head_sharded_step(attend_fn, num_heads, head_dim, group=None, out=None):
    """One decode step of ONE stream whose heads are split over the ranks of `group` (`kvquant_amd.cache.HeadShard` on
    each rank: heads [h0, h0 + n) of the layer, all tokens).  attend_fn() -> f32 [1, n, hd]: the complete, normalised
    attention output of this rank's heads -- heads are independent until o_proj, so there is no softmax merge: ONE
    all-gather of the outputs per layer (8 KB per rank at 4 of 32 heads; uneven splits are padded to the widest shard)
    gives every rank the layer's [1, H, hd].  (In a tensor-parallel model a row-sharded o_proj would consume the local
    slice and all-reduce its output instead; the gather is the form that matches the reference's full-width o_proj.)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = attend_fn()
    if world == 1:
        if out is not None:          # (a caller that reads its own buffer: same contract as with more ranks)
            out.copy_(mine.reshape(out.shape))
            return out
        return mine
    split = head_assignment(num_heads, world)
    widest = max(n for _, n in split)
    rank = dist.get_rank(group)
    send = mine.reshape(-1)
    if split[rank][1] != widest:
        send = torch.cat((send, send.new_zeros((widest - split[rank][1]) * head_dim)))
    send = send.contiguous()
    if send.is_cuda:
        gathered = torch.empty(world * send.numel(), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(gathered, send, group=group)
        parts = gathered.view(world, widest, head_dim)
    else:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(bufs, send, group=group)
        parts = torch.stack(bufs).view(world, widest, head_dim)
    if out is None:
        out = torch.empty((1, num_heads, head_dim), dtype=mine.dtype, device=mine.device)
    for r, (h0, n) in enumerate(split):
        out[0, h0:h0 + n] = parts[r, :n]
    return out

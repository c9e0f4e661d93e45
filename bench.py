#!/usr/bin/env python
"""bench.py -- decode-step throughput of the KVQuant hot path on MI355X.

metric (BASELINE.json): decode tokens/s + KV-matvec GB/s, LLaMA-2-7B head shape
(32 heads x 128), nuq4 + 1 % sparse outliers, 128K cached tokens.

A "step" is one decode step of the hot path over all 32 layers' compressed KV
caches: per layer  K append (NUQ pack + outlier row) -> q.K^T with fused RoPE +
sparse -> /sqrt(d), fp32 softmax -> V append (top-k thresholds, per-token LUT
row, pack, outlier row) -> p.V + sparse.  Inputs (q, k, v per layer: synthetic
fp16 activations) and the caches are resident in HBM before the timed region.
The model's linear layers are NOT part of this path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--ctx 131072] [--bits 4]

N > 1 (torchrun): every rank runs an independent decode stream (its own 32-layer
cache) -- the path has no data-path collective; value = N streams' tokens / max
time over ranks (weak scaling).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, HD, C = 32, 128, 4096
N_LAYERS = 32
THETA = 10000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ctx", type=int, default=131072)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--layers", type=int, default=N_LAYERS)
    ap.add_argument("--sinks", type=int, default=0,
                    help="first_few_fp16 attention-sink tokens kept in fp16 (BASELINE config 3: --bits 3 --sinks 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-tokens", type=int, default=16384)
    return ap.parse_args()


def synth_quantizer(bits, gen, dev):
    """thresholds like SimQuant's 0.5/99.5 percentiles of a per-channel scaled
    normal, centroids = normal quantiles in [-1,1] (tests/golden/gen_golden.py)."""
    scale = torch.exp(0.5 * torch.randn(C, generator=gen, device=dev))
    shift = 0.3 * torch.randn(C, generator=gen, device=dev)
    upper = (shift + 2.576 * scale).cpu().numpy()[None, :]
    lower = (shift - 2.576 * scale).cpu().numpy()[None, :]
    n = 2 ** bits
    p = (torch.arange(n, dtype=torch.float64) + 0.5) / n
    c = torch.special.ndtri(p)
    c = (c / c.abs().max() * 0.97).float().numpy().reshape(n, 1)
    return (upper, lower, [c]), scale, shift


def synth_tokens(S, scale, shift, gen, dev):
    k = torch.randn(S, C, generator=gen, device=dev) * scale + shift
    v = torch.randn(S, C, generator=gen, device=dev)
    for x in (k, v):
        m = torch.rand(S, C, generator=gen, device=dev) < 0.01
        x[m] *= 6.0
    return k.half(), v.half()


class Layer:
    def __init__(self, bits, max_len, gen, dev, sinks=0):
        from kvquant_amd.cache import QuantK, QuantV
        quant, self.scale, self.shift = synth_quantizer(bits, gen, dev)
        self.k = QuantK(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                        include_sparse=True, sparsity_threshold=0.99, rope_theta=THETA, first_few_fp16=sinks,
                        device=dev)
        self.v = QuantV(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                        include_sparse=True, sparsity_threshold=0.99, first_few_fp16=sinks, device=dev)
        self.k.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        self.v.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        self.sinks = sinks
        if sinks:
            # the first tokens stay in fp16 (post-RoPE keys), ML:1464-1466; their scores / outputs are two tiny
            # fp16 matmuls around the compressed path, as in the reference (ML:1950-1962, 1987-1995)
            self.k_sink = (torch.randn(H, HD, sinks, generator=gen, device=dev) * 0.5).half()
            self.v_sink = torch.randn(H, sinks, HD, generator=gen, device=dev).half()
            self.k.klen += sinks
            self.v.vlen += sinks

    def fill(self, ctx, gen, dev, chunk=8192):
        done = 0
        while done < ctx:
            S = min(chunk, ctx - done)
            k, v = synth_tokens(S, self.scale, self.shift, gen, dev)
            self.k.parallel_pack(k.view(S, H, HD).permute(1, 2, 0))
            self.v.parallel_pack(v.view(S, H, HD).permute(1, 2, 0))
            done += S


def decode_step(layers, qs, ks, vs, step):
    """one token through every layer's KV path (GPU-resident: 5 launches per layer, no host sync, fp16
    activations consumed directly); returns the last attention output"""
    from kvquant_amd.cache import decode_kv
    out = None
    for li, lay in enumerate(layers):
        if lay.sinks:
            q = qs[li][step]
            sink_scores = (torch.bmm(q.unsqueeze(1), lay.k_sink) / math.sqrt(HD)).squeeze(1)        # f16 [H, n_sink]
            out, sp = decode_kv(lay.k, lay.v, q, ks[li][step], vs[li][step], sink_scores)
            out = out + torch.bmm(sp.unsqueeze(1), lay.v_sink).transpose(0, 1).float()
            continue
        out, _ = decode_kv(lay.k, lay.v, qs[li][step], ks[li][step], vs[li][step])   # f32 [1,H,hd]
    return out.half()


class KernelTimers:
    """HIP-event timing of the two matvec launches on the launch stream (torch's
    current stream is the stream the C ABI is handed)."""

    def __init__(self):
        self.pairs = {"score_k": [], "mix_v": []}

    def install(self):
        from kvquant_amd import ops
        self._orig = (ops.score_k_prepared_softmax, ops.mix_v)
        pairs = self.pairs

        def wrap(fn, key):
            def inner(*a, **kw):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a, **kw)
                e1.record()
                pairs[key].append((e0, e1))
                return r
            return inner
        ops.score_k_prepared_softmax = wrap(ops.score_k_prepared_softmax, "score_k")   # (+ fused softmax pass 1)
        ops.mix_v = wrap(ops.mix_v, "mix_v")

    def uninstall(self):
        from kvquant_amd import ops
        ops.score_k_prepared_softmax, ops.mix_v = self._orig

    def reset(self):
        for k in self.pairs:
            self.pairs[k].clear()

    def mean_us(self, key):
        p = self.pairs[key]
        if not p:
            return None
        return sum(a.elapsed_time(b) for a, b in p) * 1000.0 / len(p)


def algorithmic_bytes(bits, L, kernel):
    """per cached token per layer, formats fixed by the boundary."""
    n = 2 ** bits
    dense = C * bits // 8
    sparse = 42 * 8
    if kernel == "score_k":
        per_tok = dense + sparse + 4 * H                 # + score write
        extra = H * HD * n * 4 + H * HD * 4              # LUT + q
    else:
        per_tok = dense + sparse + 4 * n + 4 * H         # + codebook row + probability read
        extra = H * HD * 4
    return L * per_tok + extra, per_tok


def cpu_baseline(bits, sample_tokens, ctx, layers):
    """The reference's CPU path = simulated quantisation (quant/kvquant/
    simquant_module_quantizer.py) feeding ordinary fp32 attention.  Timed here: one
    layer's decode step over `sample_tokens` reconstructed tokens with the C oracle
    (OpenMP over the host cores), scaled linearly to ctx tokens x layers."""
    from oracle import ckernels as ck
    g = torch.Generator().manual_seed(0)
    khat = torch.randn(sample_tokens, C, generator=g)
    vhat = torch.randn(sample_tokens, C, generator=g)
    q = torch.randn(C, generator=g)
    ck.sim_decode_step(khat[:256].contiguous(), vhat[:256].contiguous(), q, H, HD, THETA, 0)  # warm
    t0 = time.time()
    reps = 0
    while True:
        ck.sim_decode_step(khat, vhat, q, H, HD, THETA, 0)
        reps += 1
        if time.time() - t0 > 10.0 or reps >= 2000:   # a bounded ~10 s sample of CPU work
            break
    total = time.time() - t0
    dt = total / reps
    step_s = dt * (ctx / sample_tokens) * layers
    return {"value": 1.0 / step_s, "unit": "tokens/s", "cores": ck.num_threads(), "kind": "port",
            "sample": "%d x (1 layer x %d reconstructed tokens: fp32 RoPE+qK^T+softmax+pV, oracle C/OpenMP) = %.1f s of "
                      "CPU work, %.3f s each, scaled x%g tokens x%d layers"
                      % (reps, sample_tokens, total, dt, ctx / sample_tokens, layers)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from kvquant_amd import _lib
    if os.environ.get("KVQ_LIB"):      # development: time an alternative build of the library (tools/abl)
        _lib.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
    _lib.lib()  # fail loudly if the HIP library is missing

    total = args.steps + args.warmup
    max_len = (args.ctx + total + 8 + 63) // 64 * 64
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    t_setup = time.time()
    layers = []
    for li in range(args.layers):
        lay = Layer(args.bits, max_len, gen, dev, args.sinks)
        lay.fill(args.ctx, gen, dev)
        layers.append(lay)
    # per-layer decode inputs, resident before timing
    qs, ks, vs = [], [], []
    for lay in layers:
        k, v = synth_tokens(total, lay.scale, lay.shift, gen, dev)
        q = torch.randn(total, H, HD, generator=gen, device=dev).half()
        qs.append([q[i] for i in range(total)])
        ks.append([k[i] for i in range(total)])
        vs.append([v[i] for i in range(total)])
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    timers = KernelTimers()
    timers.install()
    for s in range(args.warmup):
        decode_step(layers, qs, ks, vs, s)
    timers.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        decode_step(layers, qs, ks, vs, args.warmup + s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timers.uninstall()

    ms_per_step = elapsed * 1000.0 / args.steps
    value = world * args.steps / elapsed                      # decode tokens/s over all streams

    if rank == 0:
        L_mid = args.ctx + args.warmup + args.steps // 2
        k_us, v_us = timers.mean_us("score_k"), timers.mean_us("mix_v")
        dom = "score_k" if (k_us or 0) >= (v_us or 0) else "mix_v"
        dom_us = k_us if dom == "score_k" else v_us
        dom_bytes, per_tok = algorithmic_bytes(args.bits, L_mid, dom)
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        kb, _ = algorithmic_bytes(args.bits, L_mid, "score_k")
        vb, _ = algorithmic_bytes(args.bits, L_mid, "mix_v")
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get(str(args.ctx))
            except Exception:
                traffic = None
        res = {
            "metric": "decode tokens/s, KV-cache hot path (32 layers: NUQ append + q.K^T(RoPE)+sparse + softmax + p.V+sparse), "
                      "LLaMA-2-7B head shape, nuq%d 1%%-sparse @%dK ctx" % (args.bits, args.ctx // 1024),
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LLaMA-2-7B KV path: H=32 hd=128 layers=%d nuq%d + 1%% outliers (42/token), "
                                   "ctx=%d cached tokens%s, batch 1 per GPU"
                                   % (args.layers, args.bits, args.ctx,
                                      " + %d fp16 attention-sink tokens" % args.sinks if args.sinks else ""),
                       "ctx": args.ctx, "bits": args.bits, "layers": args.layers, "sinks": args.sinks,
                       "parallelism": "independent decode streams x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "avg_launch_us": dom_us, "algorithmic_bytes_per_launch": dom_bytes,
                         "bytes_per_token": per_tok},
            "kernels": {"score_k_us": k_us, "mix_v_us": v_us,
                        "score_k_GBps": kb / (k_us * 1e-6) / 1e9 if k_us else None,
                        "mix_v_GBps": vb / (v_us * 1e-6) / 1e9 if v_us else None,
                        "kv_matvec_GBps": (kb + vb) / ((k_us + v_us) * 1e-6) / 1e9 if k_us and v_us else None},
            "setup_s": t_setup,
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.bits, args.cpu_sample_tokens, args.ctx, args.layers)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

  python bench.py [--gpus N] [--steps K] [--warmup W] [--ctx 131072] [--bits 4] [--sinks 0]
  python bench.py --sweep          # one JSON line per BASELINE configuration (4K hot/rotated, 32K, 128K, 1M;
                                   # nuq4 and nuq3 + 5 sinks), the default line last

N > 1 (torchrun, one rank per GPU): the caches are SHARDED BY LAYER over the ranks exactly like the reference's
set_devices (modeling_llama.py:2428-2453) and N decode streams are kept in flight through the N pipeline stages;
the 8 KB activation moves between neighbouring ranks point-to-point over RCCL (no collective on the data path).
Per-GPU work is fixed as N grows (32/N layers x N streams x ctx): weak scaling, value = all streams' tokens per
second.  --streams 1 is the reference's capacity mode (one long-context stream); --replicas keeps the old
"N independent 32-layer copies" mode.  --shard tokens: ONE stream whose context is split along the token axis (every
rank streams ctx / N tokens of every layer, one all-gather of [H*hd + 2H] floats per layer merges the shards exactly):
the placement that speeds a single long stream up; strong scaling.
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
TRAFFIC_PROFILE = "profiles/pmc_traffic.json"


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
    ap.add_argument("--compact", action="store_true",
                    help="opt-in compact outlier formats (fp16 residual + 16-bit channel: 168 instead of 336 B per token "
                         "and matvec; NOT the reference's format, reported with its own algorithmic bytes)")
    ap.add_argument("--score-f16", action="store_true",
                    help="3 bit: q.K^T through the opt-in fp16 pair-sum tables (QuantK.score_f16_pair; scores within 4e-4 of "
                         "the reference's, outputs within 4e-3 -- not the reference's rounding, hence not the default)")
    ap.add_argument("--streams", type=int, default=0, help="decode streams in flight (default: one per rank)")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent full copies instead of layer sharding")
    ap.add_argument("--shard", choices=("layers", "tokens", "heads"), default="layers",
                    help="N > 1: layers = the reference's placement, N streams pipelined through it (default); tokens = ONE "
                         "stream whose context is split along the token axis (strong scaling, one all-gather per layer); "
                         "heads = ONE stream, every rank holds H / N heads of every layer (strong scaling, one all-gather "
                         "of the heads' outputs per layer)")
    ap.add_argument("--prefill", action="store_true",
                    help="BASELINE config 4 only: fused prefill pack (K, V) of an 8192-token prompt + the causal MFMA attention")
    ap.add_argument("--sweep", action="store_true", help="every BASELINE configuration, one JSON line each (1 GPU)")
    ap.add_argument("--retrieval", action="store_true", help="plant a retrievable token and check it (config 5 proxy)")
    ap.add_argument("--time-every", type=int, default=11,
                    help="bracket every N-th layer's two matvec launches with HIP events (1 = all; the events cost time)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short legs over the other BASELINE configurations that the default 1-GPU line carries "
                         "under `configs` (4K / 32K decode, nuq3 + 5 sinks at 128K, the 8K prefill, 1M tokens on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp16-baseline", action="store_true")
    ap.add_argument("--no-full-model", action="store_true",
                    help="skip the full-model leg (random-init LLaMA-2-7B-shaped fp16 weights over the same filled caches)")
    ap.add_argument("--cpu-sample-tokens", type=int, default=8192)
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
    # (no boolean-mask indexing: it syncs with the host on every call, which is what made the 4-process one-GPU smoke run
    #  of round 3 look hung -- DESIGN.md 6)
    k = torch.where(torch.rand(S, C, generator=gen, device=dev) < 0.01, k * 6.0, k)
    v = torch.where(torch.rand(S, C, generator=gen, device=dev) < 0.01, v * 6.0, v)
    return k.half(), v.half()


def rope_rotate(x, pos, sign=1.0):
    """RoPE (rotate-half convention of the reference, ML:180-205) of x [H, hd] at position pos; sign = -1 inverts"""
    inv = 1.0 / (THETA ** (torch.arange(0, HD, 2, device=x.device, dtype=torch.float32) / HD))
    ang = torch.cat((inv, inv)) * float(pos) * sign
    x = x.float()
    rot = torch.cat((-x[..., HD // 2:], x[..., :HD // 2]), dim=-1)
    return x * ang.cos() + rot * ang.sin()


class Layer:
    def __init__(self, bits, max_len, gen, dev, sinks=0, compact=False):
        from kvquant_amd.cache import QuantK, QuantV
        quant, self.scale, self.shift = synth_quantizer(bits, gen, dev)
        self.k = QuantK(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                        include_sparse=True, sparsity_threshold=0.99, rope_theta=THETA, first_few_fp16=sinks,
                        device=dev, compact=compact)
        self.v = QuantV(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                        include_sparse=True, sparsity_threshold=0.99, first_few_fp16=sinks, device=dev, compact=compact)
        self.k.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        self.v.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
        self.sinks = sinks
        self.planted = None
        if sinks:
            # the first tokens stay in fp16 (post-RoPE keys), ML:1464-1466; their scores / outputs are two tiny
            # fp16 matmuls around the compressed path, as in the reference (ML:1950-1962, 1987-1995)
            self.k_sink = (torch.randn(H, HD, sinks, generator=gen, device=dev) * 0.5).half()
            self.v_sink = torch.randn(H, sinks, HD, generator=gen, device=dev).half()
            self.k.klen += sinks
            self.v.vlen += sinks

    def fill(self, ctx, gen, dev, chunk=8192, plant=None):
        """prefill-pack ctx synthetic tokens; plant = (position, key [C], value [C]) replaces one of them"""
        done = 0
        while done < ctx:
            S = min(chunk, ctx - done)
            k, v = synth_tokens(S, self.scale, self.shift, gen, dev)
            if plant is not None and done <= plant[0] < done + S:
                k[plant[0] - done] = plant[1].half()
                v[plant[0] - done] = plant[2].half()
                self.planted = plant[0]
            self.k.parallel_pack(k.view(S, H, HD).permute(1, 2, 0))
            self.v.parallel_pack(v.view(S, H, HD).permute(1, 2, 0))
            done += S


def layer_step(lay, q, k, v):
    """one token through one layer's KV path (GPU-resident: 5 launches, no host sync, fp16 activations consumed
    directly); returns the attention output f32 [1, H, hd]"""
    from kvquant_amd.cache import decode_kv
    if lay.sinks:
        # the fp16 sink tokens ride in the same launches (scores: prologue; their share of the output: softmax pass)
        out, _ = decode_kv(lay.k, lay.v, q, k, v, k_sink=lay.k_sink, v_sink=lay.v_sink)
        return out
    out, _ = decode_kv(lay.k, lay.v, q, k, v)
    return out


class KernelTimers:
    """HIP-event timing of the two matvec launches inside the timed region, on the stream they are launched on.
    decode_kv is ONE library call per layer (kvq_decode_step), so the events are recorded by the library itself
    (kvq_decode_step_events: before / after the q.K^T launch, before / after the p.V kernel + slab reduce), created
    and read here through the HIP runtime."""

    def __init__(self, every=1):
        import ctypes
        self.ct = ctypes
        self.every = every            # time every N-th layer call (N coprime with the layer count walks through all layers)
        self.calls = 0
        self.first = True
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.quads = []
        self.pool = []
        self.pairs = {"score_k": [], "mix_v": []}      # the five-call path (KVQ_DECODE_MULTICALL=1): torch events

    def _quad(self):
        ct = self.ct
        if self.pool:
            return self.pool.pop()
        arr = (ct.c_void_p * 4)()
        for i in range(4):
            e = ct.c_void_p()
            if self.hip.hipEventCreate(ct.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")
            arr[i] = e
        return arr

    def install(self):
        from kvquant_amd import _lib, ops
        self._orig = ops.decode_step
        self._orig_multi = (ops.score_k_prepared_softmax, ops.mix_v, ops.mix_v_softmax)
        lib = _lib.lib()
        me = self

        def timed(layer, *a, **kw):
            # every `self.every`-th decode_step call is bracketed (4 event records on the launch stream are not free:
            # all 32 layers of a step timed cost 0.1 - 0.4 ms per step)
            me.calls += 1
            if me.first or (me.every and me.calls % me.every == 0):      # (at least one sample per timed region)
                me.first = False
                q = me._quad()
                lib.kvq_decode_step_events(q)
                me.quads.append(q)
            return me._orig(layer, *a, **kw)
        ops.decode_step = timed

        def wrap(fn, key):
            def inner(*a, **kw):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a, **kw)
                e1.record()
                me.pairs[key].append((e0, e1))
                return r
            return inner
        ops.score_k_prepared_softmax = wrap(ops.score_k_prepared_softmax, "score_k")   # (+ fused softmax pass 1)
        ops.mix_v = wrap(ops.mix_v, "mix_v")
        ops.mix_v_softmax = wrap(ops.mix_v_softmax, "mix_v")             # (+ second softmax pass, + slab reduce)

    def uninstall(self):
        from kvquant_amd import ops
        ops.decode_step = self._orig
        ops.score_k_prepared_softmax, ops.mix_v, ops.mix_v_softmax = self._orig_multi

    def reset(self):
        torch.cuda.synchronize()
        self.pool.extend(self.quads)
        self.quads = []
        self.first = True
        for k in self.pairs:
            self.pairs[k].clear()

    def mean_us(self, key):
        if not self.quads:
            p = self.pairs[key]
            return sum(a.elapsed_time(b) for a, b in p) * 1000.0 / len(p) if p else None
        a, b = (0, 1) if key == "score_k" else (2, 3)
        ms = self.ct.c_float()
        tot = 0.0
        for q in self.quads:
            if self.hip.hipEventElapsedTime(self.ct.byref(ms), q[a], q[b]) != 0:
                raise RuntimeError("hipEventElapsedTime failed")
            tot += ms.value
        return tot * 1000.0 / len(self.quads)


def algorithmic_bytes(bits, L, kernel, compact=False):
    """per cached token per layer, formats fixed by the boundary (compact: the opt-in 4-byte entries)."""
    n = 2 ** bits
    dense = C * bits // 8
    sparse = 42 * (4 if compact else 8)
    if kernel == "score_k":
        per_tok = dense + sparse + 4 * H                 # + score write
        extra = H * HD * n * 4 + H * HD * 4              # LUT + q
    else:
        per_tok = dense + sparse + 4 * n + 4 * H         # + codebook row + probability read
        extra = H * HD * 4
    return L * per_tok + extra, per_tok


def _torch_attention_step(khat, vhat, q):
    """the reference's CPU formulation of a decode step over reconstructed K / V (what quant/llama_simquant.py's model
    runs after QuantLinearSim: stock HF attention in fp32): rotary embedding of the cached keys at their positions
    (rotate_half convention, ML:180-205), q.K^T / sqrt(d), softmax, p.V.  khat / vhat: [L, C] fp32, q: [C]."""
    L = khat.shape[0]
    k = khat.view(L, H, HD).transpose(0, 1)                                   # [H, L, hd]
    v = vhat.view(L, H, HD).transpose(0, 1)
    inv = 1.0 / (THETA ** (torch.arange(0, HD, 2, dtype=torch.float32) / HD))
    ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = torch.cat((ang.cos(), ang.cos()), dim=-1), torch.cat((ang.sin(), ang.sin()), dim=-1)
    rot = torch.cat((-k[..., HD // 2:], k[..., :HD // 2]), dim=-1)
    kr = k * cos + rot * sin
    s = torch.matmul(q.view(H, 1, HD), kr.transpose(1, 2)) / math.sqrt(HD)    # [H, 1, L]
    p = torch.softmax(s, dim=-1, dtype=torch.float32)
    return torch.matmul(p, v)                                                 # [H, 1, hd]


def cpu_baseline(bits, sample_tokens, ctx, layers):
    """The reference's CPU path = simulated quantisation (quant/kvquant/simquant_module_quantizer.py: QuantLinearSim
    fake-quantises the k_proj / v_proj outputs) feeding ordinary fp32 attention.  Legs, all on the host cores, bounded
    samples scaled linearly:
      quantize  -- the reference's own torch formulation (restated in oracle/simquant.py, bit-exact vs the reference
                   module): per-channel capped-outlier K and per-token dynamic V fake-quant of a block of tokens; a
                   decode step quantizes ONE new token per layer, a prefill all of them;
      attention -- the reference's torch-CPU formulation (fp32 RoPE + q.K^T + softmax + p.V over the reconstructed
                   tokens of one layer, stock-HF arithmetic) at the best of a thread sweep, and the same arithmetic as a
                   C / OpenMP port on all host cores; `value` is built from the FASTER of the two (`value_leg`)."""
    from oracle import ckernels as ck
    from oracle import simquant as sq
    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    # ---- thread sweep of the torch legs (torch's CPU kernels lose throughput to oversubscription on 100+ core hosts)
    khat = torch.randn(sample_tokens, C, generator=g)
    vhat = torch.randn(sample_tokens, C, generator=g)
    q = torch.randn(C, generator=g)
    sweep = {}
    cands = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores})
    for t in cands:
        torch.set_num_threads(t)
        _torch_attention_step(khat[:1024], vhat[:1024], q)
        t0 = time.time()
        n = 0
        while time.time() - t0 < 1.0:
            _torch_attention_step(khat, vhat, q)
            n += 1
        sweep[t] = (time.time() - t0) / n
    athreads = min(sweep, key=sweep.get)
    torch.set_num_threads(athreads)
    t0 = time.time()
    reps = 0
    while True:
        _torch_attention_step(khat, vhat, q)
        reps += 1
        if time.time() - t0 > 6.0 or reps >= 2000:
            break
    a_total = time.time() - t0
    dt_torch = a_total / reps
    # ---- quantize leg
    qthreads = min(cores, 16)
    torch.set_num_threads(qthreads)
    n = 2 ** bits
    scale = torch.exp(0.5 * torch.randn(C, generator=g))
    quant = ((2.576 * scale).numpy()[None], (-2.576 * scale).numpy()[None],
             [torch.sort(torch.rand(n, generator=g) * 2 - 1).values.numpy().reshape(n, 1)])
    blk = 512
    kb = (torch.randn(blk, C, generator=g) * scale).half()
    vb = torch.randn(blk, C, generator=g).half()
    sq.fake_quant_k(kb[:8], quant, bits)
    t0 = time.time()
    qreps = 0
    while True:
        sq.fake_quant_k(kb, quant, bits, cap_outliers=21)
        sq.fake_quant_v(vb, quant, bits, sparsity_threshold=0.99)
        qreps += 1
        if time.time() - t0 > 6.0 or qreps >= 200:
            break
    q_total = time.time() - t0
    prefill_tok_s = qreps * blk / q_total / layers           # tokens/s through all layers' K and V
    t1 = time.time()
    one = 0
    while True:
        sq.fake_quant_k(kb[:1], quant, bits, cap_outliers=21)
        sq.fake_quant_v(vb[:1], quant, bits, sparsity_threshold=0.99)
        one += 1
        if time.time() - t1 > 2.0 or one >= 2000:
            break
    q_one = (time.time() - t1) / one                          # one new token, one layer
    # ---- attention leg, C / OpenMP port (all host cores)
    ck.sim_decode_step(khat[:256].contiguous(), vhat[:256].contiguous(), q, H, HD, THETA, 0)  # warm
    t0 = time.time()
    creps = 0
    while True:
        ck.sim_decode_step(khat, vhat, q, H, HD, THETA, 0)
        creps += 1
        if time.time() - t0 > 5.0 or creps >= 2000:
            break
    c_total = time.time() - t0
    dt_c = c_total / creps
    step_torch = (dt_torch * (ctx / sample_tokens) + q_one) * layers
    step_c = (dt_c * (ctx / sample_tokens) + q_one) * layers
    # `value` = the FASTER of the two attention legs (the C / OpenMP port is ~16x faster than torch's CPU kernels
    # on this host; rounds 1-2 reported the C port, round 3 the torch leg -- the leg is named, both are kept)
    use_c = step_c < step_torch
    return {"value": 1.0 / min(step_c, step_torch), "unit": "tokens/s", "cores": ck.num_threads() if use_c else athreads,
            "host_cores": cores, "kind": "port", "value_leg": "c_port" if use_c else "torch",
            "definition": "simulated-quant CPU path (a PORT: oracle/simquant.py + attention restated), decode step at the full "
                          "context: value = 1 / (layers x (attention over ctx tokens [faster leg] + fake-quant of one new token))",
            "attention_thread_sweep_s": {str(k): round(v, 4) for k, v in sweep.items()},
            "torch_leg": {"value": 1.0 / step_torch, "cores": athreads, "seconds_per_sample": dt_torch},
            "c_port": {"value": 1.0 / step_c, "cores": ck.num_threads(), "seconds_per_sample": dt_c},
            "prefill_quantize_tokens_per_s": prefill_tok_s,
            "sample": "attention (value): %d x (1 layer x %d reconstructed tokens: fp32 RoPE + q.K^T + softmax + p.V in the "
                      "reference's torch-CPU formulation on %d threads, the best of the sweep %s over %d host cores) = %.1f s, "
                      "%.3f s each, scaled x%g tokens x%d layers; the same arithmetic as a C / OpenMP port on %d threads: %.3f s "
                      "each; quantize: %d x (%d-token block, per-channel capped K + per-token dynamic V fake-quant, torch CPU on "
                      "%d threads = the reference's formulation) = %.1f s -> %.0f prompt tokens/s through %d layers; one new "
                      "token per layer per decode step = %.2f ms"
                      % (reps, sample_tokens, athreads, cands, cores, a_total, dt_torch, ctx / sample_tokens, layers,
                         ck.num_threads(), dt_c, qreps, blk, qthreads, q_total, prefill_tok_s, layers, q_one * 1e3)}


def full_model_leg(args, caches, owned, max_len, dev, tokens=8):
    """BASELINE.md's "full model" column (deployment/llama.py:39-94: `benchmark()` feeds one token at a time through the
    whole network): a random-init fp16 Llama with the LLaMA-2-7B shape (4096 / 11008 / 32 heads, 13.5 GB of weights for 32
    layers) patched by kvquant_amd.llama, its attention modules given THE caches this run filled (stream 0), decoding
    `tokens` tokens at the bench's context.  Weights are random (no checkpoint offline): the time per token is what a
    real checkpoint would take, the tokens mean nothing.  Returns a dict for the JSON line."""
    import time as _t
    from transformers import LlamaConfig, LlamaForCausalLM
    from kvquant_amd import llama as kl
    t0 = _t.time()
    cfg = LlamaConfig(vocab_size=32000, hidden_size=C, intermediate_size=11008, num_hidden_layers=len(owned),
                      num_attention_heads=H, num_key_value_heads=H, max_position_embeddings=max(max_len, 4096),
                      attention_bias=False, tie_word_embeddings=False)
    kl.kvquant_config(cfg, maxseqlen=64, abits=args.bits, include_sparse=True, first_few_fp16=args.sinks)
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(torch.float32)
    model.eval()
    kl.patch_llama(model, compact=getattr(args, "compact", False))
    for i, layer in enumerate(model.model.layers):
        lay = caches[(0, owned[i])]
        attn, core = layer.self_attn, layer.self_attn.kvq
        core.kcache, core.vcache = lay.k, lay.v
        object.__setattr__(attn, "kcache", lay.k)
        object.__setattr__(attn, "vcache", lay.v)
        if args.sinks:
            core.kcache_fp16, core.vcache_fp16 = lay.k_sink.unsqueeze(0), lay.v_sink.unsqueeze(0)
    weights_gb = sum(p.numel() * p.element_size() for p in model.parameters()) / 1e9
    ids = torch.randint(0, 32000, (1, tokens + 2), device=dev)
    times = []
    with torch.no_grad():
        for i in range(tokens + 2):
            torch.cuda.synchronize()
            t1 = _t.perf_counter()
            model(ids[:, i:i + 1], use_cache=False)
            torch.cuda.synchronize()
            times.append(_t.perf_counter() - t1)
    times = sorted(times[2:])
    med = times[len(times) // 2]
    del model
    torch.cuda.empty_cache()
    return {"tokens_per_s": 1.0 / med, "ms_per_token": med * 1e3, "tokens_timed": tokens, "layers": len(owned),
            "weights_GB": weights_gb, "ctx": args.ctx,
            "note": "random-init fp16 LLaMA-2-7B-shaped model (stock HF Llama + kvquant_amd.llama patch), token-by-token "
                    "decode over the caches of this run (deployment/llama.py:72-88); median of %d steps; includes the "
                    "projections / MLP weight stream and the Python per-layer overhead of the HF module tree" % tokens,
            "setup_s": _t.time() - t0}


def fp16_matvec_baseline(ctx, dev, iters=10):
    """The un-quantised baseline of the reference's kernel benchmarks (benchmarking/scripts/test_kernel_baselines.py:
    28-61): fp16 K / V of one layer, torch.matmul for q.K^T and p.V (rocBLAS batched GEMV), two copies alternating."""
    try:
        ks = [torch.randn(H, ctx, HD, device=dev, dtype=torch.float16) for _ in range(2)]
        vs = [torch.randn(H, ctx, HD, device=dev, dtype=torch.float16) for _ in range(2)]
    except torch.cuda.OutOfMemoryError:
        return None
    q = torch.randn(H, 1, HD, device=dev, dtype=torch.float16)
    p = torch.softmax(torch.randn(H, 1, ctx, device=dev), dim=-1).half()
    res = {}
    for name, fn in (("qk", lambda i: torch.matmul(q, ks[i % 2].transpose(1, 2))), ("pv", lambda i: torch.matmul(p, vs[i % 2]))):
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        res[name + "_us"] = e0.elapsed_time(e1) * 1000.0 / iters
    byts = ctx * C * 2
    res["bytes_per_token"] = 2 * C * 2
    res["qk_GBps"] = byts / res["qk_us"] / 1e3
    res["pv_GBps"] = byts / res["pv_us"] / 1e3
    del ks, vs
    return res


def check_retrieval(lay, q, dev):
    """SURVEY 8(d) config 5 proxy for the passkey evaluation: one cached token was planted whose key is aligned with
    the (de-rotated) query.  Per head: the softmax mass the decode step gives that token (from the path's own
    probabilities) and, where the mass is ~1, the attention output against that token's dequantised value (the p.V
    kernel on a one-hot probability vector).  Retrieved = mass > 0.999; the synthetic keys are heavy-tailed
    (1 % of the entries x6), so a few heads may see a competitor."""
    from kvquant_amd import ops
    from kvquant_amd.cache import decode_kv
    kc, vc = lay.k, lay.v
    L = kc.klen - kc.first_few_fp16
    g = torch.Generator(device=dev).manual_seed(99)
    knew = torch.zeros(C, device=dev, dtype=torch.float16)
    vnew = torch.randn(C, device=dev, dtype=torch.float16, generator=g)
    out, _ = decode_kv(kc, vc, q, knew, vnew)                          # appends one more token (score ~ 0)
    s = torch.zeros(1, H, L + 1, device=dev)
    ops.score_k(kc.bits, q.float().unsqueeze(0).contiguous(), kc.kcache, s, kc.lookup_table, L + 1, THETA,
                kc.first_few_fp16, kc.outliers, kc.outlier_indices, accumulate=False)
    probs, _ = ops.softmax_scale(s[0], 1.0 / math.sqrt(HD))
    mass = probs[:, lay.planted]
    onehot = torch.zeros(1, H, L + 1, device=dev)
    onehot[:, :, lay.planted] = 1.0
    ref = torch.empty(1, H, HD, device=dev)
    ops.mix_v(vc.bits, onehot, vc.vcache, ref, vc.mix_table(), L + 1, vc.outliers, vc.outlier_indices, accumulate=False)
    hit = mass > 0.999
    err = float(((out - ref)[0][hit].abs().amax() / ref[0][hit].abs().amax())) if bool(hit.any()) else None
    top1 = int((probs.argmax(dim=-1) == lay.planted).sum())
    return {"planted_at": lay.planted, "context": L + 1, "heads_top1": top1, "heads_mass_gt_0.999": int(hit.sum()),
            "min_mass": float(mass.min()), "max_rel_err_vs_dequantised_value": err,
            "ok": top1 >= (3 * H) // 4 and err is not None and err < 5e-3}


def run_token_sharded(args, rank, world, dev, dist):
    """--shard tokens: ONE decode stream, every rank holds ctx / N cached tokens of every layer
    (kvquant_amd.cache.shard_attention), the newest tokens live on the last rank; per layer one all-gather of
    [H, hd + 2] floats merges the shards exactly (sharding.token_sharded_step).  Strong scaling: the total work is
    fixed as N grows.  Returns the result dict on rank 0."""
    from kvquant_amd import sharding
    from kvquant_amd.cache import shard_attention
    total = args.steps + args.warmup
    per = (args.ctx + world - 1) // world
    lo = rank * per
    n_here = max(0, min(args.ctx, lo + per) - lo)
    last = rank == world - 1
    max_len = (n_here + (total if last else 0) + 8 + 63) // 64 * 64
    gen = torch.Generator(device=dev).manual_seed(1234)          # the same queries / new tokens on every rank
    t_setup = time.time()
    layers = []
    for li in range(args.layers):
        lay = Layer(args.bits, max_len, gen, dev, 0)
        lay.fill(n_here, torch.Generator(device=dev).manual_seed(77 + 1000 * rank + li), dev)
        k, v = synth_tokens(total, lay.scale, lay.shift, gen, dev)
        q = torch.randn(total, H, HD, generator=gen, device=dev).half()
        layers.append((lay, q, k, v))
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    from kvquant_amd.cache import shard_record_floats
    record = torch.empty(shard_record_floats(H, HD), dtype=torch.float32, device=dev)

    def step(st):
        out = None
        for lay, q, k, v in layers:
            qq = q[st] if out is None else q[st] + out.view(H, HD).half() * 1e-3      # (a true dependency on the merge)
            fn = (lambda rec: shard_attention(lay.k, lay.v, qq, k[st], v[st], pos_base=lo, record=rec)) if last else \
                (lambda rec: shard_attention(lay.k, lay.v, qq, pos_base=lo, record=rec))
            out = sharding.token_sharded_step(fn, record)
        return out

    for st in range(args.warmup):
        step(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in range(args.warmup, total):
        step(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    kb, per_tok = algorithmic_bytes(args.bits, n_here, "score_k")
    vb, _ = algorithmic_bytes(args.bits, n_here, "mix_v")
    return {
        "metric": "decode tokens/s, KV-cache hot path (%d layers), LLaMA-2-7B head shape, nuq%d 1%%-sparse @%dK ctx, ONE "
                  "stream, context split along the token axis over %d GPU(s)" % (args.layers, args.bits, args.ctx // 1024, world),
        "value": args.steps / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1000.0 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LLaMA-2-7B KV path: H=32 hd=128 layers=%d nuq%d + 1%% outliers, ctx=%d cached tokens, 1 stream"
                               % (args.layers, args.bits, args.ctx),
                   "ctx": args.ctx, "bits": args.bits, "layers": args.layers, "streams": 1,
                   "parallelism": "token-sharded over %d GPU(s): %d tokens per GPU, one all-gather of [H, hd + 2] f32 per layer"
                                  % (world, per)},
        "roofline": {"bound": "hbm", "kernel": "score_k + mix_v per GPU", "achieved": args.layers * (kb + vb) / (elapsed / args.steps) / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": args.layers * (kb + vb) / (elapsed / args.steps) / 1e9 / 8000.0,
                     "traffic": None, "bytes_per_token": per_tok},
        "cpu_baseline": None, "setup_s": t_setup,
    }


def run_head_sharded(args, rank, world, dev, dist):
    """--shard heads: ONE decode stream, every rank holds H / N heads of every layer for all ctx tokens
    (kvquant_amd.cache.HeadShard: the whole new token is appended into a full-width staging column on every rank --
    the outlier selection is a property of the whole token --, the rank's heads are extracted, attended with the ordinary
    kernels at H / N heads, and one all-gather per layer assembles [H, hd]: sharding.head_sharded_step).  Strong scaling.
    Every rank draws the same synthetic tokens (a tensor-parallel model would all-gather the k / v slices instead)."""
    from kvquant_amd import sharding
    from kvquant_amd.cache import HeadShard
    total = args.steps + args.warmup
    max_len = (args.ctx + total + 8 + 63) // 64 * 64
    h0, n = sharding.head_assignment(H, world)[rank]
    if n == 0:
        raise SystemExit("bench.py --shard heads: more ranks (%d) than heads (%d)" % (world, H))
    gen = torch.Generator(device=dev).manual_seed(1234)          # the same quantizers / tokens / queries on every rank
    t_setup = time.time()
    layers = []
    # (priced at full width: a shard's outlier rows and codebook rows keep the token's width, only the packed words shrink)
    check_memory(args, args.layers, 1, max_len, dev,
                 "rank %d of %d (heads %d..%d of every layer, ctx %d, priced at full width)" % (rank, world, h0, h0 + n - 1, args.ctx))
    stage = None          # ONE set of staging buffers for all layers of the rank (the tables stay per layer)
    sinks = args.sinks
    for li in range(args.layers):
        quant, scale, shift = synth_quantizer(args.bits, gen, dev)
        hs = HeadShard(args.bits, C, H, (h0, n), max_len, rope_theta=THETA, device=dev, stage_len=8192,
                       first_few_fp16=sinks, staging=stage)
        if stage is None:
            stage = hs
        hs.load_lookup_table(quant, quant)
        fill = torch.Generator(device=dev).manual_seed(77 + li)
        done = 0
        while done < args.ctx:
            S = min(8192, args.ctx - done)
            k, v = synth_tokens(S, scale, shift, fill, dev)
            hs.pack(k.view(S, H, HD).permute(1, 2, 0).float(), v.view(S, H, HD).permute(1, 2, 0).float())
            done += S
        k, v = synth_tokens(total, scale, shift, gen, dev)
        # (fp32 activations here: the dependency of a layer's query on the previous layer's gathered output is then ONE
        #  torch launch per layer -- the library takes fp16 or fp32 alike)
        k, v = k.float(), v.float()
        q = torch.randn(total, H, HD, generator=gen, device=dev)
        ks = (torch.randn(H, HD, sinks, generator=gen, device=dev) * 0.5).half()[h0:h0 + n].contiguous() if sinks else None
        vs = torch.randn(H, sinks, HD, generator=gen, device=dev).half()[h0:h0 + n].contiguous() if sinks else None
        layers.append((hs, q, k, v, ks, vs))
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    gathered = torch.empty((1, H, HD), dtype=torch.float32, device=dev)

    def step(st):
        out = None
        for hs, q, k, v, ks, vs in layers:
            qq = q[st] if out is None else torch.add(q[st], out.view(H, HD), alpha=1e-3)   # (a true dependency on the gather)
            out = sharding.head_sharded_step(lambda: hs.attend(qq, k[st], v[st], k_sink=ks, v_sink=vs), H, HD, out=gathered)
        return out

    for st in range(args.warmup):
        step(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in range(args.warmup, total):
        step(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    # a rank's algorithmic bytes per cached token and layer: its heads' share of the packed words, scores and
    # probabilities -- and the WHOLE 42-wide outlier rows and codebook row (a shard's rows keep the token's width)
    dense = C * args.bits // 8 * n // H
    per_tok = (dense + 336 + 4 * n) + (dense + 336 + 4 * 2 ** args.bits + 4 * n)
    achieved = args.layers * args.ctx * per_tok / (elapsed / args.steps) / 1e9
    return {
        "metric": "decode tokens/s, KV-cache hot path (%d layers), LLaMA-2-7B head shape, nuq%d 1%%-sparse @%dK ctx, ONE "
                  "stream, heads split over %d GPU(s)" % (args.layers, args.bits, args.ctx // 1024, world),
        "value": args.steps / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1000.0 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LLaMA-2-7B KV path: H=32 hd=128 layers=%d nuq%d + 1%% outliers, ctx=%d cached tokens, 1 stream"
                               % (args.layers, args.bits, args.ctx),
                   "ctx": args.ctx, "bits": args.bits, "layers": args.layers, "streams": 1,
                   "parallelism": "head-sharded over %d GPU(s): %d of %d heads per GPU, whole-token append into a staging "
                                  "column + extract, one all-gather of [heads, hd] f32 per layer" % (world, n, H)},
        "roofline": {"bound": "hbm", "kernel": "score_k + mix_v per GPU (its heads' share of the bytes)", "achieved": achieved,
                     "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None, "bytes_per_token": per_tok},
        "cpu_baseline": None, "setup_s": t_setup,
    }


def run_prefill_config(bits, S, dev, iters=10):
    """BASELINE config 4 as a driver-visible line: the fused prefill pack of an S-token prompt (one layer's K and V:
    selection, codebook rows, codes, outlier rows + mirror -- kvq_pack_{k,v}_fused, what QuantK / QuantV.parallel_pack
    run) and the causal prompt attention on the matrix cores (kvq_prefill_attention).  Pack bytes: the fp32 prompt read
    once + the packed words, rows and mirror written; attention: 4 H S^2 d / 2 flops against the 2.5 PFLOP/s dense fp16 peak."""
    from kvquant_amd import ops
    from kvquant_amd.cache import QuantK, QuantV
    gen = torch.Generator(device=dev).manual_seed(4321)
    quant, scale, shift = synth_quantizer(bits, gen, dev)
    max_len = S + 64
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, device=dev)
    kc, vc = QuantK(rope_theta=THETA, **kw), QuantV(**kw)
    for c in (kc, vc):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    k, v = synth_tokens(S, scale, shift, gen, dev)
    k = k.float().t().reshape(H, HD, S).contiguous()
    v = v.float().t().reshape(H, HD, S).contiguous()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for it in range(iters + 2):
            kc.klen = 0                    # (the pack writes columns 0 .. S-1 of an empty cache, ML:971-972)
            vc.vlen = 0
            kc.kcache.zero_()
            vc.vcache.zero_()
            if it < 2:                     # warm-up
                fn()
                continue
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / iters

    k_ms = timed(lambda: kc.parallel_pack(k))
    v_ms = timed(lambda: vc.parallel_pack(v))
    k_bytes = C * S * 4 + C * S * bits // 8 + S * 42 * 8 * 2            # prompt + codes + rows + mirror
    v_bytes = C * S * 4 + C * S * bits // 8 + S * 42 * 8 + S * (2 ** bits) * 4
    x = [torch.randn(S, H, HD, device=dev, dtype=torch.float16, generator=gen) for _ in range(3)]
    qa, ka, va = (t.transpose(0, 1) for t in x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.prefill_attention(qa, ka, va)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.prefill_attention(qa, ka, va)
    e1.record()
    torch.cuda.synchronize()
    a_ms = e0.elapsed_time(e1) / iters
    flops = 4.0 * H * S * S * HD / 2
    return {
        "metric": "prefill: NUQ pack of an %d-token prompt (one layer, K and V) + causal MFMA attention, LLaMA-2-7B head shape, nuq%d 1%%-sparse"
                  % (S, bits),
        "value": S / ((k_ms + v_ms) * 1e-3), "unit": "tokens/s (pack K + V, one layer)", "n_gpus": 1, "steps": iters, "warmup": 2,
        "ms_per_step": k_ms + v_ms, "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: prompt of %d tokens, H=32 hd=128, nuq%d + 1%% outliers (capped, 42/token)" % (S, bits),
                   "label": "prefill S=%d nuq%d (config 4)" % (S, bits), "bits": bits, "S": S},
        "roofline": {"bound": "hbm", "kernel": "pack_tiled_kernel (K)", "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": k_bytes / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": k_bytes, "avg_launch_us": k_ms * 1e3},
        "kernels": {"pack_k_us": k_ms * 1e3, "pack_v_us": v_ms * 1e3, "pack_k_GBps": k_bytes / (k_ms * 1e-3) / 1e9,
                    "pack_v_GBps": v_bytes / (v_ms * 1e-3) / 1e9, "prefill_attention_us": a_ms * 1e3,
                    "prefill_attention_TFLOPs": flops / (a_ms * 1e-3) / 1e12,
                    "prefill_attention_frac_of_2500_TF_dense_fp16": flops / (a_ms * 1e-3) / 1e12 / 2500.0},
    }


def cache_bytes_per_layer(bits, max_len, compact=False):
    """HBM bytes one layer's compressed K + V cache occupies for max_len token slots (kvquant_amd.cache.QuantK / QuantV):
    packed codes, outlier rows (+ the token-contiguous K mirror), the per-token V codebook rows"""
    dense = 2 * C * bits // 8                                     # K + V packed words
    k_out = (42 * 4) if compact else (42 * 8 * 2)                 # K rows + mirror (compact: the mirror only, 4-byte entries)
    v_out = (42 * 4) if compact else (42 * 8)
    v_rows = (2 ** bits) * 4
    return max_len * (dense + k_out + v_out + v_rows)


def check_memory(args, n_layers, streams, max_len, dev, what):
    """fail fast, with the arithmetic, instead of running into an allocator error minutes into the fill (VERDICT r3:
    `--gpus 8 --streams 1 --ctx 1048576` is config 5's shape and must say what it needs)"""
    need = n_layers * streams * cache_bytes_per_layer(args.bits, max_len, getattr(args, "compact", False))
    need += 2 * H * max_len * 4 + (64 << 20)                      # scores + probabilities scratch, slabs, tables
    need += 3 * 8192 * C * 4                                      # the fill's prompt chunk in flight (K, V fp32 views)
    total = torch.cuda.get_device_properties(dev).total_memory
    over = need > 0.94 * total
    # (every rank decides together: a rank that exits alone leaves the others in their first collective until the
    #  backend's timeout -- ADVICE r4)
    dist_mod = None
    try:
        import torch.distributed as dist_mod
        if not (dist_mod.is_available() and dist_mod.is_initialized() and dist_mod.get_world_size() > 1):
            dist_mod = None
    except Exception:
        dist_mod = None
    if dist_mod is not None:
        flag = torch.tensor([1 if over else 0], device=dev if dist_mod.get_backend() == "nccl" else "cpu", dtype=torch.int32)
        dist_mod.all_reduce(flag, op=dist_mod.ReduceOp.MAX)
        if int(flag.item()) and not over:
            raise SystemExit("bench.py: another rank does not have the HBM for its share (%s fits here): stopping together" % what)
    if over:
        raise SystemExit("bench.py: %s needs %.1f GB of HBM on this GPU (%d layers x %d stream(s) x %d token slots x %d B "
                         "per token and layer + scratch) but the device has %.1f GB: use more GPUs (--gpus), fewer streams "
                         "(--streams) or a shorter context (--ctx)"
                         % (what, need / 1e9, n_layers, streams, max_len,
                            cache_bytes_per_layer(args.bits, 1, getattr(args, "compact", False)), total / 1e9))
    return need


def run_config(args, rank, world, dev, dist, label=None, with_baselines=True):
    """build the caches of one configuration, time args.steps decode steps, return the result dict (rank 0) or None"""
    from kvquant_amd import sharding
    total = args.steps + args.warmup
    max_len = (args.ctx + total + 32 + 63) // 64 * 64             # (+ the retrieval check's and the full-model leg's tokens)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    sharded = world > 1 and not args.replicas
    streams = args.streams if args.streams > 0 else (world if sharded else 1)
    owned = sharding.layer_assignment(args.layers, world)[rank] if sharded else list(range(args.layers))
    check_memory(args, len(owned), streams, max_len, dev,
                 "rank %d of %d (%d of %d layers, %d stream(s), ctx %d)" % (rank, world, len(owned), args.layers, streams, args.ctx))
    t_setup = time.time()
    caches, qs, ks, vs = {}, {}, {}, {}
    plant_q = None
    for s in range(streams):
        for li in owned:
            lay = Layer(args.bits, max_len, gen, dev, args.sinks, getattr(args, "compact", False))
            lay.k.score_f16_pair = bool(getattr(args, "score_f16", False)) and args.bits == 3
            plant = None
            if args.retrieval and li == owned[0] and s == 0:
                # query 3x the usual norm at the decode position; the planted key is the query rotated back to its
                # own position (pre-RoPE).  Most of its channels saturate at the quantiser's end codes (the 21 + 21
                # largest keep their exact residuals), which still leaves a score of ~80 against <= ~50 for the
                # heaviest-tailed of the synthetic tokens: softmax mass ~1
                pos = args.ctx // 2
                plant_q = (torch.randn(H, HD, generator=gen, device=dev) * 3.0).half()
                kq = rope_rotate(rope_rotate(plant_q, args.ctx + args.sinks + total), pos + args.sinks, -1.0)
                plant = (pos, kq.reshape(-1), torch.randn(C, generator=gen, device=dev))
            lay.fill(args.ctx, gen, dev, plant=plant)
            caches[(s, li)] = lay
            k, v = synth_tokens(total, lay.scale, lay.shift, gen, dev)
            q = torch.randn(total, H, HD, generator=gen, device=dev).half()
            qs[(s, li)], ks[(s, li)], vs[(s, li)] = q, k, v
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def stage(s, step, x):
        for li in owned:
            q = qs[(s, li)][step]
            if sharded and li == owned[0]:
                q = q + x.view(H, HD) * 1e-3          # the hand-over is a true data dependency of this stage
            out = layer_step(caches[(s, li)], q, ks[(s, li)][step], vs[(s, li)][step])
        return out.half().view(1, 1, C)

    template = torch.zeros(1, 1, C, dtype=torch.float16, device=dev)
    x_in = torch.zeros(1, 1, C, dtype=torch.float16, device=dev)
    pipe = sharding.StreamPipeline(stage, streams, rank=rank, world=world if sharded else 1)
    timers = KernelTimers(every=max(1, args.time_every))
    timers.install()
    pipe.run(args.warmup, lambda s, st: x_in, template, step0=0)
    timers.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(args.steps, lambda s, st: x_in, template, step0=args.warmup)
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
    retrieval = None
    if args.retrieval and plant_q is not None:
        retrieval = check_retrieval(caches[(0, owned[0])], rope_rotate(plant_q, args.ctx + args.sinks + total).half(), dev)

    n_tokens = args.steps * streams * (world if (world > 1 and not sharded) else 1)
    ms_per_step = elapsed * 1000.0 / args.steps
    value = n_tokens / elapsed
    res = None
    if rank == 0:
        L_mid = args.ctx + args.warmup + args.steps // 2
        k_us, v_us = timers.mean_us("score_k"), timers.mean_us("mix_v")
        kb, _ = algorithmic_bytes(args.bits, L_mid, "score_k", getattr(args, "compact", False))
        vb, _ = algorithmic_bytes(args.bits, L_mid, "mix_v", getattr(args, "compact", False))
        # (the route the library actually took on the last step -- not a guess from the thresholds: a request for the fused
        #  kernel falls back to the kernel pair for compact caches, Q-Norm tables, unsupported shapes)
        from kvquant_amd import _lib as _klib
        fused_attend = _klib.lib().kvq_decode_step_route() == 3
        if fused_attend:
            # one kernel per layer (kvq_fused_attend: the library's event pairs are then the fused kernel and its merge);
            # its algorithmic bytes: both matvecs' minus the score write / probability read that no longer exist
            dom, dom_us = "fused_attend (q.K^T + softmax + p.V per tile)", k_us
            dom_bytes = kb + vb - L_mid * 8 * H
            per_tok = dom_bytes // L_mid
        else:
            dom = "score_k" if (k_us or 0) >= (v_us or 0) else "mix_v"
            dom_us = k_us if dom == "score_k" else v_us
            dom_bytes, per_tok = algorithmic_bytes(args.bits, L_mid, dom, getattr(args, "compact", False))
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, TRAFFIC_PROFILE)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom, {}).get("%d_%d" % (args.bits, args.ctx), tj.get(dom, {}).get(str(args.ctx)))      # (fused: none kept)
                if args.bits != 4 and ("%d_%d" % (args.bits, args.ctx)) not in tj.get(dom, {}):
                    traffic = None
                if getattr(args, "compact", False):
                    traffic = None          # (the kept PMC bytes are those of the reference format)
                traffic_src = tj.get("_source")
                from kvquant_amd import build as kbuild
                if traffic is not None and tj.get("_kernel_code_sha") != kbuild.kernel_source_hash():
                    traffic = None      # (measured on other kernel code than the one running: not this launch's traffic)
                    traffic_src = "stale: %s was measured on kernel code %s, this tree is %s (tools/pmc_run.sh + " \
                                  "tools/pmc_traffic.py re-measure)" % (tj.get("_source"), tj.get("_kernel_code_sha"),
                                                                        kbuild.kernel_source_hash())
            except Exception:
                traffic = None
        if sharded:
            par = "layer-sharded pipeline: %d ranks x %d layers, %d streams in flight, fp16 activation over RCCL p2p" \
                  % (world, len(owned), streams)
        elif world > 1:
            par = "independent decode streams x%d (replicas)" % world
        else:
            par = "1 GPU, %d stream%s" % (streams, "s" if streams > 1 else "")
        res = {
            "metric": "decode tokens/s, KV-cache hot path (%d layers: NUQ append + q.K^T(RoPE)+sparse + softmax + p.V+sparse), "
                      "LLaMA-2-7B head shape, nuq%d 1%%-sparse @%dK ctx" % (args.layers, args.bits, args.ctx // 1024),
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LLaMA-2-7B KV path: H=32 hd=128 layers=%d nuq%d + 1%% outliers (42/token), "
                                   "ctx=%d cached tokens%s, batch 1 per stream"
                                   % (args.layers, args.bits, args.ctx,
                                      " + %d fp16 attention-sink tokens" % args.sinks if args.sinks else ""),
                       "ctx": args.ctx, "bits": args.bits, "layers": args.layers, "sinks": args.sinks,
                       "streams": streams, "parallelism": par,
                       "outlier_format": "compact (fp16 residual + u16 channel, opt-in)" if getattr(args, "compact", False)
                       else "reference (f32 + i32)",
                       "score_tables": "fp16 pair-sum (opt-in)" if (getattr(args, "score_f16", False) and args.bits == 3)
                       else "fp32 (query-premultiplied codebooks)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_us": dom_us, "algorithmic_bytes_per_launch": dom_bytes,
                         "bytes_per_token": per_tok,
                         "timed_launches": len(timers.quads) or len(timers.pairs[dom]),
                         "timing": "HIP events on the launch stream around every %d-th layer's launch inside the timed region"
                                   % max(args.time_every, 1)},
            # the whole step against the same peak: every layer's algorithmic bytes (both matvecs) / the step's wall time
            "roofline_step": {"bound": "hbm", "achieved": len(owned) * streams * (kb + vb) / (elapsed / args.steps) / 1e9,
                              "peak": 8000.0, "unit": "GB/s",
                              "frac": len(owned) * streams * (kb + vb) / (elapsed / args.steps) / 1e9 / 8000.0,
                              "algorithmic_bytes_per_step": len(owned) * streams * (kb + vb),
                              "what": "q.K^T + p.V bytes of every layer / ms_per_step (appends, softmax merge, slab reduce "
                                      "and launch gaps are in the time, not in the bytes)"},
            "kernels": ({"fused_attend_us": k_us, "fused_merge_us": v_us,
                         "kv_matvec_GBps": dom_bytes / ((k_us + v_us) * 1e-6) / 1e9 if k_us and v_us else None,
                         "step_GBps": len(owned) * streams * dom_bytes / (elapsed / args.steps) / 1e9} if fused_attend else
                        {"score_k_us": k_us, "mix_v_us": v_us,
                         "score_k_GBps": kb / (k_us * 1e-6) / 1e9 if k_us else None,
                         "mix_v_GBps": vb / (v_us * 1e-6) / 1e9 if v_us else None,
                         "kv_matvec_GBps": (kb + vb) / ((k_us + v_us) * 1e-6) / 1e9 if k_us and v_us else None,
                         "step_GBps": len(owned) * streams * (kb + vb) / (elapsed / args.steps) / 1e9}),
            "setup_s": t_setup,
        }
        if label:
            res["config"]["label"] = label
        if retrieval is not None:
            res["retrieval"] = retrieval
    if rank == 0 and with_baselines and world == 1 and not getattr(args, "no_full_model", False) and res is not None:
        try:
            res["full_model"] = full_model_leg(args, caches, owned, max_len, dev)
        except Exception as e:      # (never take the headline down with it)
            res["full_model"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del caches, qs, ks, vs
    torch.cuda.empty_cache()
    if rank == 0 and with_baselines:
        if not args.no_fp16_baseline:
            fb = fp16_matvec_baseline(args.ctx, dev)
            if fb:
                res["fp16_baseline"] = fb
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.bits, args.cpu_sample_tokens, args.ctx, args.layers)
    return res


def config_summary(r):
    """one BASELINE configuration of the `configs` list of the default line: what the headline reports, in short"""
    k = r.get("kernels", {})
    out = {"label": r["config"].get("label"), "workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"],
           "ms_per_step": r["ms_per_step"], "steps": r["steps"],
           "frac": r["roofline"]["frac"], "roofline_kernel": r["roofline"].get("kernel"),
           "avg_launch_us": r["roofline"].get("avg_launch_us")}
    if "roofline_step" in r:
        out["step_frac"] = r["roofline_step"]["frac"]
    for key in ("score_k_us", "mix_v_us", "fused_attend_us", "fused_merge_us", "pack_k_us", "pack_v_us", "pack_k_GBps",
                "pack_v_GBps", "prefill_attention_us", "prefill_attention_TFLOPs",
                "prefill_attention_frac_of_2500_TF_dense_fp16"):
        if k.get(key) is not None:
            out[key] = k[key]
    if "retrieval" in r:
        out["retrieval"] = r["retrieval"]
    return out


def other_configs(args, rank, world, dev, dist):
    """The BASELINE configurations besides the headline, as short legs of the default run: config 2
    (4K / 32K decode), config 3 (nuq3 + 5 fp16 sink tokens at 128K), config 4 (8K-token prefill: pack + MFMA attention)
    and the 1M-token shape of config 5 on one GPU (8 of its 32 layers).  Same code path as the headline (run_config),
    fewer steps, no baselines; `frac` is the dominant kernel's fraction of 8 TB/s from its in-stream event timers."""
    base = dict(vars(args))
    legs = [("config 2: 4K decode (32 layers, rotated)", 4096, 4, 0, 32, 10),
            ("config 2: 32K decode", 32768, 4, 0, 32, 10),
            ("config 3: nuq3 + 5 fp16 sink tokens, 128K decode", 131072, 3, 5, 32, 8),
            ("config 5 shape on one GPU: 1M tokens, 8 of 32 layers", 1048576, 4, 0, 8, 4)]
    out = []
    for label, ctx, bits, sinks, layers, steps in legs:
        if (ctx, bits, sinks, layers) == (args.ctx, args.bits, args.sinks, args.layers):
            continue
        a = argparse.Namespace(**base)
        a.ctx, a.bits, a.sinks, a.layers, a.steps, a.warmup = ctx, bits, sinks, layers, steps, 2
        a.compact, a.score_f16, a.retrieval = False, False, False
        try:
            r = run_config(a, rank, world, dev, dist, label=label, with_baselines=False)
            out.append(config_summary(r))
        except Exception as e:          # (a leg never takes the headline down)
            out.append({"label": label, "error": "%s: %s" % (type(e).__name__, e)})
    try:
        out.append(config_summary(run_prefill_config(4, 8192, dev)))
    except Exception as e:
        out.append({"label": "prefill S=8192 nuq4 (config 4)", "error": "%s: %s" % (type(e).__name__, e)})
    return out


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU over RCCL, exactly the command the
        # driver would have used -- instead of silently measuring one GPU
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1:
            raise SystemExit("bench.py --gpus %d under a launcher with WORLD_SIZE=1" % args.gpus)
        args.gpus = world
    # development (a box with ONE GPU): KVQ_BENCH_ONE_GPU=1 puts every rank on cuda:0 and moves the hand-overs over gloo
    # (RCCL refuses two ranks on one device) -- the multi-rank code paths of this file, not a measurement
    one_gpu = os.environ.get("KVQ_BENCH_ONE_GPU") == "1"
    if os.environ.get("KVQ_BENCH_DUMP_AFTER"):      # development: where is every rank after N seconds (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["KVQ_BENCH_DUMP_AFTER"]), exit=True)
    if one_gpu:
        local = 0
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    from kvquant_amd import _lib
    if os.environ.get("KVQ_LIB"):      # development: time an alternative build of the library (tools/abl)
        _lib.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
    _lib.lib()  # fail loudly if the HIP library is missing

    if args.sweep:
        if world > 1:
            raise SystemExit("--sweep is a single-GPU run")
        base = dict(vars(args))
        # (label, ctx, bits, sinks, layers, steps): BASELINE configs 2, 3, 5 and the north-star sizes
        cfgs = [("4K hot (1 layer, Infinity-Cache resident)", 4096, 4, 0, 1, 50, False),
                ("4K rotated (32 layers)", 4096, 4, 0, 32, 20, False),
                ("32K", 32768, 4, 0, 32, 20, False),
                ("128K nuq3 + 5 sinks (config 3)", 131072, 3, 5, 32, 10, False),
                ("32K nuq3 + 5 sinks", 32768, 3, 5, 32, 20, False),
                ("128K nuq3 + 5 sinks, fp16 pair-sum score tables (opt-in)", 131072, 3, 5, 32, 10, "f16"),
                ("128K, compact outlier format (opt-in)", 131072, 4, 0, 32, 10, True),
                ("128K nuq3 + 5 sinks, compact outlier format (opt-in)", 131072, 3, 5, 32, 10, True),
                ("1M (config 5 shape on one GPU: 8 of 32 layers, retrieval proxy)", 1048576, 4, 0, 8, 5, False),
                ("1M nuq3 + 5 sinks (8 of 32 layers)", 1048576, 3, 5, 8, 5, False)]
        for label, ctx, bits, sinks, layers, steps, compact in cfgs:
            a = argparse.Namespace(**base)
            a.ctx, a.bits, a.sinks, a.layers, a.steps, a.compact = ctx, bits, sinks, layers, steps, compact is True
            a.score_f16 = compact == "f16"
            a.retrieval = ctx >= 1048576 and not compact
            r = run_config(a, rank, world, dev, dist, label=label, with_baselines=False)
            print(json.dumps(r), flush=True)
        for bits in (4, 3):
            print(json.dumps(run_prefill_config(bits, 8192, dev)), flush=True)
    if getattr(args, "prefill", False):
        if rank == 0:
            print(json.dumps(run_prefill_config(args.bits, 8192, dev)), flush=True)
        return
    if args.shard != "layers" and (getattr(args, "compact", False) or args.retrieval or (args.sinks and args.shard == "tokens")):
        raise SystemExit("bench.py --shard %s: the compact outlier format and the retrieval check belong to the layer placement; "
                         "fp16 sink tokens are carried by --shard heads (cache.shard_attention takes them per call, "
                         "run_token_sharded does not wire them)" % args.shard)
    if args.shard == "tokens":
        res = run_token_sharded(args, rank, world, dev, dist)
    elif args.shard == "heads":
        res = run_head_sharded(args, rank, world, dev, dist)
    else:
        res = run_config(args, rank, world, dev, dist)
        if world == 1 and res is not None and not args.no_configs and not args.retrieval:
            res["configs"] = other_configs(args, rank, world, dev, dist)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

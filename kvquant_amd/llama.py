"""The reference's patched Llama on a STOCK Hugging Face Llama (pip `transformers`): what
deployment/transformers (the vendored fork, ML = .../models/llama/modeling_llama.py) adds to the model, as a patch
applied at run time instead of a fork of the library.

    cfg = kvquant_config(LlamaConfig(...), abits=4, include_sparse=True, first_few_fp16=0, maxseqlen=4096)
    model = LlamaForCausalLM(cfg).half().cuda()
    patch_llama(model)                               # every self_attn now owns kcache / vcache (QuantK / QuantV)
    load_quantizers(model, "quantizers.pickle", include_sparse=True, sparsity_threshold=0.99)
    benchmark(model, input_ids, check=True)          # the token-by-token loop of deployment/llama.py:39-94

What carries over from the reference, by name:
  * the six LlamaConfig knobs (configuration_llama.py:139-176): dynamicrope, use_orig_sparse, first_few_fp16,
    maxseqlen, abits, include_sparse;
  * `layers[i].self_attn.kcache / .vcache` with reset() / load_lookup_table(quantizer, include_sparse,
    sparsity_threshold, norm) -- the calls deployment/llama.py:186-198 makes -- and the fp16 sink caches
    kcache_fp16 / vcache_fp16 (ML:1464-1466);
  * positions come from the caches' own length, not from HF's cache object: the reference threads
    `past_key_values_length_inp` through every forward (ML:1513, 2468-2508) only to recover that number; here
    `model(input_ids, use_cache=False)` is enough and the keyword is accepted and ignored;
  * `set_devices(model)`: contiguous chunks of decoder layers per visible GPU, `min(n-1, i // (L // n))`, embedding /
    norm / lm_head on the first (ML:2428-2453), activations moved at the split points (ML:2552-2556, 2583-2585);
  * the quantizer pickle: {"model.layers.N.self_attn.k_proj": (upper, lower, [centroids], [normscale, normoffset])}
    (simquant_module_quantizer.py:550-555), keys containing ".lut" skipped (deployment/llama.py:188-189).
Batch 1, MHA only, as the reference asserts (ML:1408, 1801).  Prefill attention is torch SDPA (the reference calls
flash-attn there); decode is the GPU-resident kernel path of kvquant_amd.cache.decode_kv.
"""
import pickle
import time
import types

import torch
import torch.nn as nn

from . import sharding
from .attention import KVQuantAttention

KNOBS = dict(dynamicrope=True, use_orig_sparse=False, first_few_fp16=0, maxseqlen=-1, abits=4, include_sparse=False)


def kvquant_config(config, **kw):
    """set the reference's six KVQuant fields on a LlamaConfig (defaults of configuration_llama.py:139-145)"""
    for k, v in KNOBS.items():
        setattr(config, k, kw.pop(k, getattr(config, k, v)))
    if kw:
        raise TypeError("unknown KVQuant config field(s): %s" % sorted(kw))
    return config


def _attn_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None,
                  past_key_values_length_inp=None, **kwargs):
    """replacement of LlamaAttention.forward (ML:1778-2011): projections of the stock module, KV path of kvquant_amd"""
    bsz, q_len, _ = hidden_states.shape
    core = self.kvq
    shape = (bsz, q_len, core.num_heads, core.head_dim)
    q = self.q_proj(hidden_states).view(shape).transpose(1, 2)
    k = self.k_proj(hidden_states).view(shape).transpose(1, 2)
    v = self.v_proj(hidden_states).view(shape).transpose(1, 2)
    out = core.attend(q, k, v)
    return self.o_proj(out.to(hidden_states.dtype)), None


def patch_llama(model, sparsity_threshold=0.99):
    """give every decoder layer's self_attn the compressed KV path.  `model`: LlamaForCausalLM / LlamaModel whose
    config carries the KVQuant fields (kvquant_config).  Caches are created on each layer's current device."""
    cfg = model.config
    kvquant_config(cfg)
    if getattr(cfg, "num_key_value_heads", cfg.num_attention_heads) != cfg.num_attention_heads:
        raise ValueError("KVQuant's deployment path is MHA only (ML:1408)")
    if cfg.maxseqlen is None or cfg.maxseqlen <= 0:
        raise ValueError("config.maxseqlen must be set (size of the preallocated compressed cache)")
    theta = getattr(cfg, "rope_theta", None)
    if theta is None:      # transformers >= 5 keeps it in rope_parameters
        theta = (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
    base = model.model if hasattr(model, "model") else model
    for layer in base.layers:
        attn = layer.self_attn
        dev = next(attn.parameters()).device
        core = KVQuantAttention(hidden_size=cfg.hidden_size, num_heads=cfg.num_attention_heads, abits=cfg.abits,
                                include_sparse=cfg.include_sparse, first_few_fp16=cfg.first_few_fp16,
                                maxseqlen=cfg.maxseqlen, rope_theta=float(theta), sparsity_threshold=sparsity_threshold,
                                device=dev, dtype=next(attn.parameters()).dtype, make_proj=False)
        # plain attributes, like the reference's (ML:1440-1466): they do not follow module.to(); use set_devices
        object.__setattr__(attn, "kvq", core)
        object.__setattr__(attn, "kcache", core.kcache)
        object.__setattr__(attn, "vcache", core.vcache)
        if cfg.first_few_fp16 > 0:
            object.__setattr__(attn, "kcache_fp16", core.kcache_fp16)
            object.__setattr__(attn, "vcache_fp16", core.vcache_fp16)
        attn.forward = types.MethodType(_attn_forward, attn)
    model.kvquant_patched = True
    return model


def reset_caches(model):
    base = model.model if hasattr(model, "model") else model
    for layer in base.layers:
        layer.self_attn.kvq.reset()


def load_quantizers(model, quantizers, include_sparse=True, sparsity_threshold=0.99, norm=False):
    """deployment/llama.py:178-198: `quantizers` is the pickle path or the dict it holds."""
    if isinstance(quantizers, (str, bytes)):
        with open(quantizers, "rb") as f:
            quantizers = pickle.load(f)
    base = model.model if hasattr(model, "model") else model
    layers = base.layers
    for k, q in quantizers.items():
        if ".lut" in k:
            continue
        ln = int(k.split(".")[-3])          # "model.layers.<N>.self_attn.k_proj"
        if "k_proj" in k:
            layers[ln].self_attn.kcache.reset()
            layers[ln].self_attn.kcache.load_lookup_table(q, include_sparse, sparsity_threshold, norm)
        elif "v_proj" in k:
            layers[ln].self_attn.vcache.reset()
            layers[ln].self_attn.vcache.load_lookup_table(q, include_sparse, sparsity_threshold, norm)
    return model


def set_devices(model, devices=None):
    """LlamaModel.set_devices (ML:2428-2453) for one process driving several GPUs: layer i lives on device
    min(n-1, i // (L // n)); embedding, final norm and lm_head on the first; hooks move the activation at the split
    points and back (ML:2552-2556, 2583-2585).  Call BEFORE patch_llama so that the caches are created where their
    layer lives (the reference's .cuda()-pinned caches stay on device 0).  The one-process-per-GPU
    form used for the scaling benchmark is kvquant_amd.sharding.StreamPipeline."""
    if devices is None:
        devices = ["cuda:%d" % i for i in range(torch.cuda.device_count())]
    devices = [torch.device(d) for d in devices]
    base = model.model if hasattr(model, "model") else model
    n = len(devices)
    base.embed_tokens.to(devices[0])
    base.norm.to(devices[0])
    if hasattr(base, "rotary_emb"):
        base.rotary_emb.to(devices[0])
    if hasattr(model, "lm_head"):
        model.lm_head.to(devices[0])
    placement = []
    for i, layer in enumerate(base.layers):
        d = devices[sharding.layer_device(i, len(base.layers), n)]
        layer.to(d)
        placement.append(d)

        def pre(mod, args, kwargs, d=d):
            def mv(t):
                if torch.is_tensor(t):
                    return t.to(d)
                if isinstance(t, tuple):
                    return tuple(mv(x) for x in t)
                return t
            return tuple(mv(a) for a in args), {k: mv(v) for k, v in kwargs.items()}
        if n > 1:
            layer.register_forward_pre_hook(pre, with_kwargs=True)
    if n > 1:
        base.norm.register_forward_pre_hook(lambda mod, args: tuple(a.to(devices[0]) if torch.is_tensor(a) else a for a in args))
    model.gpus = devices
    model.split_indices = sharding.split_indices(len(base.layers), n)
    return placement


@torch.no_grad()
def benchmark(model, input_ids, check=False, verbose=False):
    """deployment/llama.py:39-94: feed the tokens one at a time through the compressed-cache decode path, time every
    step, optionally accumulate the next-token loss.  Returns {"median_s", "times", "ppl" (check), "max_memory_mib"}."""
    dev = model.gpus[0] if hasattr(model, "gpus") else next(model.parameters()).device
    input_ids = input_ids.to(dev)
    loss_fn = nn.CrossEntropyLoss()
    tot = 0.0
    times = []
    max_mem = 0.0
    n = input_ids.numel()

    def sync():
        for g in (model.gpus if hasattr(model, "gpus") else [dev]):
            torch.cuda.synchronize(g)
    sync()
    for i in range(n):
        tick = time.time()
        out = model(input_ids[:, i:i + 1], use_cache=False)
        sync()
        times.append(time.time() - tick)
        if verbose:
            print(i, times[-1])
        max_mem = max(max_mem, torch.cuda.memory_allocated() / 1024 / 1024)
        if check and i != n - 1:
            tot += loss_fn(out.logits[0].float(), input_ids[:, i + 1]).float()
        del out
    res = {"median_s": float(torch.tensor(times).median()), "times": times, "max_memory_mib": max_mem}
    if check:
        res["ppl"] = float(torch.exp(tot / (n - 1)))
    return res


@torch.no_grad()
def prefill_then_decode(model, input_ids, n_prompt, check=True):
    """the generation protocol of the reference (generation/utils.py:2325-2416): one parallel prefill of the first
    n_prompt tokens (parallel pack of the prompt's K / V), then one token at a time; PPL over all next-token
    predictions as in benchmark()."""
    dev = next(model.parameters()).device
    input_ids = input_ids.to(dev)
    loss_fn = nn.CrossEntropyLoss(reduction="sum")
    n = input_ids.numel()
    out = model(input_ids[:, :n_prompt], use_cache=False)
    tot = loss_fn(out.logits[0, :-1].float(), input_ids[0, 1:n_prompt]).double()
    last = out.logits[0, -1:]
    for i in range(n_prompt, n):
        tot += loss_fn(last.float(), input_ids[:, i]).double()
        last = model(input_ids[:, i:i + 1], use_cache=False).logits[0]
    return {"ppl": float(torch.exp(tot / (n - 1)))} if check else {}

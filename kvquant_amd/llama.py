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
Batch 1, MHA only, as the reference asserts (ML:1408, 1801).  Prefill attention is the library's own MFMA flash
kernel (kvq_prefill_attention; the reference calls flash-attn there, ML:2013-2070); decode is the GPU-resident kernel
path of kvquant_amd.cache.decode_kv.

Command line = deployment/llama.py:100-216, same positional arguments and flags:

    python -m kvquant_amd.llama <model> <dataset> --abits 4 --include_sparse --sparsity-threshold 0.99 \
        --first_few_fp16 1 --maxseqlen 4096 --quantizer-path quantizers.pickle --benchmark 128 --check

<dataset> is wikitext2 | ptb | c4 (through `datasets`, as kvquant/datautils.py; needs the hub or a warm cache) or,
for machines without network, `synthetic` (seeded random token ids) or a path to a saved LongTensor / .npy of token ids.
`generate()` is the greedy loop of the reference's `generate(..., kvquant=True)` (generation/utils.py:2325-2416).
"""
import os
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


def patch_llama(model, sparsity_threshold=0.99, compact=False):
    """give every decoder layer's self_attn the compressed KV path.  `model`: LlamaForCausalLM / LlamaModel whose
    config carries the KVQuant fields (kvquant_config).  Caches are created on each layer's current device.
    compact=True: the opt-in 4-byte outlier entries (fp16 residual + channel; kvquant_amd.cache) -- the
    GPU-resident paths only, lossy against the reference format by the fp16 rounding of the residuals."""
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
                                device=dev, dtype=next(attn.parameters()).dtype, make_proj=False, compact=compact)
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


# ---- the reference's driver (deployment/llama.py) ------------------------------------------------------------------
def get_model(model, seqlen, maxseqlen, bits, include_sparse, first_few_fp16):
    """deployment/llama.py:20-36: config knobs set before the weights are loaded, fp16 weights on the CPU, no
    initialisation work; `model` is a hub name or a local directory.  Stock transformers (no vendored fork): the
    compressed KV path is attached afterwards by patch_llama."""
    import transformers

    def skip(*args, **kwargs):
        pass
    torch.nn.init.kaiming_uniform_ = skip
    torch.nn.init.uniform_ = skip
    torch.nn.init.normal_ = skip
    config = transformers.AutoConfig.from_pretrained(model)
    kvquant_config(config, first_few_fp16=first_few_fp16, maxseqlen=maxseqlen, abits=bits, include_sparse=include_sparse)
    m = transformers.AutoModelForCausalLM.from_pretrained(model, config=config, torch_dtype=torch.half)
    m.seqlen = seqlen
    return m


def get_loaders(name, nsamples=128, seed=0, seqlen=2048, model="", vocab_size=32000):
    """kvquant/datautils.py:get_loaders -> (trainloader [(inp [1, seqlen], tar)], test ids [1, N]).  wikitext2 / ptb / c4
    as the reference reads them; `synthetic`: seeded uniform token ids (offline machines); a file path: a saved
    LongTensor (torch.save) or .npy of token ids, cut into nsamples windows the same way (random.seed(seed))."""
    import random
    if name in ("wikitext2", "ptb", "c4"):
        from datasets import load_dataset
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(model, use_fast=False)
        if name == "wikitext2":
            tr = load_dataset("wikitext", "wikitext-2-raw-v1", split="train")
            te = load_dataset("wikitext", "wikitext-2-raw-v1", split="test")
            train_ids = tok("\n\n".join(tr["text"]), return_tensors="pt").input_ids
            test_ids = tok("\n\n".join(te["text"]), return_tensors="pt").input_ids
        elif name == "ptb":
            tr = load_dataset("ptb_text_only", "penn_treebank", split="train")
            te = load_dataset("ptb_text_only", "penn_treebank", split="validation")
            train_ids = tok("\n\n".join(tr["sentence"]), return_tensors="pt").input_ids
            test_ids = tok("\n\n".join(te["sentence"]), return_tensors="pt").input_ids
        else:
            tr = load_dataset("allenai/c4", data_files={"train": "en/c4-train.00000-of-01024.json.gz"}, split="train")
            te = load_dataset("allenai/c4", data_files={"validation": "en/c4-validation.00000-of-00008.json.gz"},
                              split="validation")
            train_ids = tok(" ".join(tr[:1100]["text"]), return_tensors="pt").input_ids
            test_ids = tok(" ".join(te[:1100]["text"]), return_tensors="pt").input_ids[:, :256 * seqlen]
    elif name == "synthetic":
        g = torch.Generator().manual_seed(seed)
        train_ids = torch.randint(0, vocab_size, (1, max(4 * seqlen, nsamples * 64 + seqlen + 1)), generator=g)
        test_ids = torch.randint(0, vocab_size, (1, 8 * seqlen), generator=g)
    elif os.path.exists(name):
        if name.endswith(".npy"):
            import numpy as np
            ids = torch.from_numpy(np.load(name)).long().reshape(1, -1)
        else:
            ids = torch.load(name).long().reshape(1, -1)
        train_ids = test_ids = ids
    else:
        raise ValueError("dataset must be wikitext2, ptb, c4, synthetic or a file of token ids: %r" % name)
    random.seed(seed)
    loader = []
    for _ in range(nsamples):
        i = random.randint(0, train_ids.shape[1] - seqlen - 1)
        inp = train_ids[:, i:i + seqlen]
        tar = inp.clone()
        tar[:, :-1] = -100
        loader.append((inp, tar))
    return loader, test_ids


@torch.no_grad()
def _warp_logits(logits, temperature, top_k, top_p):
    """HF's sampling warpers in their order (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)"""
    if temperature is not None and temperature != 1.0:
        logits = logits / temperature
    if top_k is not None and 0 < top_k < logits.shape[-1]:
        kth = torch.topk(logits, top_k, dim=-1).values[..., -1, None]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(logits, descending=False, dim=-1)
        cum = srt.softmax(dim=-1).cumsum(dim=-1)
        drop = cum <= (1 - top_p)
        drop[..., -1:] = False                                  # (min_tokens_to_keep = 1)
        logits = logits.masked_fill(drop.scatter(-1, idx, drop), float("-inf"))
    return logits


@torch.no_grad()
def generate(model, input_ids, max_length=None, max_new_tokens=None, eos_token_id=None, pad_token_id=None,
             do_sample=False, temperature=1.0, top_k=50, top_p=1.0, min_length=0, generator=None):
    """Decoding with the compressed cache: the reference's `model.generate(..., kvquant=True)`
    (generation/utils.py:2325-2416) -- the prompt goes through the model once (parallel pack of its K / V), then every
    step feeds ONLY the newest token (`input_ids = next_tokens[:, None]`, :2378-2380), the returned sequence is the
    prompt with the generated tokens appended, and the loop stops at `max_length` positions (:2401-2406) or once EOS
    has been produced.  Greedy by default; `do_sample=True` is what lwm/llama_inference.py:124-131 asks for
    (multinomial sampling after HF's default warpers: temperature 1, top-k 50; `min_length` suppresses EOS until the
    sequence is that long).  Batch 1."""
    if input_ids.shape[0] != 1:
        raise ValueError("the compressed-cache path is batch 1 (ML:1408)")
    dev = model.gpus[0] if hasattr(model, "gpus") else next(model.parameters()).device
    input_ids = input_ids.to(dev)
    n_prompt = input_ids.shape[1]
    if max_length is None:
        max_length = n_prompt + (max_new_tokens if max_new_tokens is not None else 20)
    if eos_token_id is not None and not isinstance(eos_token_id, (list, tuple)):
        eos_token_id = [eos_token_id]
    if eos_token_id is not None and pad_token_id is None:
        raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")
    seq = input_ids
    cur = input_ids
    unfinished = True
    pos = n_prompt - 1                   # position of the last token fed so far
    while True:
        out = model(cur, use_cache=False)
        logits = out.logits[:, -1, :].float()
        if eos_token_id is not None and seq.shape[1] < min_length:
            logits[:, eos_token_id] = float("-inf")             # MinLengthLogitsProcessor
        if do_sample:
            probs = _warp_logits(logits, temperature, top_k, top_p).softmax(dim=-1)
            nxt = torch.multinomial(probs, num_samples=1, generator=generator).squeeze(1)
        else:
            nxt = torch.argmax(logits, dim=-1)
        if eos_token_id is not None and not unfinished:
            nxt = torch.full_like(nxt, pad_token_id)
        seq = torch.cat((seq, nxt[:, None]), dim=-1)
        cur = nxt[:, None]
        pos += 1
        if eos_token_id is not None and int(nxt) in eos_token_id:
            unfinished = False
        if not unfinished or seq.shape[1] >= max_length or max_length <= pos + 1:
            break
    return seq


def inference_main(argv=None):
    """lwm/llama_inference.py:39-138 (the same arguments): load a long-context Llama with the compressed cache sized by
    --maxseqlen, read the quantizer pickle, sample a continuation of --text and print it with the wall time.
    Extras for boxes without a tokenizer: --token-ids "1,2,3" instead of --text (the ids are printed back), --seed."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m kvquant_amd.llama inference")
    ap.add_argument("model", type=str, help="llama model to load")
    ap.add_argument("--abits", type=int, default=16, choices=[2, 3, 4, 16],
                    help="#bits to use for quantization; use 16 for evaluating base model.")
    ap.add_argument("--text", type=str, help="input text")
    ap.add_argument("--token-ids", type=str, default=None, help="(extra) comma-separated prompt token ids instead of --text")
    ap.add_argument("--min_length", type=int, default=10, help="The minimum length of the sequence to be generated.")
    ap.add_argument("--max_length", type=int, default=256, help="The maximum length of the sequence to be generated.")
    ap.add_argument("--maxseqlen", type=int, default=-1, help="Used to set KV cache size")
    ap.add_argument("--quantizer-path", type=str, help="Path to quantizers.")
    ap.add_argument("--include_sparse", action="store_true", help="Whether to use dense-and-sparse quantization.")
    ap.add_argument("--sparsity-threshold", type=float, default=1, help="Outlier percentile.")
    ap.add_argument("--first_few_fp16", type=int, default=0, help="Store first few tokens separately in fp16")
    ap.add_argument("--norm", action="store_true", help="Whether to use q-norm.")
    ap.add_argument("--seed", type=int, default=None, help="(extra) seed of the sampler")
    ap.add_argument("--greedy", action="store_true", help="(extra) arg-max decoding instead of sampling")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("kvquant_amd.llama needs a GPU (no CPU fallback)")
    if args.abits != 16 and args.maxseqlen <= 0:
        raise SystemExit("--maxseqlen must be set: it sizes the preallocated compressed cache")
    if (args.text is None) == (args.token_ids is None):
        raise SystemExit("give the prompt as --text or as --token-ids")
    dev = torch.device("cuda:0")
    model = get_model(args.model, 2048, args.maxseqlen, args.abits if args.abits != 16 else 4, args.include_sparse,
                      args.first_few_fp16)
    model.eval()
    model = model.half()
    tokenizer = None
    if args.text is not None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model, use_fast=False)
        input_ids = tokenizer.encode(args.text, return_tensors="pt")
    else:
        input_ids = torch.tensor([[int(t) for t in args.token_ids.split(",")]], dtype=torch.long)
    eos = getattr(model.config, "eos_token_id", None)
    pad = getattr(model.config, "pad_token_id", None)
    if eos is not None and pad is None:
        pad = eos if not isinstance(eos, (list, tuple)) else eos[0]
    gen = None
    if args.seed is not None:
        gen = torch.Generator(device=dev).manual_seed(args.seed)
    if args.abits != 16:
        set_devices(model)
        patch_llama(model, sparsity_threshold=args.sparsity_threshold)
        print("Load quantizers.")
        load_quantizers(model, args.quantizer_path, args.include_sparse, args.sparsity_threshold, args.norm)
        t1 = time.time()
        with torch.no_grad():
            ids = generate(model, input_ids, max_length=args.max_length, min_length=args.min_length, eos_token_id=eos,
                           pad_token_id=pad, do_sample=not args.greedy, generator=gen)
        torch.cuda.synchronize()
        t2 = time.time()
    else:
        model.to(dev)
        if args.seed is not None:
            torch.manual_seed(args.seed)
        t1 = time.time()
        with torch.no_grad():
            ids = model.generate(input_ids.to(dev), do_sample=not args.greedy, min_length=args.min_length,
                                 max_length=args.max_length, use_cache=True, pad_token_id=pad)
        torch.cuda.synchronize()
        t2 = time.time()
    if tokenizer is not None:
        print(tokenizer.decode([el.item() for el in ids[0]]))
    else:
        print("token ids:", ids[0].tolist())
    print("Time: ", t2 - t1)
    return 0


def main(argv=None):
    """deployment/llama.py:100-216 (the same arguments); prints what the reference prints."""
    import argparse
    import numpy as np
    ap = argparse.ArgumentParser(prog="python -m kvquant_amd.llama")
    ap.add_argument("model", type=str, help="llama model to load")
    ap.add_argument("dataset", type=str, help="wikitext2 | ptb | c4 | synthetic | file of token ids")
    ap.add_argument("--nsamples", type=int, default=128, help="Number of calibration data samples.")
    ap.add_argument("--seed", type=int, default=0, help="Seed for sampling the calibration data.")
    ap.add_argument("--abits", type=int, default=16, choices=[2, 3, 4, 16],
                    help="#bits to use for quantization; use 16 for evaluating base model.")
    ap.add_argument("--benchmark", type=int, default=0, help="Number of tokens to use for benchmarking.")
    ap.add_argument("--check", action="store_true", help="Whether to compute perplexity during benchmarking for verification.")
    ap.add_argument("--torch_profile", action="store_true", help="Use the torch profiler for timing runs.")
    ap.add_argument("--seqlen", type=int, default=2048, help="Used by dataloader")
    ap.add_argument("--maxseqlen", type=int, default=-1, help="Used to set KV cache size")
    ap.add_argument("--quantizer-path", type=str, help="Path to quantizers.")
    ap.add_argument("--include_sparse", action="store_true", help="Whether to use dense-and-sparse quantization.")
    ap.add_argument("--sparsity-threshold", type=float, default=1, help="Outlier percentile.")
    ap.add_argument("--first_few_fp16", type=int, default=0, help="Store first few tokens separately in fp16")
    ap.add_argument("--norm", action="store_true", help="Whether to use q-norm.")
    ap.add_argument("--generate", type=int, default=0, help="(extra) greedy-generate this many tokens after the benchmark prompt")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("kvquant_amd.llama needs a GPU (no CPU fallback)")
    if args.abits != 16 and args.maxseqlen <= 0:
        raise SystemExit("--maxseqlen must be set: it sizes the preallocated compressed cache")
    dev = torch.device("cuda:0")
    model = get_model(args.model, args.seqlen, args.maxseqlen, args.abits if args.abits != 16 else 4,
                      args.include_sparse, args.first_few_fp16)
    model.eval()
    model = model.half()
    dataloader, _ = get_loaders(args.dataset, nsamples=args.nsamples, seed=args.seed, model=args.model,
                                seqlen=model.seqlen, vocab_size=model.config.vocab_size)
    if args.abits != 16:
        set_devices(model)                       # (ML:2428-2453) before the caches exist: they are created where the layer lives
        patch_llama(model, sparsity_threshold=args.sparsity_threshold)
        print("Load quantizers.")
        load_quantizers(model, args.quantizer_path, args.include_sparse, args.sparsity_threshold, args.norm)
    else:
        model.to(dev)
    if args.benchmark:
        input_ids = next(iter(dataloader))[0][:, :args.benchmark]
        print("Benchmarking ...")

        def run():
            if args.abits == 16:
                return benchmark_fp16(model, input_ids, check=args.check, verbose=True)
            return benchmark(model, input_ids, check=args.check, verbose=True)
        if args.torch_profile:
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU,
                                                    torch.profiler.ProfilerActivity.CUDA]) as p:
                res = run()
            print(p.key_averages().table(sort_by="self_cuda_time_total", row_limit=-1))
        else:
            res = run()
        print("Median:", float(np.median(res["times"])))
        if args.check:
            print("PPL:", res["ppl"])
            print("max memory(MiB):", res["max_memory_mib"])
        if args.generate and args.abits != 16:
            seq = generate(model, input_ids[:, -1:], max_new_tokens=args.generate)
            print("generated:", seq[0, 1:].tolist())
    return 0


@torch.no_grad()
def benchmark_fp16(model, input_ids, check=False, verbose=False):
    """--abits 16: the un-patched model, token by token with HF's own fp16 KV cache (the reference's baseline run)"""
    dev = next(model.parameters()).device
    input_ids = input_ids.to(dev)
    loss_fn = nn.CrossEntropyLoss()
    tot, times, past = 0.0, [], None
    n = input_ids.numel()
    for i in range(n):
        torch.cuda.synchronize()
        tick = time.time()
        out = model(input_ids[:, i:i + 1], past_key_values=past, use_cache=True)
        torch.cuda.synchronize()
        times.append(time.time() - tick)
        if verbose:
            print(i, times[-1])
        past = out.past_key_values
        if check and i != n - 1:
            tot += loss_fn(out.logits[0].float(), input_ids[:, i + 1]).float()
    res = {"median_s": float(torch.tensor(times).median()), "times": times,
           "max_memory_mib": torch.cuda.max_memory_allocated() / 1024 / 1024}
    if check:
        res["ppl"] = float(torch.exp(tot / (n - 1)))
    return res


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "inference":        # lwm/llama_inference.py
        raise SystemExit(inference_main(sys.argv[2:]))
    raise SystemExit(main())

#!/usr/bin/env python
"""Generate tests/golden/ref_attention_*.npz by EXECUTING the reference's own patched ``LlamaAttention.forward``
(deployment/transformers/src/transformers/models/llama/modeling_llama.py:1388-1760: eager prefill branch, fp16
attention-sink caches ML:1464-1466 / 1932-1995, decode over QuantK / QuantV) on CPU.

Runs only in the build container (needs /root/reference); the fixtures are committed and travel to the GPU box.

How the reference is run here: its vendored ``transformers`` fork is imported from where it lies (the fork's
``tokenizers`` / ``huggingface-hub`` version gate is answered with the versions it pins), nothing of it is copied into
this repo.  ``quant_cuda`` -- the CUDA extension the module imports -- is oracle.quant_cuda_ref (the C restatement of
the kernels, itself pinned to the reference's kernels on the GPU by tests/test_ref_gpu.py); ``Tensor.cuda()`` and the
side stream of ML:1804-1820 are patched to CPU no-ops.  So these fixtures pin the ATTENTION MODULE of the reference --
RoPE on the query only, the order of half() / division / softmax dtypes, the sink-token bookkeeping, which tokens are
packed and which stay fp16, the prefill branch -- not a restatement of it.

Stored per scenario: the hidden states, what q/k/v_proj returned (hooks), the input of o_proj and the module's output for
the prefill call and every decode call, and the final state of the compressed cache + fp16 sink caches.

usage: python tests/golden/gen_attention.py
"""
import contextlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/deployment/transformers/src"
OUT = os.path.dirname(os.path.abspath(__file__))
H, HD, C = 32, 128, 4096


def import_reference():
    sys.path.insert(0, REF_SRC)
    from oracle import quant_cuda_ref
    sys.modules["quant_cuda"] = quant_cuda_ref.as_module()
    torch.Tensor.cuda = lambda self, *a, **k: self

    class _Stream:
        def __init__(self, *a, **k):
            pass

        def wait_stream(self, *a, **k):
            pass

    torch.cuda.Stream = _Stream
    torch.cuda.default_stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    import importlib.metadata as im
    real = im.version
    pins = {"tokenizers": "0.15.2", "huggingface-hub": "0.20.3", "huggingface_hub": "0.20.3"}
    im.version = lambda name: pins.get(name) or real(name)
    import warnings
    warnings.filterwarnings("ignore")
    import transformers
    assert transformers.__file__.startswith(REF_SRC), transformers.__file__
    from transformers.models.llama import modeling_llama as ML
    from transformers.models.llama.configuration_llama import LlamaConfig
    return ML, LlamaConfig


def nf_centroids(bits):
    from tests.golden.gen_golden import nf_centroids as f
    return f(bits)


def tie_free(k16, v16, upper16, lower16):
    """tokens whose selections are unambiguous: no equal values across the cut of the V top-22 / top-21 and of the K
    rescaled top-21 (torch.topk leaves the choice among equal values unspecified)"""
    v = v16.float()
    hi = torch.topk(v, 23, dim=-1).values
    lo = torch.topk(v, 23, dim=-1, largest=False).values
    ok = (hi[:, 20] != hi[:, 21]) & (hi[:, 21] != hi[:, 22]) & (lo[:, 20] != lo[:, 21]) & (lo[:, 21] != lo[:, 22])
    up, dn = upper16.float(), lower16.float()
    r = (k16.float() - (up + dn) / 2) / ((up - dn) / 2)          # KCU:1759-1764
    rh = torch.topk(r, 22, dim=-1).values
    rl = torch.topk(r, 22, dim=-1, largest=False).values
    return ok & (rh[:, 20] != rh[:, 21]) & (rl[:, 20] != rl[:, 21])


def run(ML, LlamaConfig, name, bits, sinks, S, steps, seed):
    from tests import attn_fixture
    max_len = 64
    cfg = LlamaConfig(hidden_size=C, num_attention_heads=H, num_key_value_heads=H, num_hidden_layers=1,
                      intermediate_size=64, max_position_embeddings=max_len, rope_theta=10000.0,
                      first_few_fp16=sinks, maxseqlen=max_len, abits=bits, include_sparse=True, dynamicrope=True)
    # built as deployment/llama.py:34 builds it -- from_pretrained(torch_dtype=torch.half) constructs the modules under
    # a float16 default dtype: fp16 projections, while the buffers with explicit dtypes (inv_freq of the dynamic RoPE, the
    # caches) keep theirs.  (module.half() afterwards would round inv_freq to fp16, which the deployment never does.)
    torch.set_default_dtype(torch.float16)
    try:
        att = ML.LlamaAttention(cfg, layer_idx=0)
    finally:
        torch.set_default_dtype(torch.float32)
    assert att.rotary_emb.inv_freq.dtype == torch.float32 and att.q_proj.weight.dtype == torch.float16
    w = attn_fixture.make_weights(seed)
    with torch.no_grad():
        att.q_proj.weight.copy_(w["q"])
        att.k_proj.weight.copy_(w["k"])
        att.v_proj.weight.copy_(w["v"])
        att.o_proj.weight.copy_(w["o"])
    att.eval()
    g = torch.Generator().manual_seed(seed + 1)
    # quantizer: calibrated on the module's own K activations (percentile thresholds as SQ:465-474)
    calib = (torch.randn(256, C, generator=g) * 1.0).half()
    with torch.no_grad():
        kcal = att.k_proj(calib).float().numpy()
    upper = np.percentile(kcal, 99.5, axis=0)[None, :]
    lower = np.percentile(kcal, 0.5, axis=0)[None, :]
    quant = (upper, lower, [nf_centroids(bits)])
    # candidate tokens, keep the first S + steps whose selections are free of ties
    T = S + steps
    cand = (torch.randn(6 * T, C, generator=g) * 1.0).half()
    with torch.no_grad():
        kc, vc = att.k_proj(cand), att.v_proj(cand)
    ok = tie_free(kc, vc, torch.tensor(upper[0]).half(), torch.tensor(lower[0]).half())
    idx = torch.nonzero(ok).flatten()[:T]
    assert len(idx) == T, "not enough tie-free tokens"
    hidden = cand[idx].unsqueeze(0).contiguous()             # [1, T, C] fp16
    # deployment/llama.py:186-198
    att.kcache.reset()
    att.vcache.reset()
    att.kcache.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=False)
    att.vcache.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=False)

    cap = {}
    hooks = [att.q_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("q", o.detach().clone())),
             att.k_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("k", o.detach().clone())),
             att.v_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("v", o.detach().clone())),
             att.o_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("ctx", i[0].detach().clone()))]
    out = {"bits": bits, "sinks": sinks, "S": S, "steps": steps, "max_len": max_len, "theta": 10000.0, "seed": seed,
           "weights_checksum": attn_fixture.checksum(w),
           "q_upper": upper.astype(np.float32), "q_lower": lower.astype(np.float32), "q_centroids": quant[2][0],
           "hidden": hidden.numpy()}
    qs, ks, vs, ctxs, outs = [], [], [], [], []
    with torch.no_grad():
        # prefill: the causal mask the model builds (additive, dtype minimum above the diagonal)
        mask = torch.full((S, S), torch.finfo(torch.float16).min, dtype=torch.float16).triu(1)[None, None]
        y, _, _ = att(hidden[:, :S], attention_mask=mask, position_ids=torch.arange(S)[None])
        for lst, key in ((qs, "q"), (ks, "k"), (vs, "v"), (ctxs, "ctx")):
            lst.append(cap[key][0])
        outs.append(y[0])
        for i in range(steps):
            t = S + i
            y, _, _ = att(hidden[:, t:t + 1], attention_mask=None, position_ids=torch.tensor([[t]]))
            for lst, key in ((qs, "q"), (ks, "k"), (vs, "v"), (ctxs, "ctx")):
                lst.append(cap[key][0])
            outs.append(y[0])
    for h in hooks:
        h.remove()
    out["q_states"] = torch.cat(qs).numpy()                  # [T, C] fp16, what q_proj returned (pre-RoPE)
    out["k_states"] = torch.cat(ks).numpy()
    out["v_states"] = torch.cat(vs).numpy()
    out["ctx"] = torch.cat(ctxs).numpy()                     # [T, C] fp16: the input of o_proj
    out["out"] = torch.cat(outs).numpy()                     # [T, C] fp16: the module's output
    kq, vq = att.kcache, att.vcache
    L = kq.klen - sinks
    assert kq.klen == T and vq.vlen == T, (kq.klen, vq.vlen)
    out["L"] = L
    out["kcache"] = kq.kcache[:, :, :L].numpy().copy()
    out["vcache"] = vq.vcache[:, :, :L].numpy().copy()
    out["k_outliers"] = kq.outliers[:L].numpy().copy()
    out["k_outlier_indices"] = kq.outlier_indices[:L].numpy().copy()
    out["v_outliers"] = vq.outliers[:L].numpy().copy()
    out["v_outlier_indices"] = vq.outlier_indices[:L].numpy().copy()
    out["v_lookup_table"] = vq.lookup_table[:L].numpy().copy()
    if sinks > 0:
        out["kcache_fp16"] = att.kcache_fp16.numpy().copy()
        out["vcache_fp16"] = att.vcache_fp16.numpy().copy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "L=%d" % L)


def main():
    ML, LlamaConfig = import_reference()
    run(ML, LlamaConfig, "ref_attention_nuq4", 4, 0, 40, 6, seed=11)
    run(ML, LlamaConfig, "ref_attention_nuq3_sink5", 3, 5, 40, 6, seed=12)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own QuantK / QuantV
Python classes (deployment/transformers/.../modeling_llama.py:318-1385) on CPU.

Runs only in the build container (needs /root/reference); the resulting
fixtures are committed and are what travels to the GPU box.

How the reference is run here: the class source is exec'd verbatim from the
reference file (never copied into this repo), ``Tensor.cuda()`` is patched to
a no-op and the ``quant_cuda`` extension it calls is replaced by
oracle.quant_cuda_ref (the C restatement of the CUDA kernels).  So these
fixtures pin the HOST GLUE of the reference (LUT construction incl. fp16
rounding, top-k outlier selection, the <=1 zeroing rule, index sort, per-token
V LUT rows, buffer bookkeeping) bit-for-bit, with the kernel arithmetic being
the oracle's restatement (the CUDA kernels themselves cannot run here).

usage: python tests/golden/gen_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import quant_cuda_ref  # noqa: E402

REF = "/root/reference/deployment/transformers/src/transformers/models/llama/modeling_llama.py"
OUT = os.path.dirname(os.path.abspath(__file__))

H, HD, C = 32, 128, 4096
THETA = 10000.0


def load_reference_classes():
    src = open(REF).read()
    a = src.index("def compute_lut(")
    b = src.index("class LlamaAttention(nn.Module):")
    ns = {"torch": torch, "nn": nn, "math": math, "np": np, "quant_cuda": quant_cuda_ref.as_module()}
    torch.Tensor.cuda = lambda self, *a, **k: self  # CPU stand-in for the allocation calls
    import warnings
    warnings.filterwarnings("ignore")
    exec(compile(src[a:b], REF, "exec"), ns)
    return ns["QuantK"], ns["QuantV"]


def nf_centroids(bits):
    """Deterministic NUQ signposts in [-1,1]: normal quantiles (fp32, shape (n,1)
    like sklearn's cluster_centers_), deliberately unsorted."""
    n = 2 ** bits
    p = (np.arange(n, dtype=np.float64) + 0.5) / n
    from scipy.stats import norm
    c = norm.ppf(p)
    c = (c / np.abs(c).max() * 0.97).astype(np.float32)
    rng = np.random.RandomState(bits)
    return c[rng.permutation(n)].reshape(n, 1)


def synth(seed, S):
    """K/V-like activations: N(0,1) with a per-channel scale on K and ~1 % heavy
    tails, rounded to fp16 (the model's activation dtype)."""
    g = torch.Generator().manual_seed(seed)
    scale = torch.exp(0.5 * torch.randn(C, generator=g))
    shift = 0.3 * torch.randn(C, generator=g)
    k = torch.randn(S, C, generator=g) * scale + shift
    v = torch.randn(S, C, generator=g)
    for x in (k, v):
        m = torch.rand(S, C, generator=g) < 0.01
        x[m] *= 6.0
    q = torch.randn(S, C, generator=g)
    return k.half(), v.half(), q.half(), scale, shift


def quantizer_for(k_calib, bits):
    """(upper[1,C], lower[1,C], [centroids(n,1)]) as SimQuant.quantize returns
    (quant/kvquant/simquant_module_quantizer.py:465-474, 550-555)."""
    x = k_calib.float().numpy()
    upper = np.percentile(x, 99.5, axis=0)[None, :]
    lower = np.percentile(x, 0.5, axis=0)[None, :]
    return (upper, lower, [nf_centroids(bits)])


def run_scenario(QuantK, QuantV, name, bits, include_sparse, sinks, S, steps, seed, orig=False):
    max_len = S + steps + sinks + 3
    T = S + steps
    # torch.topk leaves the choice among EQUAL values unspecified; keep the fixtures free of ties at
    # the 21st/22nd selection boundary so they pin semantics, not a tie-break accident
    while True:
        k_all, v_all, q_all, _, _ = synth(seed, T + sinks)
        vf = v_all.float()
        hi = torch.topk(vf, 23, dim=-1).values
        lo = torch.topk(vf, 23, dim=-1, largest=False).values
        if bool((hi[:, 20] != hi[:, 21]).all() and (lo[:, 20] != lo[:, 21]).all()):
            break
        seed += 100
    print(name, "seed", seed)
    calib, _, _, _, _ = synth(seed % 100 + 1000, 256)
    quant = quantizer_for(calib, bits)

    kc = QuantK(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                include_sparse=include_sparse, sparsity_threshold=0.99, rope_theta=THETA,
                first_few_fp16=sinks, use_orig_sparse=orig)
    vc = QuantV(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len,
                include_sparse=include_sparse, sparsity_threshold=0.99, first_few_fp16=sinks)
    kc.load_lookup_table(quant, include_sparse=include_sparse, sparsity_threshold=0.99)
    vc.load_lookup_table(quant, include_sparse=include_sparse, sparsity_threshold=0.99)

    out = {"bits": bits, "include_sparse": int(include_sparse), "sinks": sinks, "S": S, "steps": steps,
           "orig": int(orig),
           "max_len": max_len, "theta": THETA,
           "q_upper": quant[0].astype(np.float32), "q_lower": quant[1].astype(np.float32),
           "q_centroids": quant[2][0],
           "k_all": k_all.numpy(), "v_all": v_all.numpy(), "q_all": q_all.numpy(),
           "k_lookup_table": kc.lookup_table.numpy().copy(),
           "k_thr_upper": kc.outlier_threshold_upper.numpy().copy(),
           "k_thr_lower": kc.outlier_threshold_lower.numpy().copy(),
           "v_lut": vc.lut.numpy().copy()}

    thr = int(((1 - 0.99) / 2) * C) + 2  # modeling_llama.py:1539
    # ---- prefill: parallel pack of tokens [sinks, sinks+S) -----------------
    if include_sparse and S > 0:
        ks = k_all[sinks:sinks + S].view(S, H, HD).permute(1, 2, 0)      # [H, hd, S] like ML:1913
        vs = v_all[sinks:sinks + S].view(S, H, HD).permute(1, 2, 0)
        if orig:
            kc.parallel_pack_orig(ks)                                    # ML:1886-1887
        else:
            kc.parallel_pack(ks)
        kc.klen += sinks                                                 # ML:1890
        vflat = v_all[sinks:sinks + S].float()
        uv, ui = torch.topk(vflat, thr, dim=-1)                          # ML:1556-1557
        lv, li = torch.topk(vflat, thr, dim=-1, largest=False)
        vc.parallel_pack(vs, uv, ui, lv, li)
        vc.vlen += sinks
    else:
        kc.klen += sinks
        vc.vlen += sinks
        S = 0
    # ---- decode steps -------------------------------------------------------
    for i in range(steps):
        t = sinks + S + i
        q = q_all[t].view(H, 1, HD)                                      # already "post-RoPE"
        k = k_all[t].view(1, H, 1, HD)
        scores = kc.forward_fused_sparse_orig(q, k) if orig else kc.forward_fused_sparse(q, k)   # [H,1,L] half
        out["score_%d" % i] = scores.numpy().copy()
        aw = (scores.unsqueeze(0) / math.sqrt(HD))                       # ML:1972-1973 (fp16)
        aw = torch.softmax(aw, dim=-1, dtype=torch.float32).to(torch.float16).squeeze(0)
        v = v_all[t].view(1, H, 1, HD)
        if include_sparse:
            vf = v.flatten().float()
            uv, ui = torch.topk(vf, thr)                                  # ML:1540-1541
            lv, li = torch.topk(vf, thr, largest=False)
            o = vc.forward_fused_sparse(aw, v, uv, ui, lv, li)
        else:
            o = vc.forward_fused_sparse(aw, v, None, None, None, None)
        out["attn_%d" % i] = o.numpy().copy()
        out["prob_%d" % i] = aw.numpy().copy()

    L = kc.klen - sinks
    out["L"] = L
    out["kcache"] = kc.kcache[:, :, :L].numpy().copy()
    out["vcache"] = vc.vcache[:, :, :L].numpy().copy()
    out["v_lookup_table"] = vc.lookup_table[:L].numpy().copy()
    if orig:
        out["k_rows"] = kc.rows.numpy().copy()
        out["k_cols"] = kc.cols.numpy().copy()
        out["k_vals"] = kc.vals.numpy().copy()
        out["k_start_rows"] = kc.start_rows.numpy().copy()
        out["k_num_threads"] = int(kc.num_threads)
    elif include_sparse:
        out["k_outliers"] = kc.outliers[:L].numpy().copy()
        out["k_outlier_indices"] = kc.outlier_indices[:L].numpy().copy()
        out["v_outliers"] = vc.outliers[:L].numpy().copy()
        out["v_outlier_indices"] = vc.outlier_indices[:L].numpy().copy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "L=%d" % L)


def main():
    QuantK, QuantV = load_reference_classes()
    run_scenario(QuantK, QuantV, "ref_nuq4_sparse", 4, True, 0, 6, 2, seed=4)
    run_scenario(QuantK, QuantV, "ref_nuq3_sparse_sink5", 3, True, 5, 6, 2, seed=3)
    run_scenario(QuantK, QuantV, "ref_nuq2_sparse", 2, True, 0, 5, 2, seed=2)
    run_scenario(QuantK, QuantV, "ref_nuq4_dense", 4, False, 0, 0, 3, seed=14)
    run_scenario(QuantK, QuantV, "ref_nuq4_orig", 4, True, 0, 5, 3, seed=24, orig=True)


if __name__ == "__main__":
    main()

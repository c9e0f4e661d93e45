"""On-box calibration of the KVQuant quantizers: what `SimQuant.quantize` produces from captured k_proj / v_proj
outputs (SQ = quant/kvquant/simquant_module_quantizer.py:400-555, driven by quant/llama_simquant.py:149-295), so that
a model can be quantised without the reference environment.

    quantizers = calibrate_llama(model, calib_ids, bits=4, include_sparse=True, sparsity_threshold=0.99)
    pickle.dump(quantizers, open("quantizers.pickle", "wb"))      # the reference's format (SQ:550-555)

Per projection: per-channel thresholds = the (1-t)/t percentiles of the calibration activations with
t = 1 - (1 - sparsity_threshold)/2 (SQ:413-416, 465-466); NUQ signposts = 1-d k-means (2^bits clusters,
sklearn KMeans(random_state=0, n_init="auto", max_iter=50), SQ:511-528) on the activations shifted and normalised to
[-1, 1] per channel with the threshold-crossers removed (SQ:476-505), optionally weighted by Fisher information
(SQ:507-518); Q-Norm scale / offset (SQ:531-548, with the reference's `return_freq` call made to work).  V uses the
same per-channel pass for its signposts only -- its thresholds are recomputed per token at run time.
Host code (numpy / sklearn on the CPU), run once per model.
"""
import math

import numpy as np
import torch


def _nearest(x, poles):
    d = (x.unsqueeze(0) - poles.view(-1, 1)).abs()
    return poles[d.argmin(dim=0)]


def calibrate_tensor(data, bits, include_sparse=True, sparsity_threshold=0.99, fisher=None, norm=False, seed=0,
                     max_points=2_000_000, qchannel=0):
    """data: [tokens, channels] activations of one projection.  qchannel = 0: per-channel thresholds (K, SimQuant(
    perchannel=True, qchannel=0)); qchannel = -1: per-token thresholds (V, llama_simquant.py:229-235 -- only its
    signposts are used at run time).  Returns the quantizer tuple (upper, lower, [centroids (2^bits, 1)])
    (+ normscale, normoffset tensors with norm); upper / lower are [1, C] (K) or [tokens, 1] (V)."""
    from sklearn.cluster import KMeans
    x = torch.as_tensor(data).float().cpu()
    t = 1 - ((1 - sparsity_threshold) / 2) if include_sparse else 1.0
    xn = x.numpy()
    upper = np.percentile(xn, t * 100, axis=qchannel)
    lower = np.percentile(xn, (1 - t) * 100, axis=qchannel)
    up, lo = torch.tensor(upper).float().unsqueeze(qchannel), torch.tensor(lower).float().unsqueeze(qchannel)
    zero_point = (up + lo) / 2
    distance = (up - lo) / 2
    shifted = (x - zero_point) / distance                      # normalised to [-1, 1] (SQ:476-480)
    outlier = torch.logical_or(shifted > 1, shifted < -1)
    flat = shifted.flatten()
    keep = ~outlier.flatten()
    pts = flat[keep].numpy().reshape(-1, 1)
    w = None
    if fisher is not None:
        w = torch.as_tensor(fisher).float().flatten()[keep].numpy()
    if pts.shape[0] > max_points:                              # (k-means on a fixed-seed subsample of very large sets)
        idx = np.random.RandomState(seed).choice(pts.shape[0], max_points, replace=False)
        pts = pts[idx]
        w = None if w is None else w[idx]
    km = KMeans(n_clusters=2 ** bits, random_state=seed, n_init="auto", max_iter=50).fit(pts, sample_weight=w)
    centroids = [km.cluster_centers_]
    q = (np.expand_dims(upper, qchannel).astype(np.float32), np.expand_dims(lower, qchannel).astype(np.float32), centroids)
    if norm:
        cent = torch.tensor(centroids[0]).flatten().float()
        ok = (~outlier).float()
        cnt = ok.sum()
        m1 = (shifted * ok).sum() / cnt
        s1 = torch.sqrt((((shifted - m1) * ok) ** 2).sum() / cnt)
        aug = _nearest(flat, cent).reshape(shifted.shape)
        m2 = (aug * ok).sum() / cnt
        s2 = torch.sqrt((((aug - m2) * ok) ** 2).sum() / cnt)
        q = q + (s1 / s2, (-m2) * (s1 / s2) + m1)
    return q


@torch.no_grad()
def capture_kv(model, input_ids):
    """k_proj / v_proj outputs of every decoder layer for the calibration tokens (llama_simquant.py:214-243 hooks
    the same two Linear modules): {"model.layers.N.self_attn.k_proj": [tokens, C] f32, ...}"""
    base = model.model if hasattr(model, "model") else model
    store, hooks = {}, []
    for i, layer in enumerate(base.layers):
        for nm in ("k_proj", "v_proj"):
            key = "model.layers.%d.self_attn.%s" % (i, nm)

            def hook(mod, inp, out, key=key):
                store.setdefault(key, []).append(out.detach().float().reshape(-1, out.shape[-1]).cpu())
            hooks.append(getattr(layer.self_attn, nm).register_forward_hook(hook))
    for ids in (input_ids if isinstance(input_ids, (list, tuple)) else [input_ids]):
        model(ids.to(next(model.parameters()).device), use_cache=False)
    for h in hooks:
        h.remove()
    return {k: torch.cat(v, dim=0) for k, v in store.items()}


def calibrate_llama(model, input_ids, bits=4, include_sparse=True, sparsity_threshold=0.99, norm=False, seed=0):
    """run the (un-patched) model on the calibration tokens and fit every layer's K and V quantizer.  Returns the dict
    the reference pickles (SQ:550-555 / llama_simquant.py:285-295)."""
    acts = capture_kv(model, input_ids)
    return {k: calibrate_tensor(v, bits, include_sparse, sparsity_threshold, norm=norm, seed=seed,
                                qchannel=0 if "k_proj" in k else -1)
            for k, v in acts.items()}

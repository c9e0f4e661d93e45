"""On-box calibration of the KVQuant quantizers: what `SimQuant.quantize` produces from captured k_proj / v_proj
outputs (SQ = quant/kvquant/simquant_module_quantizer.py:400-555, driven by quant/llama_simquant.py:149-295), so that
a model can be quantised without the reference environment.

    quantizers = calibrate_llama(model, calib_ids, bits=4, include_sparse=True, sparsity_threshold=0.99,
                                 cap_outliers=True, first_few_fp16=1)
    pickle.dump(quantizers, open("quantizers.pickle", "wb"))      # the reference's format (SQ:550-555)

Per projection (`calibrate_tensor`):
  * thresholds = the (1-t)/t percentiles of the calibration activations with t = 1 - (1 - sparsity_threshold)/2, per
    channel for K (qchannel = 0) and per token for V (qchannel = -1) (SQ:413-416, 465-474);
  * NUQ signposts = 1-d k-means (2^bits clusters, sklearn KMeans(random_state=0, n_init="auto", max_iter=50),
    SQ:511-528) on the activations shifted and normalised to [-1, 1] with the outliers masked out (SQ:476-505),
    optionally weighted by Fisher information (`fisher`: same shape as the data, SQ:507-518);
  * `cap_outliers` (K only, llama_simquant.py:262-269): the mask that keeps values out of the k-means fit comes from
    the capped-outlier rule (SQ:421-461) -- per token the ceil((1-t) C) entries with the largest AND the smallest
    normalised magnitude are replaced by the channel median, the per-channel min / max of what is left re-normalise
    the data, and everything with |normalised| > 1 is masked.  (As in the reference, the thresholds that are
    RETURNED stay the plain percentiles: SQ:463-466 recompute them after the capped block.)
  * `first_few_fp16` > -1: the first tokens of every calibration sample are attention sinks kept in fp16 at run time
    and are masked out of the fit (SQ:441-446, 487-491; samples are `seqlen` rows each -- the reference hard-codes 2048);
  * Q-Norm scale / offset (SQ:531-548, with the reference's `return_freq` call made to work).
V uses the per-token pass for its signposts only -- its thresholds are recomputed per token at run time.
Host code (numpy / sklearn on the CPU), run once per model.  `max_points` (default: no limit, like the reference)
subsamples the k-means input with a fixed seed and says so.
"""
import math
import sys

import numpy as np
import torch


def _nearest(x, poles, chunk=1 << 22):
    """round every element of x to the nearest pole (round_to_nearest_pole_sim, SQ:10-28), in chunks: the
    reference materialises a [2^bits, N] distance stack"""
    flat = x.reshape(-1)
    out = torch.empty_like(flat)
    p = poles.view(-1, 1)
    for s in range(0, flat.numel(), chunk):
        seg = flat[s:s + chunk]
        out[s:s + chunk] = poles[(seg.unsqueeze(0) - p).abs().argmin(dim=0)]
    return out.reshape(x.shape)


def _mask_first_tokens(mask, first_few_fp16, nsamples, seqlen):
    for i in range(nsamples):
        mask[i * seqlen:i * seqlen + first_few_fp16, :] = True


def calibrate_tensor(data, bits, include_sparse=True, sparsity_threshold=0.99, fisher=None, norm=False, seed=0,
                     max_points=None, qchannel=0, cap_outliers=False, first_few_fp16=-1, nsamples=None, seqlen=2048,
                     log=None):
    """data: [tokens, channels] activations of one projection (nsamples * seqlen rows when first_few_fp16 is used).
    qchannel = 0: per-channel thresholds (K, SimQuant(perchannel=True, qchannel=0)); qchannel = -1: per-token
    thresholds (V, llama_simquant.py:229-235 -- only its signposts are used at run time).  Returns the quantizer tuple
    (upper, lower, [centroids (2^bits, 1)]) (+ normscale, normoffset tensors with norm); upper / lower are [1, C] (K)
    or [tokens, 1] (V)."""
    from sklearn.cluster import KMeans
    x = torch.as_tensor(data).float().cpu()
    t = 1 - ((1 - sparsity_threshold) / 2) if include_sparse else 1.0
    xn = x.numpy()
    if nsamples is None:
        nsamples = max(1, x.shape[0] // seqlen)
    outlier = None
    if cap_outliers:
        # SQ:421-461 (per-channel quantisation only: the caller passes cap_outliers for k_proj)
        up0 = torch.tensor(np.percentile(xn, t * 100, axis=qchannel)).unsqueeze(qchannel)
        lo0 = torch.tensor(np.percentile(xn, (1 - t) * 100, axis=qchannel)).unsqueeze(qchannel)
        d2 = ((x - (up0 + lo0) / 2) / ((up0 - lo0) / 2)).abs()
        num = math.ceil((1 - t) * x.shape[-1])
        m0 = torch.zeros_like(d2, dtype=torch.bool)
        m0.scatter_(-1, torch.topk(d2, num, largest=False).indices, True)
        m0.scatter_(-1, torch.topk(d2, num).indices, True)
        if first_few_fp16 > -1:
            _mask_first_tokens(m0, first_few_fp16, nsamples, seqlen)
        med = torch.median(x, dim=0).values.unsqueeze(0).expand_as(x)
        trimmed = torch.where(m0, med, x)
        up1 = torch.max(trimmed, dim=qchannel).values
        lo1 = torch.min(trimmed, dim=qchannel).values
        zp1 = ((up1 + lo1) / 2).unsqueeze(0)
        dist1 = ((up1 - lo1) / 2).unsqueeze(0)
        dsn = ((x - zp1) / dist1).abs()
        outlier = torch.logical_or(dsn > 1, dsn < -1)
    upper = np.percentile(xn, t * 100, axis=qchannel)
    lower = np.percentile(xn, (1 - t) * 100, axis=qchannel)
    up, lo = torch.tensor(upper).float().unsqueeze(qchannel), torch.tensor(lower).float().unsqueeze(qchannel)
    zero_point = (up + lo) / 2
    distance = (up - lo) / 2
    shifted = (x - zero_point) / distance                      # normalised to [-1, 1] (SQ:476-480)
    if outlier is None:
        outlier = torch.logical_or(shifted > 1, shifted < -1)
    if first_few_fp16 > -1:
        _mask_first_tokens(outlier, first_few_fp16, nsamples, seqlen)
    flat = shifted.flatten()
    keep = ~outlier.flatten()
    pts = flat[keep].numpy().reshape(-1, 1)
    w = None
    if fisher is not None:
        w = torch.as_tensor(fisher).float().flatten()[keep].numpy()
    if max_points is not None and pts.shape[0] > max_points:   # (opt-in: the reference fits all points)
        idx = np.random.RandomState(seed).choice(pts.shape[0], max_points, replace=False)
        (log or (lambda m: print(m, file=sys.stderr)))(
            "calibrate: k-means on a %d-point subsample of %d (max_points)" % (max_points, pts.shape[0]))
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
        aug = _nearest(shifted, cent)
        m2 = (aug * ok).sum() / cnt
        s2 = torch.sqrt((((aug - m2) * ok) ** 2).sum() / cnt)
        q = q + (s1 / s2, (-m2) * (s1 / s2) + m1)
    return q


@torch.no_grad()
def capture_kv(model, input_ids):
    """k_proj / v_proj outputs of every decoder layer for the calibration tokens (llama_simquant.py:214-243 hooks
    the same two Linear modules): {"model.layers.N.self_attn.k_proj": [tokens, C] f32, ...}; `input_ids`: one
    [1, seqlen] tensor or a list of them (the calibration samples, in order)."""
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


def calibrate_llama(model, input_ids, bits=4, include_sparse=True, sparsity_threshold=0.99, norm=False, seed=0,
                    cap_outliers=False, first_few_fp16=-1, fisher=None, max_points=None, seqlen=None):
    """run the (un-patched) model on the calibration samples and fit every layer's K and V quantizer.  Returns the dict
    the reference pickles (SQ:550-555 / llama_simquant.py:285-295).  input_ids: a [1, seqlen] tensor or a list of
    them (nsamples x seqlen, llama_simquant.py:149-176); cap_outliers applies to k_proj only (:262-269);
    first_few_fp16 > -1 excludes the sink tokens of every sample from the fit; fisher: {"<key>.weight": tensor shaped
    like that projection's activations} (the reference's gradient-square files) or None; max_points: opt-in subsample
    of the k-means input."""
    samples = input_ids if isinstance(input_ids, (list, tuple)) else [input_ids]
    if seqlen is None:
        seqlen = int(samples[0].shape[-1])
    acts = capture_kv(model, samples)
    out = {}
    for k, v in acts.items():
        is_k = "k_proj" in k
        f = None if fisher is None else fisher.get(k + ".weight")
        out[k] = calibrate_tensor(v, bits, include_sparse, sparsity_threshold, fisher=f, norm=norm, seed=seed,
                                  max_points=max_points, qchannel=0 if is_k else -1,
                                  cap_outliers=bool(cap_outliers) and is_k, first_few_fp16=first_few_fp16,
                                  nsamples=len(samples), seqlen=seqlen)
    return out

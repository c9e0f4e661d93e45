"""CPU restatement of the reference's SIMULATED-quantisation path
(SQ = /root/reference/quant/kvquant/simquant_module_quantizer.py): the
fake-quant functions `QuantLinearSim.forward` applies to the k_proj / v_proj
outputs, which is the reference's CPU-runnable path (BASELINE config 1) and the
arithmetic bench.py's cpu_baseline times.

TEST INFRASTRUCTURE ONLY (checker for the config-1 perplexity-delta harness and
the CPU baseline); nothing under kvquant_amd/ imports it.  Pinned to the
reference's own functions by tests/test_simquant_cpu.py: bit-exact against the
reference module imported from /root/reference when it is there, and against
tests/golden/simquant_ref.npz (generated from it by tests/golden/gen_simquant.py)
everywhere else.

Everything is plain torch and device-agnostic, like the reference functions.
"""
import torch


def round_to_nearest_pole_sim(w, poles):
    """SQ:10-28: every element of the 1-d `w` replaced by the nearest pole (first minimum wins: argmin)."""
    diff = torch.stack([(w - c).abs() for c in poles])
    idx = diff.argmin(axis=0)
    aug = 0
    for i, c in enumerate(poles):
        aug = aug + (idx == i) * c
    return aug


def get_outliers(w, channel=-1, outlier_threshold_upper=None, outlier_threshold_lower=None, cap_outliers=-1,
                 first_few_fp16=-1):
    """SQ:30-79 (per-channel thresholds; capped variant keeps the 21 largest / smallest normalised
    threshold-crossers per token -- the reference hard-codes 21)."""
    up = outlier_threshold_upper.unsqueeze(channel)
    lo = outlier_threshold_lower.unsqueeze(channel)
    outlier_mask = torch.logical_or(w < lo, w > up)
    if cap_outliers > -1:
        zero_point = (up + lo) / 2
        distance = (up - lo) / 2
        outliers = w * outlier_mask
        values = torch.zeros_like(outliers)
        values[outlier_mask] = ((w - zero_point) / distance)[outlier_mask]
        uv, ui = torch.topk(values, 21, dim=-1)
        lv, li = torch.topk(values, 21, dim=-1, largest=False)
        values2 = torch.zeros_like(outliers)
        values2.scatter_(-1, torch.cat((ui, li), dim=-1), torch.cat((uv, lv), dim=-1))
        outlier_mask = values2 != 0
    if first_few_fp16 > -1:
        outlier_mask[:first_few_fp16, :] = True
    return outlier_mask


def get_outliers_dynamic(w, channel=-1, thresh=0.999, first_few_fp16=-1):
    """SQ:81-113 (per-token quantile thresholds, inclusive comparisons)."""
    t = 1 - ((1 - thresh) / 2)
    w = w.float()
    up = torch.quantile(w, t, dim=channel).unsqueeze(channel)
    lo = torch.quantile(w, 1 - t, dim=channel).unsqueeze(channel)
    outlier_mask = torch.logical_or(w <= lo, w >= up)
    if first_few_fp16 > -1:
        outlier_mask[:first_few_fp16, :] = True
    return outlier_mask


def quant_fn_nuq_recon(inp, bits=8, qchannel=-1, dynamicquantization=False, include_sparse=False, outlier_mask=None,
                       maxval=-1, minval=-1, lut=None, norm=False, normscale=None, normoffset=None, first_few_fp16=-1):
    """SQ:265-361: simulated NUQ quantisation of a [tokens, channels] matrix."""
    if first_few_fp16 > -1:
        orig = inp
    if dynamicquantization:
        if include_sparse:
            outliers = inp * outlier_mask
            median = torch.median(inp, dim=qchannel).values.unsqueeze(qchannel)
            tmp_inp = inp - outliers + median * outlier_mask
            maxval = torch.max(tmp_inp, dim=qchannel).values
            minval = torch.min(tmp_inp, dim=qchannel).values
        else:
            maxval = torch.max(inp, dim=qchannel).values
            minval = torch.min(inp, dim=qchannel).values
    offset = ((maxval + minval) / 2).unsqueeze(qchannel)
    rangeval = ((maxval - minval) / 2).unsqueeze(qchannel)
    inp = inp - offset
    if include_sparse:
        outliers = inp * outlier_mask
        inp = inp - outliers
    inp_scaled = inp / rangeval
    lut_t = torch.as_tensor(lut[0]).to(inp_scaled.device)
    Q = round_to_nearest_pole_sim(inp_scaled.flatten(), lut_t)
    qinp_out = Q.reshape(inp.shape).float().to(inp_scaled.device)
    if norm:
        qinp_out = qinp_out * normscale.to(inp_scaled.device) + normoffset.to(inp_scaled.device)
    qinp_out = qinp_out * rangeval
    if include_sparse:
        qinp_out[outlier_mask] = 0
        qinp_out = qinp_out + outliers
    qinp_out = qinp_out + offset
    qinp_out = torch.nan_to_num(qinp_out, nan=0.0, posinf=0.0, neginf=0.0)
    if first_few_fp16 > -1:
        qinp_out[:first_few_fp16, :] = orig[:first_few_fp16, :]
    return qinp_out.float()


def fake_quant_k(y, quantizer, bits, include_sparse=True, cap_outliers=21, first_few_fp16=-1, norm=False):
    """QuantLinearSim.forward (SQ:700-795) for a k_proj output y [tokens, C] (per-channel: qchannel = 0,
    static thresholds rounded to fp16 as SQ:612-613)."""
    up = torch.as_tensor(quantizer[0]).to(y.device).flatten().half()
    lo = torch.as_tensor(quantizer[1]).to(y.device).flatten().half()
    y = y.float()
    mask = get_outliers(y, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo,
                        cap_outliers=cap_outliers, first_few_fp16=first_few_fp16) if include_sparse else None
    return quant_fn_nuq_recon(y, bits=bits, qchannel=0, maxval=up, minval=lo, include_sparse=include_sparse,
                              outlier_mask=mask, dynamicquantization=False, lut=quantizer[2], norm=norm,
                              normscale=quantizer[3] if norm else None, normoffset=quantizer[4] if norm else None,
                              first_few_fp16=first_few_fp16).half()


def fake_quant_v(y, quantizer, bits, include_sparse=True, sparsity_threshold=0.99, first_few_fp16=-1, norm=False):
    """the same for a v_proj output (per-token: qchannel = -1, dynamic thresholds)."""
    y = y.float()
    mask = get_outliers_dynamic(y, channel=-1, thresh=sparsity_threshold, first_few_fp16=first_few_fp16) \
        if include_sparse else None
    return quant_fn_nuq_recon(y, bits=bits, qchannel=-1, include_sparse=include_sparse, outlier_mask=mask,
                              dynamicquantization=True, lut=quantizer[2], norm=norm,
                              normscale=quantizer[3] if norm else None, normoffset=quantizer[4] if norm else None,
                              first_few_fp16=first_few_fp16).half()

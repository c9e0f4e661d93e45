"""kvquant_amd.calibrate against the reference's SimQuant.quantize (imported live from /root/reference when it is
there): identical thresholds, the same k-means signposts; plus format / property checks that run everywhere."""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from kvquant_amd import calibrate

REF = "/root/reference/quant/kvquant/simquant_module_quantizer.py"


def _data(seed=0, T=2048, C=256):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, C, generator=g) * torch.exp(0.5 * torch.randn(C, generator=g)) + 0.2 * torch.randn(C, generator=g)
    x[torch.rand(T, C, generator=g) < 0.01] *= 5
    return x


def test_quantizer_tuple_format_and_properties():
    x = _data()
    q = calibrate.calibrate_tensor(x, 3, True, 0.99, norm=True)
    up, lo, cent, ns, no = q
    assert up.shape == (1, 256) and lo.shape == (1, 256) and cent[0].shape == (8, 1)
    frac = float(((x > torch.tensor(up)) | (x < torch.tensor(lo))).float().mean())
    assert 0.005 < frac < 0.015                                # 1 % of the entries cross the thresholds
    c = np.sort(cent[0].flatten())
    assert -1.0 <= c[0] < c[-1] <= 1.0                         # signposts live in the normalised range
    assert 0.5 < float(ns) < 2.0 and abs(float(no)) < 0.5


@pytest.mark.skipif(not os.path.exists(REF), reason="no /root/reference here")
@pytest.mark.parametrize("bits", [2, 4])
def test_matches_reference_simquant(bits):
    spec = importlib.util.spec_from_file_location("ref_sq", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    x = _data(seed=bits)
    sq = ref.SimQuant(nn.Linear(8, 256, bias=False), bits, perchannel=True, qchannel=0)
    sq.add_batch(None, x.clone())
    r = sq.quantize(include_sparse=True, sparsity_threshold=0.99, nuq=True, fisher=None, norm=False,
                    cap_outliers=False, first_few_fp16=-1)
    m = calibrate.calibrate_tensor(x, bits, True, 0.99)
    assert np.array_equal(np.asarray(r[0]).flatten(), m[0].flatten())
    assert np.array_equal(np.asarray(r[1]).flatten(), m[1].flatten())
    assert np.abs(np.sort(r[2][0].flatten()) - np.sort(m[2][0].flatten())).max() < 1e-4


def _ref_module():
    spec = importlib.util.spec_from_file_location("ref_sq", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def _same(r, m, tol=1e-4):
    assert np.array_equal(np.asarray(r[0]).flatten(), m[0].flatten())
    assert np.array_equal(np.asarray(r[1]).flatten(), m[1].flatten())
    assert np.abs(np.sort(r[2][0].flatten()) - np.sort(m[2][0].flatten())).max() < tol


@pytest.mark.skipif(not os.path.exists(REF), reason="no /root/reference here")
def test_capped_outliers_and_sink_exclusion_match_reference():
    """SQ:421-461 + 441-446 / 487-491: the k-means mask of the capped-outlier K calibration with the first tokens of
    every sample excluded (the reference hard-codes 16 samples x 2048 tokens = 32768 rows)."""
    ref = _ref_module()
    x = _data(seed=5, T=32768, C=24)
    sq = ref.SimQuant(nn.Linear(8, 24, bias=False), 2, perchannel=True, qchannel=0)
    for i in range(16):
        sq.add_batch(None, x[i * 2048:(i + 1) * 2048].clone().unsqueeze(0))
    assert sq.nsamples == 16
    r = sq.quantize(include_sparse=True, sparsity_threshold=0.99, nuq=True, fisher=None, norm=False,
                    cap_outliers=True, first_few_fp16=5)
    m = calibrate.calibrate_tensor(x, 2, True, 0.99, cap_outliers=True, first_few_fp16=5, nsamples=16, seqlen=2048)
    _same(r, m)
    # the exclusion matters: without it the fit is a different one
    m2 = calibrate.calibrate_tensor(x, 2, True, 0.99, cap_outliers=False, first_few_fp16=-1)
    assert np.abs(np.sort(m2[2][0].flatten()) - np.sort(m[2][0].flatten())).max() > 1e-6


@pytest.mark.skipif(not os.path.exists(REF), reason="no /root/reference here")
def test_fisher_weighted_kmeans_matches_reference():
    """SQ:507-518: the squared-gradient weights enter the k-means fit as sample weights"""
    ref = _ref_module()
    x = _data(seed=7, T=2048, C=64)
    g = torch.Generator().manual_seed(8)
    fisher = torch.rand(2048, 64, generator=g) ** 4 + 1e-4
    sq = ref.SimQuant(nn.Linear(8, 64, bias=False), 3, perchannel=True, qchannel=0)
    sq.add_batch(None, x.clone())
    r = sq.quantize(include_sparse=True, sparsity_threshold=0.99, nuq=True, fisher=fisher, norm=False,
                    cap_outliers=False, first_few_fp16=-1)
    m = calibrate.calibrate_tensor(x, 3, True, 0.99, fisher=fisher)
    _same(r, m)
    plain = calibrate.calibrate_tensor(x, 3, True, 0.99)
    assert np.abs(np.sort(plain[2][0].flatten()) - np.sort(m[2][0].flatten())).max() > 1e-4


@pytest.mark.skipif(not os.path.exists(REF), reason="no /root/reference here")
def test_value_pass_matches_reference():
    """llama_simquant.py:229-235: v_proj is calibrated with SimQuant(perchannel=True, qchannel=-1) -- per-token
    thresholds [T, 1], signposts from the per-token normalised values"""
    ref = _ref_module()
    x = _data(seed=9, T=512, C=256)
    sq = ref.SimQuant(nn.Linear(8, 256, bias=False), 4, perchannel=True, qchannel=-1)
    sq.add_batch(None, x.clone())
    r = sq.quantize(include_sparse=True, sparsity_threshold=0.99, nuq=True, fisher=None, norm=False,
                    cap_outliers=False, first_few_fp16=-1)
    m = calibrate.calibrate_tensor(x, 4, True, 0.99, qchannel=-1)
    assert m[0].shape == (512, 1)
    _same(r, m)


def test_subsampling_is_opt_in_and_reported():
    x = _data(seed=3, T=1024, C=64)
    msgs = []
    q = calibrate.calibrate_tensor(x, 2, True, 0.99, max_points=5000, log=msgs.append)
    assert len(msgs) == 1 and "subsample" in msgs[0] and q[2][0].shape == (4, 1)
    msgs.clear()
    calibrate.calibrate_tensor(x, 2, True, 0.99, log=msgs.append)
    assert not msgs


def test_nearest_is_chunked():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, 7, generator=g)
    poles = torch.tensor([-1.0, -0.2, 0.3, 0.9])
    a = calibrate._nearest(x, poles, chunk=333)
    b = poles[(x.reshape(1, -1) - poles.view(-1, 1)).abs().argmin(dim=0)].reshape(x.shape)
    assert torch.equal(a, b)

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

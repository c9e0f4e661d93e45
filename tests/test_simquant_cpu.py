"""oracle/simquant.py (restatement of the reference's simulated-quantisation functions) against the reference:
bit-exact vs tests/golden/simquant_ref.npz (made by the reference's own module, tests/golden/gen_simquant.py)
and, where /root/reference exists, vs that module imported live."""
import os

import numpy as np
import pytest
import torch

from oracle import simquant as sq
from tests.golden import gen_simquant as gen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simquant_ref.npz")


def _ours(bits):
    k, v, up, lo, lut = gen.inputs(bits)
    mk = sq.get_outliers(k, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo, cap_outliers=21)
    qk = sq.quant_fn_nuq_recon(k, bits=bits, qchannel=0, maxval=up, minval=lo, include_sparse=True, outlier_mask=mk, lut=lut)
    mv = sq.get_outliers_dynamic(v, channel=-1, thresh=0.99, first_few_fp16=2)
    qv = sq.quant_fn_nuq_recon(v, bits=bits, qchannel=-1, include_sparse=True, outlier_mask=mv, dynamicquantization=True,
                               lut=lut, first_few_fp16=2)
    qn = sq.quant_fn_nuq_recon(k, bits=bits, qchannel=0, maxval=up, minval=lo, include_sparse=True, outlier_mask=mk, lut=lut,
                               norm=True, normscale=torch.tensor(1.05), normoffset=torch.tensor(-0.01))
    return dict(mk=mk, qk=qk, mv=mv, qv=qv, qn=qn)


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_simquant_matches_reference_golden(bits):
    g = np.load(GOLD)
    for name, t in _ours(bits).items():
        ref = g["b%d_%s" % (bits, name)]
        assert np.array_equal(t.numpy(), ref), name


@pytest.mark.skipif(not os.path.exists(gen.REF), reason="no /root/reference here")
def test_simquant_matches_reference_live():
    ref = gen.load_ref()
    k, v, up, lo, lut = gen.inputs(4, T=40, C=512, seed=7)
    mk = ref.get_outliers(k, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo, cap_outliers=21, first_few_fp16=3)
    assert torch.equal(mk, sq.get_outliers(k, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo, cap_outliers=21,
                                           first_few_fp16=3))
    a = ref.quant_fn_nuq_recon(k, bits=4, qchannel=0, maxval=up, minval=lo, include_sparse=True, outlier_mask=mk, lut=lut,
                               first_few_fp16=3)
    b = sq.quant_fn_nuq_recon(k, bits=4, qchannel=0, maxval=up, minval=lo, include_sparse=True, outlier_mask=mk, lut=lut,
                              first_few_fp16=3)
    assert torch.equal(a, b)
    quant = (up.float().numpy()[None], lo.float().numpy()[None], lut)
    # the wrappers reproduce QuantLinearSim.forward's argument plumbing (SQ:700-795)
    mk0 = ref.get_outliers(k, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo, cap_outliers=21)
    a0 = ref.quant_fn_nuq_recon(k, bits=4, qchannel=0, maxval=up, minval=lo, include_sparse=True, outlier_mask=mk0, lut=lut)
    assert torch.equal(sq.fake_quant_k(k, quant, 4, cap_outliers=21), a0.half())
    mv = ref.get_outliers_dynamic(v, channel=-1, thresh=0.99)
    c = ref.quant_fn_nuq_recon(v, bits=4, qchannel=-1, include_sparse=True, outlier_mask=mv, dynamicquantization=True, lut=lut)
    assert torch.equal(sq.fake_quant_v(v, quant, 4, sparsity_threshold=0.99), c.half())

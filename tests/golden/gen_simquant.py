"""Generates tests/golden/simquant_ref.npz from the reference's OWN simulated-quantisation functions
(/root/reference/quant/kvquant/simquant_module_quantizer.py imported here; nothing is copied):
seeded inputs -> get_outliers / get_outliers_dynamic / quant_fn_nuq_recon outputs for the K (per-channel,
static thresholds, capped outliers) and V (per-token, dynamic) configurations at 2, 3 and 4 bit.
Run in the build container:  python tests/golden/gen_simquant.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/quant/kvquant/simquant_module_quantizer.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_simquant", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def inputs(bits, T=24, C=256, seed=0):
    g = torch.Generator().manual_seed(seed + bits)
    scale = torch.exp(0.5 * torch.randn(C, generator=g))
    shift = 0.3 * torch.randn(C, generator=g)
    k = (torch.randn(T, C, generator=g) * scale + shift)
    k[torch.rand(T, C, generator=g) < 0.02] *= 4.0
    v = torch.randn(T, C, generator=g)
    v[torch.rand(T, C, generator=g) < 0.02] *= 5.0
    up = (shift + 2.4 * scale).half()
    lo = (shift - 2.4 * scale).half()
    n = 2 ** bits
    cent = torch.sort(torch.rand(n, generator=g) * 2 - 1).values.numpy().reshape(n, 1)
    return k.half().float(), v.half().float(), up, lo, [cent]


def main():
    ref = load_ref()
    out = {}
    for bits in (2, 3, 4):
        k, v, up, lo, lut = inputs(bits)
        mk = ref.get_outliers(k, channel=0, outlier_threshold_upper=up, outlier_threshold_lower=lo, cap_outliers=21,
                              first_few_fp16=-1)
        qk = ref.quant_fn_nuq_recon(k, bits=bits, qchannel=0, maxval=up, minval=lo, include_sparse=True,
                                    outlier_mask=mk, dynamicquantization=False, lut=lut)
        mv = ref.get_outliers_dynamic(v, channel=-1, thresh=0.99, first_few_fp16=2)
        qv = ref.quant_fn_nuq_recon(v, bits=bits, qchannel=-1, include_sparse=True, outlier_mask=mv,
                                    dynamicquantization=True, lut=lut, first_few_fp16=2)
        qn = ref.quant_fn_nuq_recon(k, bits=bits, qchannel=0, maxval=up, minval=lo, include_sparse=True,
                                    outlier_mask=mk, dynamicquantization=False, lut=lut, norm=True,
                                    normscale=torch.tensor(1.05), normoffset=torch.tensor(-0.01))
        for name, t in (("mk", mk), ("qk", qk), ("mv", mv), ("qv", qv), ("qn", qn)):
            out["b%d_%s" % (bits, name)] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "simquant_ref.npz"), **out)
    print("wrote simquant_ref.npz:", sorted(out))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    main()

"""CPU: the restated attention protocol the GPU tests compare against (tests/test_attention_gpu.py::RefAttention -- the
oracle's QuantK / QuantV glue + the reference's prefill / sink / decode bookkeeping) against fixtures made by EXECUTING the
reference's own LlamaAttention.forward (tests/golden/gen_attention.py, ML:1388-1760): decode outputs equal, cache state
and fp16 sink caches bit for bit.  This is what makes the restatement a pinned checker rather than a reading."""
import os

import numpy as np
import pytest
import torch

from tests import util
from tests.test_attention_gpu import RefAttention, H, HD


@pytest.mark.parametrize("name", ["ref_attention_nuq4", "ref_attention_nuq3_sink5"])
def test_restated_protocol_equals_executed_reference(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    bits, sinks, S, steps, max_len = int(g["bits"]), int(g["sinks"]), int(g["S"]), int(g["steps"]), int(g["max_len"])
    quant = (g["q_upper"], g["q_lower"], [g["q_centroids"]])
    ref = RefAttention(bits, sinks, max_len, quant, float(g["theta"]))

    def st(x, a, b):
        return torch.from_numpy(x[a:b]).view(1, b - a, H, HD).transpose(1, 2).contiguous()

    ctx = torch.from_numpy(g["ctx"]).float()
    o = ref.attend(st(g["q_states"], 0, S), st(g["k_states"], 0, S), st(g["v_states"], 0, S))
    # prefill: the reference's eager branch rounds q.K^T to fp16 before the softmax (ML:1594-1596), the restatement
    # evaluates it in fp32
    assert util.rel_err(o.float().reshape(S, -1), ctx[:S]) < 5e-3
    for i in range(steps):
        t = S + i
        o = ref.attend(st(g["q_states"], t, t + 1), st(g["k_states"], t, t + 1), st(g["v_states"], t, t + 1))
        assert torch.equal(o.reshape(1, -1), torch.from_numpy(g["ctx"][t:t + 1])), i      # decode: the same fp16 values
    L = int(g["L"])
    assert np.array_equal(g["kcache"], ref.k.kcache[:, :, :L].numpy())
    assert np.array_equal(g["vcache"], ref.v.vcache[:, :, :L].numpy())
    assert np.array_equal(g["k_outlier_indices"], ref.k.outlier_indices[:L].numpy())
    assert np.array_equal(g["v_outlier_indices"], ref.v.outlier_indices[:L].numpy())
    assert np.array_equal(g["k_outliers"].view(np.int32), ref.k.outliers[:L].numpy().view(np.int32))
    assert np.array_equal(g["v_outliers"].view(np.int32), ref.v.outliers[:L].numpy().view(np.int32))
    assert np.array_equal(g["v_lookup_table"].view(np.int32), ref.v.lookup_table[:L].numpy().view(np.int32))
    if sinks:
        assert torch.equal(ref.kf, torch.from_numpy(g["kcache_fp16"]))
        assert torch.equal(ref.vf, torch.from_numpy(g["vcache_fp16"]))

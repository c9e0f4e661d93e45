"""CPU: host logic around the hot path that needs no GPU -- the sampling warpers of kvquant_amd.llama.generate against
HF's own logits processors, bench.py's byte accounting and its event sampling."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 50, 1.0), (0.7, 5, 1.0), (1.3, 0, 0.9), (0.8, 12, 0.5), (1.0, 1, 1.0)])
def test_sampling_warpers_match_transformers(temperature, top_k, top_p):
    """generate(do_sample=True) applies temperature, top-k, top-p in HF's order with HF's semantics
    (the reference's lwm/llama_inference.py samples through model.generate with the defaults)"""
    from transformers.generation.logits_process import (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    from kvquant_amd.llama import _warp_logits
    g = torch.Generator().manual_seed(int(temperature * 10) + top_k)
    logits = torch.randn(3, 200, generator=g) * 3
    ref = logits.clone()
    ids = torch.zeros(3, 1, dtype=torch.long)
    if temperature != 1.0:
        ref = TemperatureLogitsWarper(temperature)(ids, ref)
    if top_k:
        ref = TopKLogitsWarper(top_k)(ids, ref)
    if top_p < 1.0:
        ref = TopPLogitsWarper(top_p)(ids, ref)
    out = _warp_logits(logits.clone(), temperature, top_k, top_p)
    assert torch.equal(torch.isinf(out), torch.isinf(ref))
    keep = ~torch.isinf(ref)
    assert torch.allclose(out[keep], ref[keep], rtol=1e-6, atol=0)


def test_bench_byte_accounting():
    """algorithmic HBM bytes per cached token per layer, formats fixed by the boundary"""
    import bench
    L = 131072
    k, per_k = bench.algorithmic_bytes(4, L, "score_k")
    v, per_v = bench.algorithmic_bytes(4, L, "mix_v")
    assert per_k == 2048 + 336 + 128 == 2512 and per_v == 2048 + 336 + 64 + 128 == 2576
    # + the L-independent extras of a layer step (K LUT 256 KiB + q 16 KiB; out 16 KiB)
    assert k == L * 2512 + 32 * 128 * 16 * 4 + 32 * 128 * 4 and v == L * 2576 + 32 * 128 * 4
    assert bench.algorithmic_bytes(3, L, "score_k")[1] == 1536 + 336 + 128
    assert bench.algorithmic_bytes(3, L, "mix_v")[1] == 1536 + 336 + 32 + 128
    # opt-in compact outlier entries: 4 instead of 8 bytes per entry
    assert bench.algorithmic_bytes(4, L, "score_k", compact=True)[1] == 2512 - 168
    assert bench.algorithmic_bytes(4, L, "mix_v", compact=True)[1] == 2576 - 168
    # 60 % of 8 TB/s on the score kernel at 128K is the 69 us of DESIGN.md section 3
    assert math.isclose(k / (0.6 * 8e12) * 1e6, 68.6, abs_tol=0.3)


def test_bench_event_sampling_covers_all_layers():
    """--time-every 11 (default) on a 32-layer stack: successive steps bracket different layers, every layer gets its turn"""
    import bench
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        every = bench.parse().time_every
    finally:
        sys.argv = argv
    assert every >= 1 and math.gcd(every, 32) == 1
    timed = {(c % 32) for c in range(1, 32 * every + 1) if c % every == 0}
    assert timed == set(range(32))


def test_quant_cuda_shadow_mirror_decisions():
    """which rows kvquant_amd.quant_cuda transposes into its shadow mirror before an `_opt2` call: one
    row after the reference glue's append (one in-place write per array, ML:748-749), everything whenever older rows may
    have changed.  Host logic only: CPU tensors, a recorder instead of the transposing launch."""
    import gc
    import torch
    from kvquant_amd import quant_cuda as qc
    calls = []

    def rec(o, i, ot, it, t0, t1):
        assert tuple(ot.shape) == (o.shape[1], o.shape[0]) and ot.dtype == torch.float32 and it.dtype == torch.int32
        calls.append((t0, t1))

    def step(o, i, L, stream=7):
        calls.clear()
        s = qc._shadow_mirror(o, i, L, stream=stream, transpose=rec)
        assert s is not None and s.length == L
        return calls[0]

    qc.shadow_invalidate()
    vals, idx = torch.zeros(64, 6), torch.zeros(64, 6, dtype=torch.int32)
    assert step(vals, idx, 10) == (0, 10)                     # first call: everything
    vals[10] = 1.0
    idx[10] = 3
    assert step(vals, idx, 11) == (10, 11)                    # the glue's append: one row
    vals[11] = 2.0
    idx[11] = 4
    vals[12] = 2.0
    idx[12] = 4
    assert step(vals, idx, 13) == (11, 13)                    # two appends between two calls: two rows
    vals[3] = 9.0                                             # an older row rewritten + an append: the counters moved by 2
    vals[13] = 1.0
    idx[13] = 1
    assert step(vals, idx, 14) == (0, 14)
    assert step(vals, idx, 14) == (0, 14)                     # the same length again
    assert step(vals, idx, 5) == (0, 5)                       # a shorter cache (reset)
    vals[5:9] = 1.0                                           # a prompt: many rows, ONE in-place write per array
    idx[5:9] = 2
    assert step(vals, idx, 9) == (0, 9)
    vals[9] = 1.0                                             # only ONE of the two arrays written
    assert step(vals, idx, 10) == (0, 10)
    vals[10] = 1.0
    idx[10] = 1
    assert step(vals, idx, 11, stream=8) == (0, 11)           # another stream
    vals[11] = 1.0
    idx[11] = 1
    assert step(vals, idx, 12, stream=8) == (11, 12)
    idx2 = idx.clone()                                        # another index tensor
    vals[12] = 1.0
    idx2[12] = 1
    assert step(vals, idx2, 13, stream=8) == (0, 13)
    vals[13] = 1.0
    idx2[13] = 1
    qc.shadow_invalidate(vals)                                # the documented way out for writers torch does not see
    assert step(vals, idx2, 14, stream=8) == (0, 14)
    # the entry dies with the tensor
    n0 = len(qc._shadow)
    del vals
    gc.collect()
    assert len(qc._shadow) == n0 - 1
    qc.shadow_invalidate()

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

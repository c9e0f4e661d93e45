"""kvq_prefill_attention (MFMA flash-style causal attention of the prompt, BASELINE config 4) against a plain
fp32 torch reference of the same op (the reference itself delegates to flash-attn, third party: SURVEY 8c)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v):
    S = q.shape[1]
    s = torch.matmul(q.float(), k.float().transpose(1, 2)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(~torch.ones(S, S, device=q.device, dtype=torch.bool).tril(), float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v.float())           # [H, S, D]


@pytest.mark.parametrize("S,H", [(1, 2), (31, 3), (64, 2), (129, 4), (300, 32), (1000, 5), (2048, 32)])
def test_prefill_attention_matches_fp32_reference(S, H):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(S)
    # token-major activations viewed head-major, as the attention module has them (non-contiguous)
    q, k, v = ((torch.randn(S, H, 128, device=dev, generator=g) * sc).half().transpose(0, 1) for sc in (1.5, 1.5, 1.0))
    out = ops.prefill_attention(q, k, v).view(S, H, 128).transpose(0, 1).float()
    ref = _ref(q, k, v)
    err = float((out - ref).abs().max() / ref.abs().max())
    print("S=%d H=%d max rel err %.2e" % (S, H, err))
    assert err < 2e-3, err
    # asymmetric check of the layouts: a one-hot value matrix must come back permuted by the attention weights only
    if S <= 129:
        v2 = torch.zeros_like(v)
        v2[:, :, :S] = torch.eye(S, 128, device=dev)[:, :S].half() if S <= 128 else v2[:, :, :S]
        if S <= 128:
            o2 = ops.prefill_attention(q, k, v2.contiguous()).view(S, H, 128).transpose(0, 1).float()
            assert float((o2 - _ref(q, k, v2)).abs().max()) < 2e-3


def test_prefill_attention_at_8192():
    """BASELINE config 4: the S = 8192 prompt, 32 heads, against the fp32 reference evaluated head by head (an S x S
    fp32 score matrix is 256 MB)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    S, H = 8192, 32
    g = torch.Generator(device=dev).manual_seed(8192)
    q, k, v = ((torch.randn(S, H, 128, device=dev, generator=g) * sc).half().transpose(0, 1) for sc in (1.5, 1.5, 1.0))
    out = ops.prefill_attention(q, k, v).view(S, H, 128).transpose(0, 1).float()
    worst, bias = 0.0, 0.0
    for h in range(H):
        ref = _ref(q[h:h + 1], k[h:h + 1], v[h:h + 1])[0]
        worst = max(worst, float((out[h] - ref).abs().max() / ref.abs().max()))
        bias += float(((out[h] - ref) * ref.sign()).mean() / ref.abs().mean())
    print("S=8192 H=32 max rel err %.2e, mean signed rel err %.2e" % (worst, bias / H))
    assert worst < 2e-3
    assert abs(bias / H) < 1e-4          # no systematic shrink (P and O are rounded to nearest)

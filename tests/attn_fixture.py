"""Deterministic projection weights of the attention fixtures (tests/golden/gen_attention.py writes the fixtures by
running the reference's own LlamaAttention with these weights; tests/test_attention_gpu.py rebuilds them on the GPU
box -- same torch build, same CPU generator -- and checks the checksum stored in the fixture)."""
import torch

C = 4096


def make_weights(seed):
    """fp16 [C, C] weights of q / k / v / o_proj: N(0, 1/C) with a per-output-channel scale on K (so that the
    per-channel thresholds and codebooks differ) -- activations of O(1) for hidden states of O(1)."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name in ("q", "k", "v", "o"):
        m = torch.randn(C, C, generator=g) / C ** 0.5
        if name == "k":
            m = m * torch.exp(0.5 * torch.randn(C, 1, generator=g))
        if name == "v":
            m = m * 1.5
        w[name] = m.half()
    return w


def checksum(w):
    return float(sum(w[k].double().abs().sum().item() * (i + 1) for i, k in enumerate(("q", "k", "v", "o"))))

"""GPU: KVQuantAttention.attend (prefill + GPU-resident decode, with and without fp16
attention-sink tokens) against the reference protocol of modeling_llama.py:1861-2006
restated on the CPU with the oracle classes."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C


def _rot(x):
    return torch.cat((-x[..., HD // 2:], x[..., :HD // 2]), dim=-1)


def _cos_sin(start, end, theta, dtype):
    inv = 1.0 / (theta ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD))
    t = torch.arange(start, end, dtype=torch.int64).float()
    emb = torch.cat((torch.outer(t, inv),) * 2, dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


class RefAttention:
    """the reference's attention protocol on CPU tensors (fp16 storage, oracle kernels)"""

    def __init__(self, bits, sinks, max_len, quant, theta):
        from oracle.glue import OracleQuantK, OracleQuantV
        kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
                  sparsity_threshold=0.99, first_few_fp16=sinks)
        self.k, self.v = OracleQuantK(rope_theta=theta, **kw), OracleQuantV(**kw)
        self.k.load_lookup_table(quant, True, 0.99)
        self.v.load_lookup_table(quant, True, 0.99)
        self.sinks, self.theta = sinks, theta
        self.kf = torch.zeros(1, H, HD, sinks, dtype=torch.float16)
        self.vf = torch.zeros(1, H, sinks, HD, dtype=torch.float16)

    def attend(self, q, k, v):
        from oracle.glue import OracleQuantV
        q_len, sinks = q.shape[2], self.sinks
        cos, sin = _cos_sin(self.k.klen, self.k.klen + q_len, self.theta, q.dtype)
        qr = (q * cos) + (_rot(q) * sin)
        if q_len > 1 and self.k.klen == 0:
            kr = (k * cos) + (_rot(k) * sin)
            attn = F.scaled_dot_product_attention(qr.float(), kr.float(), v.float(), is_causal=True).half()
            out = attn.transpose(1, 2).reshape(1, q_len, C)
            n = min(sinks, q_len)
            if sinks > 0:
                self.kf[:, :, :, :n] = kr[:, :, :n, :].transpose(2, 3)
                self.vf[:, :, :n, :] = v[:, :, :n, :]
            if q_len > sinks:
                self.k.parallel_pack(k[0, :, sinks:, :].transpose(1, 2))
                vt = v[0, :, sinks:, :].transpose(1, 2)
                vflat = vt.reshape(C, -1).t().float().contiguous()
                self.v.parallel_pack(vt, *OracleQuantV.topk_inputs(vflat, 0.99, C))
            self.k.klen += n
            self.v.vlen += n
            return out
        w = self.k.forward_fused_sparse(qr[0], k).unsqueeze(0) / math.sqrt(HD)
        if sinks > 0:
            w = torch.cat((torch.matmul(qr.float(), self.kf.float()).half() / math.sqrt(HD), w), dim=-1)
        w = F.softmax(w, dim=-1, dtype=torch.float32).half()
        uv, ui, lv, li = OracleQuantV.topk_inputs(v.flatten().float().unsqueeze(0), 0.99, C)
        out = self.v.forward_fused_sparse(w[0, :, :, sinks:], v, uv[0], ui[0], lv[0], li[0]).unsqueeze(0)
        if sinks > 0:
            out = out + torch.matmul(w[..., :sinks].float(), self.vf.float()).half()
        return out.transpose(1, 2).reshape(1, q_len, C)


@pytest.mark.parametrize("bits,sinks", [(4, 0), (3, 5), (4, 5)])
def test_attend_prefill_and_decode(bits, sinks):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.attention import KVQuantAttention
    from tests.test_fused_gpu import _quantizer
    dev = torch.device("cuda")
    quant, scale, shift = _quantizer(bits, seed=bits)
    S, steps, theta = 17, 3, 10000.0
    max_len = 64
    g = torch.Generator().manual_seed(7 + bits)
    kall = util.k_tokens(S + steps, scale, shift, seed=31).half()
    vall = util.v_tokens_no_ties(S + steps, seed=41 + sinks).half()
    qall = torch.randn(S + steps, C, generator=g).half()
    att = KVQuantAttention(hidden_size=C, num_heads=H, abits=bits, include_sparse=True, first_few_fp16=sinks,
                           maxseqlen=max_len, rope_theta=theta, device=dev)
    att.load_quantizers(quant, quant)
    ref = RefAttention(bits, sinks, max_len, quant, theta)

    def st(x, a, b):
        return x[a:b].view(1, b - a, H, HD).transpose(1, 2).contiguous()

    o_gpu = att.attend(st(qall, 0, S).to(dev), st(kall, 0, S).to(dev), st(vall, 0, S).to(dev))
    o_ref = ref.attend(st(qall, 0, S), st(kall, 0, S), st(vall, 0, S))
    assert util.rel_err(o_gpu.float().cpu().reshape(S, -1), o_ref.float().reshape(S, -1)) < 5e-3
    for i in range(steps):
        a, b = S + i, S + i + 1
        o_gpu = att.attend(st(qall, a, b).to(dev), st(kall, a, b).to(dev), st(vall, a, b).to(dev))
        o_ref = ref.attend(st(qall, a, b), st(kall, a, b), st(vall, a, b))
        err = util.rel_err(o_gpu.float().cpu().reshape(1, -1), o_ref.float().reshape(1, -1))
        assert err < 5e-3, (i, err)
    L = ref.k.klen - sinks
    assert att.kcache.klen == ref.k.klen and att.vcache.vlen == ref.v.vlen
    assert torch.equal(ref.k.kcache[:, :, :L], att.kcache.kcache[:, :, :L].cpu())
    assert torch.equal(ref.v.vcache[:, :, :L], att.vcache.vcache[:, :, :L].cpu())
    assert torch.equal(ref.k.outlier_indices[:L], att.kcache.outlier_indices[:L].cpu())


def test_forward_runs_with_projections():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.attention import KVQuantAttention
    from tests.test_fused_gpu import _quantizer
    dev = torch.device("cuda")
    quant, _, _ = _quantizer(4, seed=4)
    torch.manual_seed(0)
    att = KVQuantAttention(abits=4, first_few_fp16=5, maxseqlen=64, device=dev)
    att.load_quantizers(quant, quant)
    x = (torch.randn(1, 12, C, device=dev) * 0.5).half()
    y = att(x)
    assert y.shape == (1, 12, C) and torch.isfinite(y).all()
    y1 = att((torch.randn(1, 1, C, device=dev) * 0.5).half())
    assert y1.shape == (1, 1, C) and torch.isfinite(y1).all()
    assert att.kcache.klen == 13


@pytest.mark.parametrize("name", ["ref_attention_nuq4", "ref_attention_nuq3_sink5"])
def test_replay_reference_attention_module(name):
    """KVQuantAttention against fixtures made by EXECUTING the reference's own LlamaAttention.forward (ML:1388-1760;
    tests/golden/gen_attention.py: prefill 40 tokens + 6 decode tokens, first_few_fp16 in {0, 5}).  The post-projection
    states the reference module saw are replayed through attend(): every call's attention output (the input of o_proj)
    within 1e-3 in decode (the prefill branch is an fp16 matmul chain there and the MFMA flash kernel here: 2e-2), and
    the state the calls leave behind -- packed caches, outlier rows, per-token codebooks -- bit for bit."""
    import os
    import numpy as np
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.attention import KVQuantAttention
    from tests import attn_fixture
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    bits, sinks, S, steps, max_len = int(g["bits"]), int(g["sinks"]), int(g["S"]), int(g["steps"]), int(g["max_len"])
    dev = torch.device("cuda")
    w = attn_fixture.make_weights(int(g["seed"]))
    assert abs(attn_fixture.checksum(w) - float(g["weights_checksum"])) < 1e-6 * float(g["weights_checksum"])
    quant = (g["q_upper"], g["q_lower"], [g["q_centroids"]])

    def build():
        att = KVQuantAttention(hidden_size=C, num_heads=H, abits=bits, include_sparse=True, first_few_fp16=sinks,
                               maxseqlen=max_len, rope_theta=float(g["theta"]), device=dev)
        with torch.no_grad():
            for n in ("q", "k", "v", "o"):
                getattr(att, n + "_proj").weight.copy_(w[n].to(dev))
        att.load_quantizers(quant, quant)
        return att

    def st(x, a, b):
        return torch.from_numpy(x[a:b]).view(1, b - a, H, HD).transpose(1, 2).contiguous().to(dev)

    att = build()
    ctx = torch.from_numpy(g["ctx"]).float()
    with torch.no_grad():
        o = att.attend(st(g["q_states"], 0, S), st(g["k_states"], 0, S), st(g["v_states"], 0, S))
        # (the fixture's prefill is the reference's EAGER branch: q.K^T rounded to fp16 before the softmax, ML:1594-1596 --
        #  scores of +-10 carry 1e-2 of rounding there; the flash kernel keeps them in fp32 like flash-attn does in the
        #  deployed LlamaFlashAttention2)
        assert util.rel_err(o.float().cpu().reshape(S, -1), ctx[:S]) < 2e-2
        for i in range(steps):
            t = S + i
            o = att.attend(st(g["q_states"], t, t + 1), st(g["k_states"], t, t + 1), st(g["v_states"], t, t + 1))
            err = util.rel_err(o.float().cpu().reshape(1, -1), ctx[t:t + 1])
            assert err < 1e-3, (i, err)
    L = int(g["L"])
    kq, vq = att.kcache, att.vcache
    assert kq.klen == S + steps and vq.vlen == S + steps
    assert np.array_equal(g["kcache"], kq.kcache[:, :, :L].cpu().numpy())
    assert np.array_equal(g["vcache"], vq.vcache[:, :, :L].cpu().numpy())
    assert np.array_equal(g["k_outlier_indices"], kq.outlier_indices[:L].cpu().numpy())
    assert np.array_equal(g["v_outlier_indices"], vq.outlier_indices[:L].cpu().numpy())
    assert np.array_equal(g["k_outliers"].view(np.int32), kq.outliers[:L].cpu().numpy().view(np.int32))
    assert np.array_equal(g["v_outliers"].view(np.int32), vq.outliers[:L].cpu().numpy().view(np.int32))
    assert np.array_equal(g["v_lookup_table"].view(np.int32), vq.lookup_table[:L].cpu().numpy().view(np.int32))
    if sinks:
        # post-RoPE sink keys: cos / sin come from the device's libm here and the host's there -- one fp16 ulp
        assert torch.allclose(att.kcache_fp16.float().cpu(), torch.from_numpy(g["kcache_fp16"]).float(), rtol=2e-3, atol=2e-3)
        assert torch.equal(att.vcache_fp16.cpu(), torch.from_numpy(g["vcache_fp16"]))
    # the whole module from the hidden states (projections on this GPU: their fp16 rounding differs from the host's)
    att2 = build()
    hid = torch.from_numpy(g["hidden"]).to(dev)
    ref_out = torch.from_numpy(g["out"]).float()
    with torch.no_grad():
        y = att2(hid[:, :S])
        assert util.rel_err(y.float().cpu().reshape(S, -1), ref_out[:S]) < 1e-2
        for i in range(steps):
            t = S + i
            y = att2(hid[:, t:t + 1])
            assert util.rel_err(y.float().cpu().reshape(1, -1), ref_out[t:t + 1]) < 1e-2, i


def test_rope_q_f16_is_bit_identical_to_the_torch_glue():
    """kvq_rope_q_f16 (the decode query's RoPE in one launch) against the reference's expression on fp16 tensors
    (ML:1851-1853: q * cos + rotate_half(q) * sin, three rounded torch kernels); and the rotary tables served from the
    per-token cache are the tables computed afresh"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    from kvquant_amd.attention import RotaryDynamic, rotate_half
    dev = torch.device("cuda:0")
    rot = RotaryDynamic(128, base=10000.0, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.zeros(1, dtype=torch.float16, device=dev)
    for pos in (0, 1, 777, 131071, 1048575):
        q = (torch.randn(32, 128, generator=g, device=dev) * 3).half()
        RotaryDynamic._last.clear()
        cos, sin = rot(x, pos, pos + 1)
        cos2, sin2 = rot(x, pos, pos + 1)                       # (second layer of the same token: served from the cache)
        assert cos2 is cos and sin2 is sin
        want = (q[None, :, None, :] * cos) + (rotate_half(q[None, :, None, :]) * sin)
        got = ops.rope_q_f16(q, cos[0], sin[0])
        assert torch.equal(got, want[0, :, 0, :])
    cos3, _ = rot(x, 5, 9)                                       # another range replaces the cached one
    assert cos3.shape == (4, 128)
    RotaryDynamic._last.clear()
    cos4, _ = rot(x, 5, 9)
    assert torch.equal(cos3, cos4)

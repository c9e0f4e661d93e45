"""Cross-validation of the whole GPU-resident path against an independent formulation: simulated
quantisation in plain torch (the semantics of quant/kvquant/simquant_module_quantizer.py SQ:30-113 with the
deployment path's outlier rule: per-channel NUQ keys / per-token NUQ values, the selected outliers kept
exactly) followed by ordinary attention evaluated in float64, with the reference's dtype order for the
softmax (scores -> fp16 -> /sqrt(d) in fp16 -> softmax in fp32 -> fp16, ML:873-874, 1972-1977).  No oracle code
involved: SURVEY 8(c) cross-checks (2) and (3)."""
import math

import pytest
import torch

from tests import decode_check, util

pytestmark = pytest.mark.gpu
H, HD, C = util.H, util.HD, util.C


def _fake_quant_k(k, lut, lo, hi, thr_k):
    """k [S, C] fp32 -> reconstructed keys [S, C] (float64)"""
    rangeval, zp = (hi - lo) / 2, (hi + lo) / 2
    resc = (k - zp) / rangeval
    code = (lut.unsqueeze(0) - k.unsqueeze(-1)).abs().argmin(dim=-1)              # first minimum, [S, C]
    deq = torch.gather(lut.unsqueeze(0).expand(k.shape[0], -1, -1), 2, code.unsqueeze(-1)).squeeze(-1).double()
    up = torch.topk(resc, thr_k, dim=-1).indices
    dn = torch.topk(resc, thr_k, dim=-1, largest=False).indices
    keep = torch.zeros_like(k, dtype=torch.bool)
    keep.scatter_(1, up, True)
    keep.scatter_(1, dn, True)
    keep &= resc.abs() > 1            # inside [-1, 1]: residual zeroed (ML:745-747), the code value stands
    return torch.where(keep, k.double(), deq)


def _fake_quant_v(v, cent, thr_k, zero_code):
    """v [S, C] fp32 -> reconstructed values [S, C] (float64); per-token codebooks from the 22nd order statistics"""
    uv = torch.topk(v, thr_k + 1, dim=-1)
    dv = torch.topk(v, thr_k + 1, dim=-1, largest=False)
    vmax, vmin = uv.values[:, -1], dv.values[:, -1]
    rows = cent.unsqueeze(0) * ((vmax - vmin) / 2).unsqueeze(1) + ((vmax + vmin) / 2).unsqueeze(1)    # [S, n]
    code = (rows.unsqueeze(1) - v.unsqueeze(-1)).abs().argmin(dim=-1)
    deq = torch.gather(rows, 1, code).double()
    keep = torch.zeros_like(v, dtype=torch.bool)
    keep.scatter_(1, uv.indices[:, :-1], True)
    keep.scatter_(1, dv.indices[:, :-1], True)
    return torch.where(keep, v.double(), deq)


@pytest.mark.parametrize("bits", [4, 3])
def test_gpu_path_against_simulated_quantisation(bits):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import QuantK, QuantV, decode_kv
    dev = torch.device("cuda:0")
    quant, scale, shift = decode_check.quantizer(bits, seed=21 + bits)
    prefill, steps, max_len = 40, 8, 64
    n = prefill + steps
    ks = util.k_tokens(n, scale, shift, seed=80 + bits).half().float()
    vs = util.v_tokens_no_ties(n, seed=90 + bits).half().float()
    g = torch.Generator().manual_seed(100 + bits)
    qs = (torch.randn(steps, H, HD, generator=g) * 0.5).half()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0, device=dev)
    gk, gv = QuantK(rope_theta=10000.0, **kw), QuantV(**kw)
    for c in (gk, gv):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    gk.parallel_pack(ks[:prefill].t().reshape(H, HD, prefill).contiguous().to(dev))
    gv.parallel_pack(vs[:prefill].t().reshape(H, HD, prefill).contiguous().to(dev))
    outs = []
    for i in range(steps):
        out, _ = decode_kv(gk, gv, qs[i].to(dev), ks[prefill + i].half().to(dev), vs[prefill + i].half().to(dev))
        outs.append(out[0].double().cpu())
    # ---- the same in plain torch -------------------------------------------------------------------------
    thr_k = gk.num_outliers // 2
    khat = _fake_quant_k(ks, gk.lookup_table.reshape(C, -1).cpu(), gk.outlier_threshold_lower.cpu(),
                         gk.outlier_threshold_upper.cpu(), thr_k)
    vhat = _fake_quant_v(vs, gv.lut.cpu().float(), thr_k, None)
    inv_freq = torch.tensor([float(10000.0 ** (-2.0 * j / HD)) for j in range(HD // 2)], dtype=torch.float32)
    pos = torch.arange(n, dtype=torch.float32)
    ang = (pos.unsqueeze(1) * inv_freq.unsqueeze(0)).double()              # fl32(theta_j * pos), then exact trig
    cos, sin = torch.cat((ang.cos(), ang.cos()), -1), torch.cat((ang.sin(), ang.sin()), -1)       # [n, hd]
    kh = khat.reshape(n, H, HD)
    krot = kh * cos.unsqueeze(1) + torch.cat((-kh[..., HD // 2:], kh[..., :HD // 2]), -1) * sin.unsqueeze(1)
    worst = 0.0
    for i in range(steps):
        L = prefill + i + 1
        s = torch.einsum("hd,lhd->hl", qs[i].double(), krot[:L])                      # [H, L] raw scores
        p = torch.softmax(s.float().half() / math.sqrt(HD), dim=-1, dtype=torch.float32).half().double()
        o = torch.einsum("hl,lhd->hd", p, vhat[:L].reshape(L, H, HD))
        worst = max(worst, util.rel_err(outs[i].reshape(1, -1), o.reshape(1, -1)))
    assert worst < 1e-3, worst

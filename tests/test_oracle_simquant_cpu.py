"""The oracle's kernel arithmetic (scalar C restatement of the reference CUDA kernels) against an
independent derivation in plain torch: unpack the codes, dequantise through the codebooks, add the sparse
residuals, rotate the keys (RoPE on the dequantised pre-RoPE keys) and evaluate q.K^T and p.V in float64.
The reference itself cannot run here, so this is the cross-check SURVEY 8(c) asks for: it cannot pin the
oracle to the reference, but it does rule out an error that the oracle and the HIP kernels would share."""
import math

import torch

from oracle import ckernels as ck
from tests import util

H, HD, C = util.H, util.HD, util.C


def _dequant(bits, mat, L):
    """codes [L, C] (long) from the packed cache"""
    return ck.unpack_codes(bits, mat, C, L).long()


def test_oracle_scores_and_outputs_against_float64_formulation():
    for bits in (4, 3, 2):
        n = 2 ** bits
        L, max_len, pos_offset = 190, 200, 7
        g = torch.Generator().manual_seed(bits)
        W = HD // 32 * bits
        kmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), generator=g, dtype=torch.int64).to(torch.int32)
        vmat = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), generator=g, dtype=torch.int64).to(torch.int32)
        klut = torch.randn(H, HD, n, generator=g).sort(dim=-1).values.contiguous()
        vrows = torch.randn(max_len, n, generator=g).sort(dim=-1).values.contiguous()
        q = torch.randn(1, H, HD, generator=g)
        p = torch.softmax(torch.randn(1, H, L, generator=g), dim=-1).half().float()
        kv = torch.randn(max_len, 42, generator=g)
        vv = torch.randn(max_len, 42, generator=g)
        ki = torch.stack([torch.sort(torch.randperm(C, generator=g)[:42]).values for _ in range(max_len)]).int()
        vi = torch.stack([torch.sort(torch.randperm(C, generator=g)[:42]).values for _ in range(max_len)]).int()
        # ---- oracle
        s_or = torch.zeros(1, H, L)
        ck.score_k(bits, q, kmat, s_or, klut, L, 10000.0, pos_offset)
        ck.spmv_k_rope(kv, ki, q, s_or, L, 10000.0, pos_offset)
        o_or = torch.zeros(1, H, HD)
        ck.mix_v(bits, p, vmat, o_or, vrows, L)
        ck.spmv_v(vv, vi, p, o_or, L)
        # ---- plain torch, float64
        kc = _dequant(bits, kmat, L)                                        # [L, C]
        khat = torch.gather(klut.reshape(C, n).unsqueeze(0).expand(L, -1, -1), 2, kc.unsqueeze(-1)).squeeze(-1).double()
        khat.scatter_add_(1, ki[:L].long(), kv[:L].double())
        vc = _dequant(bits, vmat, L)
        vhat = torch.gather(vrows[:L], 1, vc).double()
        vhat.scatter_add_(1, vi[:L].long(), vv[:L].double())
        inv_freq = torch.tensor([float(10000.0 ** (-2.0 * j / HD)) for j in range(HD // 2)], dtype=torch.float32)
        pos = (torch.arange(L) + pos_offset).float()
        ang = (pos.unsqueeze(1) * inv_freq.unsqueeze(0)).double()           # fl32(theta_j * pos), exact trig after that
        cos, sin = torch.cat((ang.cos(), ang.cos()), -1), torch.cat((ang.sin(), ang.sin()), -1)
        kh = khat.reshape(L, H, HD)
        krot = kh * cos.unsqueeze(1) + torch.cat((-kh[..., HD // 2:], kh[..., :HD // 2]), -1) * sin.unsqueeze(1)
        s_t = torch.einsum("hd,lhd->hl", q[0].double(), krot)
        o_t = torch.einsum("hl,lhd->hd", p[0].double(), vhat.reshape(L, H, HD))
        assert util.rel_err(s_or[0].reshape(1, -1), s_t.reshape(1, -1)) < 1e-5, bits
        assert util.rel_err(o_or[0].reshape(1, -1), o_t.reshape(1, -1)) < 1e-5, bits

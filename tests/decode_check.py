"""decode_kv (the one-call GPU-resident decode step: prologue -> q.K^T + softmax pass 1 -> softmax finish
-> p.V -> slab reduce) against the restated reference pipeline (oracle.glue classes + torch softmax in the
reference's dtype order).  Shared by tests/test_decode_kv_gpu.py and __graft_entry__.smoke()."""
import math

import torch

from tests import util

H, HD, C = util.H, util.HD, util.C


def quantizer(bits, seed=0, C=C):
    g = torch.Generator().manual_seed(seed)
    scale = torch.exp(0.5 * torch.randn(C, generator=g))
    shift = 0.3 * torch.randn(C, generator=g)
    upper = (shift + 2.5 * scale).numpy()[None, :]
    lower = (shift - 2.5 * scale).numpy()[None, :]
    cent = util.centroids(bits)[torch.randperm(2 ** bits, generator=g)].numpy().reshape(-1, 1)
    return (upper, lower, [cent]), scale, shift


def run(device, bits=4, prefill=40, steps=4, max_len=64, tol=1e-3, norm=False, heads=H):
    """prefill `prefill` tokens with parallel_pack, then `steps` decode tokens through decode_kv; returns the
    worst relative error of the attention outputs (asserts the packed state bit for bit).  heads: other model
    widths (hidden = heads * 128; the outlier count follows the reference's int(((1 - t) / 2) * hidden) + 1)."""
    from kvquant_amd.cache import QuantK, QuantV, decode_kv
    from oracle.glue import OracleQuantK, OracleQuantV
    H, C = heads, heads * HD
    quant, scale, shift = quantizer(bits, seed=bits, C=C)
    if norm:      # (upper, lower, [centroids], normscale, normoffset), SQ:550-555
        quant = tuple(quant) + (torch.tensor(1.07), torch.tensor(-0.02))
    n = prefill + steps
    ks = util.k_tokens(n, scale, shift, seed=30 + bits)
    vs = util.v_tokens_no_ties(n, Cn=C, seed=40 + bits, k=int(((1 - 0.99) / 2) * C) + 1)
    g = torch.Generator().manual_seed(50 + bits)
    qs = torch.randn(steps, H, 1, HD, generator=g).half()
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, first_few_fp16=0)
    ok, ov = OracleQuantK(rope_theta=10000.0, **kw), OracleQuantV(**kw)
    gk, gv = QuantK(rope_theta=10000.0, device=device, **kw), QuantV(device=device, **kw)
    for c in (ok, ov, gk, gv):
        c.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99, norm=norm)
    if prefill:
        kp = ks[:prefill].half().float().t().reshape(H, HD, prefill).contiguous()
        vp = vs[:prefill].half().float().t().reshape(H, HD, prefill).contiguous()
        ok.parallel_pack(kp)
        gk.parallel_pack(kp.to(device))
        uv, ui, lv, li = OracleQuantV.topk_inputs(vs[:prefill].half().float(), 0.99, C)
        ov.parallel_pack(vp, uv, ui, lv, li)
        gv.parallel_pack(vp.to(device))
    worst = 0.0
    for i in range(steps):
        k16, v16, q16 = ks[prefill + i].half(), vs[prefill + i].half(), qs[i]
        s_ref = ok.forward_fused_sparse(q16, k16)                                  # half [H, 1, L]
        p = torch.softmax(s_ref / math.sqrt(HD), dim=-1, dtype=torch.float32).half()
        uv, ui, lv, li = OracleQuantV.topk_inputs(v16.float().unsqueeze(0), 0.99, C)
        o_ref = ov.forward_fused_sparse(p, v16, uv[0], ui[0], lv[0], li[0])        # half [H, 1, hd]
        out, _ = decode_kv(gk, gv, q16[:, 0, :].contiguous().to(device), k16.to(device), v16.to(device))
        err = util.rel_err(out.float().cpu().reshape(1, -1), o_ref.float().reshape(1, -1))
        worst = max(worst, err)
        assert err < tol, (i, err)
    L = n
    assert torch.equal(ok.kcache[:, :, :L], gk.kcache[:, :, :L].cpu())
    assert torch.equal(ov.vcache[:, :, :L], gv.vcache[:, :, :L].cpu())
    assert torch.equal(ok.outlier_indices[:L], gk.outlier_indices[:L].cpu())
    assert torch.equal(ok.outliers[:L].view(torch.int32), gk.outliers[:L].cpu().view(torch.int32))
    assert torch.equal(gk.outlier_indices_t[:, :L].t().cpu(), gk.outlier_indices[:L].cpu())
    assert torch.equal(gk.outliers_t[:, :L].t().cpu().view(torch.int32), gk.outliers[:L].cpu().view(torch.int32))
    assert torch.equal(ov.outlier_indices[:L], gv.outlier_indices[:L].cpu())
    assert torch.equal(ov.outliers[:L].view(torch.int32), gv.outliers[:L].cpu().view(torch.int32))
    assert torch.equal(ov.lookup_table[:L].view(torch.int32), gv.lookup_table[:L].cpu().view(torch.int32))
    if norm:
        assert torch.equal(ov.lookup_table2[:L].view(torch.int32), gv.lookup_table2[:L].cpu().view(torch.int32))
    return worst

"""The constant-table ("affine") p.V kernel of the decode path (kvq_mix_va.hip, kvq_mix_v_softmax_affine) against the
per-row kernel it replaces (kvq_mix_v_softmax) on identical inputs -- the two evaluate the same sum, the per-row form
from the stored codebook rows lut*sf_t + off_t (modeling_llama.py:1113), the affine one from (sf_t, off_t) recovered from
those rows -- and against a float64 evaluation of the reference's formula (KCU:3211-3433 + 437-470)."""
import pytest
import torch

from tests import decode_check

pytestmark = pytest.mark.gpu

H, HD, C = decode_check.H, decode_check.HD, decode_check.C


def _cache(bits, L, max_len, dev, seed, compact=False):
    from kvquant_amd.cache import QuantV
    quant, _, _ = decode_check.quantizer(bits, seed=seed)
    g = torch.Generator().manual_seed(seed)
    # heavy-tailed values with a per-token offset, so that off_t matters
    v = torch.randn(C, L, generator=g) * 1.3 + torch.randn(1, L, generator=g) * 0.7
    v[torch.rand(C, L, generator=g) < 0.01] *= 6.0
    kw = dict(bits=bits, hidden_size=C, num_heads=H, max_position_embeddings=max_len, include_sparse=True,
              sparsity_threshold=0.99, device=dev)
    if compact:
        kw["compact"] = True
    vc = QuantV(**kw)
    vc.load_lookup_table(quant, include_sparse=True, sparsity_threshold=0.99)
    vc.parallel_pack(v.half().float().reshape(H, HD, L).contiguous().to(dev))
    return vc


def _parts(bits, scores, inv):
    """what the score kernel leaves behind: per (head, tile) max and sum of exp of the scaled fp16 scores"""
    from kvquant_amd import _lib
    L = scores.shape[-1]
    n_parts = _lib.lib().kvq_score_k_softmax_parts(bits, L, 1)
    T = (L + n_parts - 1) // n_parts
    T = 256 if L >= 16384 else 128
    x = (scores[0].half() * inv).half().float()
    pad = n_parts * T - L
    xp = torch.nn.functional.pad(x, (0, pad), value=float("-inf")).reshape(H, n_parts, T)
    m = xp.max(dim=2).values
    s = torch.exp(xp - m[..., None]).sum(dim=2)
    return torch.stack([m, s], dim=2).contiguous(), n_parts, x


def _reference(vc, bits, x, L, sink=None, v_sink=None):
    """float64: fp16 probabilities over [sink | scores] times the dequantised values (rows + residuals)"""
    allx = x if sink is None else torch.cat([sink.float(), x], dim=1)
    p = torch.softmax(allx.double(), dim=1).half().double()
    ns = 0 if sink is None else sink.shape[1]
    rows = vc.lookup_table[:L].double().cpu()                      # [L, n]
    mat = vc.vcache.cpu()
    from tests import util
    codes = util.unpack_codes(mat, bits, L)                        # [C, L]
    deq = torch.gather(rows.t().contiguous(), 0, codes.long())      # [C, L]: row_t[code]
    if not getattr(vc, "compact", False):
        idx = vc.outlier_indices[:L].long().cpu()
        val = vc.outliers[:L].double().cpu()
    else:
        w = vc.outlier_indices[:L].cpu()
        idx = (w & 0xffff).long()
        val = ((w >> 16) & 0xffff).to(torch.int16).view(torch.float16).double()
    deq.scatter_add_(0, idx.t().contiguous(), val.t().contiguous())
    out = torch.einsum("hcl,hl->hc", deq.reshape(H, HD, L), p[:, ns:].cpu())
    if ns:
        out = out + torch.einsum("hn,hnc->hc", p[:, :ns].cpu(), v_sink.double().cpu()).half().double()
    return out


@pytest.mark.parametrize("bits", [4, 3])
@pytest.mark.parametrize("L,max_len", [(1, 64), (17, 64), (130, 192), (1000, 1024), (4099, 4160), (70001, 70016)])
def test_affine_matches_per_row_kernel(bits, L, max_len):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    vc = _cache(bits, L, max_len, dev, seed=300 + bits + L % 97)
    g = torch.Generator().manual_seed(L)
    scores = (torch.randn(1, H, L, generator=g) * 6.0).to(dev)
    inv = 1.0 / HD ** 0.5
    parts, n_parts, x = _parts(bits, scores, inv)
    a = torch.empty(1, H, HD, device=dev)
    b = torch.empty(1, H, HD, device=dev)
    ops.mix_v_softmax(bits, scores, parts, n_parts, inv, vc.vcache, a, vc.lookup_table, L, vc.outliers, vc.outlier_indices)
    ops.mix_v_softmax(bits, scores, parts, n_parts, inv, vc.vcache, b, vc.lookup_table, L, vc.outliers, vc.outlier_indices,
                      table=vc.lut)
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-5 * scale, ((a - b).abs().max().item(), scale)
    if L <= 4099:
        ref = _reference(vc, bits, x.cpu(), L)
        assert (b[0].double().cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("bits", [4, 3])
def test_affine_with_sinks_and_compact(bits):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    L, max_len, ns = 1500, 1536, 5
    inv = 1.0 / HD ** 0.5
    g = torch.Generator().manual_seed(17 + bits)
    scores = (torch.randn(1, H, L, generator=g) * 5.0).to(dev)
    sink = (torch.randn(H, ns, generator=g) * 0.5).half().to(dev)
    v_sink = torch.randn(H, ns, HD, generator=g).half().to(dev)
    for compact in (False, True):
        vc = _cache(bits, L, max_len, dev, seed=41 + bits, compact=compact)
        parts, n_parts, x = _parts(bits, scores, inv)
        outs, probs = [], []
        for table in (None, vc.lut):
            o = torch.empty(1, H, HD, device=dev)
            sp = ops.mix_v_softmax(bits, scores, parts, n_parts, inv, vc.vcache, o, vc.lookup_table, L,
                                   None if compact else vc.outliers, vc.outlier_indices, sink, v_sink, table=table)
            outs.append(o)
            probs.append(sp)
        scale = outs[0].abs().max().item()
        assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale
        assert torch.equal(probs[0], probs[1])
        ref = _reference(vc, bits, x.cpu(), L, sink.cpu(), v_sink)
        assert (outs[1][0].double().cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()

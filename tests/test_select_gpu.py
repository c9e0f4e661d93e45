"""The outlier selection of the GPU-resident appends (kvq_select.h: pruning + bitwise search, radix select as the fallback)
against a plain torch statement of what it has to return, on tokens built to hit each of its paths:
  * ordinary tokens (a few dozen candidates: the one-value-per-lane search);
  * tokens with hundreds of keys tied at the selection boundary or above it (the candidate list overflows: the radix
    select takes the token), all-equal tokens (padding rows), ties on one side only;
  * tokens whose outliers sit in few lanes / one wave (the per-wave bounds are far apart).
Specification (kvq_fused_append.hip, ML:706-751 / 1086-1176): per side the thr_k largest (smallest) selection keys; among
keys equal to the boundary value the LOWEST channels; the row holds the selected channels in ascending order.  K selects
on the rescaled value, stores 0 for a selected value inside [-1, 1], V stores value - zero point; V's clip thresholds
are the (thr_k + 1)-th values.  Checked for the decode append (1024-lane group) and the prefill pack (256-lane token
groups, two tokens per workgroup -- an overflowing token sends its neighbour through the radix select too)."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

H, HD, C = util.H, util.HD, util.C
K = 21


def tokens(seed):
    g = torch.Generator().manual_seed(seed)
    xs, names = [], []

    def add(name, x):
        xs.append(x.half().float())
        names.append(name)
    base = torch.randn(C, generator=g) * 1.5
    add("gaussian", base)
    for n_tie in (30, 70, 300, 1500):
        x = torch.randn(C, generator=g)
        pos = torch.randperm(C, generator=g)
        x[pos[:n_tie]] = 6.0                       # a run of equal maxima that the cut goes through
        x[pos[n_tie:n_tie + n_tie]] = -5.5
        add("ties x %d at both ends" % n_tie, x)
    x = torch.randn(C, generator=g)
    pos = torch.randperm(C, generator=g)
    x[pos[:10]] = 9.0
    x[pos[10:410]] = 4.0                           # 10 above, then 400 equal at the boundary
    add("10 above + 400 tied at the upper boundary", x)
    add("all equal", torch.full((C,), 0.75))
    add("all zero", torch.zeros(C))
    x = torch.randn(C, generator=g) * 0.01
    x[64:64 + 50] = torch.linspace(3, 5, 50)       # every upper outlier in one wave of the decode append
    x[1000:1000 + 4] = torch.tensor([-7.0, -7.5, -8.0, -8.5])   # the lower ones inside one lane
    add("outliers in one wave / one lane", x)
    x = torch.randn(C, generator=g)
    x[::2] = x[1::2]                               # every value twice
    add("pairs", x)
    return torch.stack(xs), names


def spec_select(sel, k):
    """channels of the k largest keys, ties at the boundary -> lowest channels (a stable descending sort)"""
    order = torch.sort(sel, descending=True, stable=True).indices
    return order[:k]


def spec_rows(sel, k):
    hi = spec_select(sel, k)
    lo = spec_select(-sel, k)
    return hi, lo


def check_v_row(x, got_v, got_i, rows_t, zero_code, name):
    hi, lo = spec_rows(x, K)
    if set(hi.tolist()) & set(lo.tolist()):
        return        # (an all-equal token: a channel selected on both sides is stored once -- only decode == prefill is held)
    want_i = torch.sort(torch.cat((hi, lo))).values
    assert torch.equal(got_i.long().cpu(), want_i), name
    want_v = x[want_i] - rows_t[zero_code]
    assert torch.equal(got_v.cpu().view(torch.int32), want_v.view(torch.int32)), name


@pytest.mark.parametrize("bits", [4, 3])
def test_v_append_and_pack_select_paths(bits):
    from kvquant_amd import ops
    from kvquant_amd.cache import ZERO_CODE
    dev = torch.device("cuda:0")
    xs, names = tokens(7 + bits)
    n = xs.shape[0]
    S = (n + 3) // 4 * 4
    xs = torch.cat((xs, xs[: S - n])) if S > n else xs
    names = names + names[: S - n]
    max_len = 64
    lut_sorted = util.centroids(bits).to(dev)
    W = HD // 32 * bits

    def expect_rows(x):
        uv = torch.sort(x, descending=True).values[K]        # the (K + 1)-th values: clip thresholds
        lv = torch.sort(x).values[K]
        off, sf = (uv + lv) / 2, (uv - lv) / 2
        return util.centroids(bits) * sf + off

    # decode append, token by token
    mat = torch.zeros(H, W, max_len, dtype=torch.int32, device=dev)
    rows = torch.zeros(max_len, 2 ** bits, device=dev)
    ov, oi = torch.zeros(max_len, 2 * K, device=dev), torch.zeros(max_len, 2 * K, dtype=torch.int32, device=dev)
    for t in range(S):
        ops.append_v_fused(bits, mat, rows, lut_sorted, xs[t].to(dev), ov, oi, K, t)
    torch.cuda.synchronize()
    for t in range(S):
        want_rows = expect_rows(xs[t])
        assert torch.equal(rows[t].cpu().view(torch.int32), want_rows.view(torch.int32)), names[t]
        check_v_row(xs[t], ov[t], oi[t], want_rows, ZERO_CODE[bits], "decode append: " + names[t])
    # prefill pack of the same tokens: channel-major prompt [H, hd, S]
    mat2 = torch.zeros_like(mat)
    rows2 = torch.zeros_like(rows)
    ov2, oi2 = torch.zeros_like(ov), torch.zeros_like(oi)
    xp = xs.t().contiguous().view(H, HD, S).to(dev)
    ops.pack_v_fused(bits, mat2, rows2, lut_sorted, xp, ov2, oi2, K, 0)
    torch.cuda.synchronize()
    assert torch.equal(mat2[:, :, :S], mat[:, :, :S])
    assert torch.equal(rows2[:S].view(torch.int32), rows[:S].view(torch.int32))
    assert torch.equal(ov2[:S].view(torch.int32), ov[:S].view(torch.int32))
    assert torch.equal(oi2[:S], oi[:S])


@pytest.mark.parametrize("bits", [4, 2])
def test_k_append_and_pack_select_paths(bits):
    """K selects on the rescaled value (x - zp) / range; thresholds -1 / +1 for every channel make that the value itself,
    so the same tie tokens cut runs of equal keys here too"""
    from kvquant_amd import ops
    dev = torch.device("cuda:0")
    xs, names = tokens(11 + bits)
    n = xs.shape[0]
    S = (n + 3) // 4 * 4
    xs = torch.cat((xs, xs[: S - n])) if S > n else xs
    names = names + names[: S - n]
    max_len = 64
    W = HD // 32 * bits
    g = torch.Generator().manual_seed(3)
    lut = torch.randn(H, HD, 2 ** bits, generator=g).sort(dim=-1).values.contiguous().to(dev)
    lo, hi = -torch.ones(C, device=dev), torch.ones(C, device=dev)
    lut_c = lut.view(C, -1).cpu()

    def run(fn_pack):
        mat = torch.zeros(H, W, max_len, dtype=torch.int32, device=dev)
        ov, oi = torch.zeros(max_len, 2 * K, device=dev), torch.zeros(max_len, 2 * K, dtype=torch.int32, device=dev)
        ovt, oit = torch.zeros(2 * K, max_len, device=dev), torch.zeros(2 * K, max_len, dtype=torch.int32, device=dev)
        fn_pack(mat, ov, oi, ovt, oit)
        torch.cuda.synchronize()
        return mat, ov, oi, ovt, oit

    def decode(mat, ov, oi, ovt, oit):
        for t in range(S):
            ops.append_k_fused(bits, mat, lut, lut, xs[t].to(dev), lo, hi, ov, oi, K, t, ovt, oit)

    def prefill(mat, ov, oi, ovt, oit):
        ops.pack_k_fused(bits, mat, lut, lut, xs.t().contiguous().view(H, HD, S).to(dev), lo, hi, ov, oi, K, 0, ovt, oit)

    d = run(decode)
    for t in range(S):
        x = xs[t]
        hi_i, lo_i = spec_rows(x, K)                 # (x - 0) / 1: the selection key is the value
        if set(hi_i.tolist()) & set(lo_i.tolist()):
            continue
        want_i = torch.sort(torch.cat((hi_i, lo_i))).values
        assert torch.equal(d[2][t].long().cpu(), want_i), names[t]
        is_hi = torch.zeros(C, dtype=torch.bool)
        is_hi[hi_i] = True
        end = torch.where(is_hi[want_i], lut_c[want_i, -1], lut_c[want_i, 0])
        inside = torch.where(is_hi[want_i], x[want_i] <= 1.0, x[want_i] >= -1.0)
        want_v = torch.where(inside, torch.zeros_like(end), x[want_i] - end)
        assert torch.equal(d[1][t].cpu().view(torch.int32), want_v.view(torch.int32)), names[t]
    assert torch.equal(d[3][:, :S].t().contiguous().view(torch.int32), d[1][:S].view(torch.int32))     # the mirror
    assert torch.equal(d[4][:, :S].t().contiguous(), d[2][:S])
    p = run(prefill)
    for a, b in zip(d, p):
        if a.shape[-1] == max_len:
            assert torch.equal(a[..., :S].view(torch.int32), b[..., :S].view(torch.int32))
        else:
            assert torch.equal(a[:S].view(torch.int32), b[:S].view(torch.int32))


@pytest.mark.parametrize("Hs,k", [(3, 2), (8, 6), (5, 21), (40, 21)])
def test_v_select_other_widths(Hs, k):
    """hidden sizes other than 4096: waves of the selection group that hold no keys publish no bound (everything is a
    candidate: short lists are searched as they are, long ones go to the radix select), C > 4096 takes the per-token pack"""
    from kvquant_amd import ops
    from kvquant_amd.cache import ZERO_CODE
    bits = 4
    dev = torch.device("cuda:0")
    Cs = Hs * HD
    g = torch.Generator().manual_seed(Hs * 100 + k)
    S, max_len = 8, 64
    xs = (torch.randn(S, Cs, generator=g) * 2).half().float()
    xs[1, : Cs // 2] = 3.25                     # a long run of ties through the cut
    xs[2] = 0.5                                 # all equal
    lut_sorted = util.centroids(bits).to(dev)
    W = HD // 32 * bits
    mat = torch.zeros(Hs, W, max_len, dtype=torch.int32, device=dev)
    rows = torch.zeros(max_len, 2 ** bits, device=dev)
    ov, oi = torch.zeros(max_len, 2 * k, device=dev), torch.zeros(max_len, 2 * k, dtype=torch.int32, device=dev)
    for t in range(S):
        ops.append_v_fused(bits, mat, rows, lut_sorted, xs[t].to(dev), ov, oi, k, t)
    torch.cuda.synchronize()
    for t in range(S):
        x = xs[t]
        hi, lo = spec_select(x, k), spec_select(-x, k)
        if set(hi.tolist()) & set(lo.tolist()):
            continue
        uv, lv = torch.sort(x, descending=True).values[k], torch.sort(x).values[k]
        want_rows = util.centroids(bits) * ((uv - lv) / 2) + (uv + lv) / 2
        assert torch.equal(rows[t].cpu().view(torch.int32), want_rows.view(torch.int32)), t
        want_i = torch.sort(torch.cat((hi, lo))).values
        assert torch.equal(oi[t].long().cpu(), want_i), t
        assert torch.equal(ov[t].cpu().view(torch.int32), (x[want_i] - want_rows[ZERO_CODE[bits]]).view(torch.int32)), t
    mat2, rows2, ov2, oi2 = torch.zeros_like(mat), torch.zeros_like(rows), torch.zeros_like(ov), torch.zeros_like(oi)
    ops.pack_v_fused(bits, mat2, rows2, lut_sorted, xs.t().contiguous().view(Hs, HD, S).to(dev), ov2, oi2, k, 0)
    torch.cuda.synchronize()
    assert torch.equal(mat2[:, :, :S], mat[:, :, :S])
    assert torch.equal(rows2[:S].view(torch.int32), rows[:S].view(torch.int32))
    assert torch.equal(ov2[:S].view(torch.int32), ov[:S].view(torch.int32)) and torch.equal(oi2[:S], oi[:S])

"""Synthetic inputs shared by the parity tests (seeded)."""
import numpy as np
import torch

H, HD, C = 32, 128, 4096


def centroids(bits, seed=0):
    n = 2 ** bits
    g = torch.Generator().manual_seed(100 + bits + seed)
    c = torch.sort(torch.rand(n, generator=g) * 2 - 1).values
    c[0], c[-1] = -0.98, 0.97
    return c.float()


def k_tables(bits, H=H, hd=HD, seed=0):
    """per-channel LUT [H,hd,n] (rows ascending), thresholds lo/hi [C] (fp16-representable)."""
    g = torch.Generator().manual_seed(seed)
    Cn = H * hd
    scale = torch.exp(0.5 * torch.randn(Cn, generator=g))
    shift = 0.3 * torch.randn(Cn, generator=g)
    hi = (shift + 2.6 * scale).half().float()
    lo = (shift - 2.6 * scale).half().float()
    off = ((hi.half() + lo.half()) / 2).float()
    rng = ((hi.half() - lo.half()) / 2).float()
    lut = centroids(bits).unsqueeze(0) * rng.unsqueeze(1) + off.unsqueeze(1)
    return lut.reshape(H, hd, -1).contiguous(), lo.contiguous(), hi.contiguous(), scale, shift


def k_tokens(S, scale, shift, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, scale.numel(), generator=g) * scale + shift
    m = torch.rand(S, scale.numel(), generator=g) < 0.01
    x[m] *= 4.0
    return x.half().float()


def v_tokens(S, Cn=C, seed=2):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, Cn, generator=g)
    m = torch.rand(S, Cn, generator=g) < 0.01
    x[m] *= 5.0
    return x.half().float()


def v_tokens_no_ties(S, Cn=C, seed=2, k=21):
    """like v_tokens but without ties at the k/(k+1) selection boundary of any token: torch.topk
    leaves the choice among equal values unspecified (the GPU selection takes the lowest channel).
    Tokens with such a tie are redrawn one by one (a whole-batch rejection never ends for large S x Cn)."""
    x = v_tokens(S, Cn, seed)
    for _ in range(1000):
        hi = torch.topk(x, k + 2, dim=-1).values
        lo = torch.topk(x, k + 2, dim=-1, largest=False).values
        bad = (hi[:, k - 1] == hi[:, k]) | (lo[:, k - 1] == lo[:, k])
        if not bool(bad.any()):
            return x
        seed += 1000
        x[bad] = v_tokens(S, Cn, seed)[bad]
    raise RuntimeError("v_tokens_no_ties: could not remove the ties")


def rel_err(a, b, dim=-1):
    """max |a-b| relative to the max magnitude of the reference row"""
    a = a.double()
    b = b.double()
    scale = b.abs().amax(dim=dim, keepdim=True).clamp_min(1e-12)
    return float(((a - b).abs() / scale).max())


def sync_oracle_freqs(theta=10000.0):
    """Give the CPU oracle the RoPE frequency table of THIS GPU (theta_j = powf on the device, as the
    reference evaluates it): call at the top of GPU tests that compare scores at long positions."""
    import torch as _t
    from kvquant_amd import ops
    from oracle import ckernels
    ckernels.set_rope_freqs(theta, ops.rope_freqs(theta).cpu())


def ref_pack_k_agrees_where_defined(ref, name, lut, k, lo, hi, ours_mat, ours_resc, runs=3):
    """The reference's parallel K pack kernels fill a per-block LDS table (codebook rows, range, zero point of
    the block's 128 channels, one channel per thread) and read it back WITHOUT a barrier (KCU:1857-1877,
    2303-2322, 2793-2812).  A thread's reads of the entries its own 64-lane wave wrote are
    ordered; reads of the other wave's entries are a race that, at prompt sizes, one wave loses on every run
    (measured on MI355X: exactly the 64 columns of one wave per block come out wrong).  The reference's result is
    therefore DEFINED only for (token column, channel) pairs whose table entry was written by the reading thread's
    own wave: column % 128 < 64 with channel % 128 < 64, or both >= 64.  There it must agree with ours bit for bit
    (packed words and rescaled values); elsewhere differences are counted and returned, not judged."""
    import torch
    H, W, _ = ours_mat.shape
    hd, S = k.shape[1], k.shape[2]
    dev = k.device
    col_half = (torch.arange(S, device=dev) % 128) >= 64                     # wave of the token's thread
    ch_half = (torch.arange(hd, device=dev) % 128) >= 64                     # wave that wrote the channel's entry
    row_half = (torch.arange(W, device=dev) * hd // W % 128) >= 64           # packed word-row -> its channels' half
    racy_words, racy_vals = [], []
    for _ in range(runs):
        mr = torch.zeros_like(ours_mat)
        rr = torch.zeros_like(ours_resc)
        getattr(ref, name)(mr, lut, k, rr, lo, hi)
        torch.cuda.synchronize()
        bad_w = mr[:, :, :S] != ours_mat[:, :, :S]                           # [H, W, S]
        bad_v = rr.view(torch.int32) != ours_resc.view(torch.int32)          # [H, hd, S]
        defined_w = row_half[None, :, None] == col_half[None, None, :]
        defined_v = ch_half[None, :, None] == col_half[None, None, :]
        assert not bool((bad_w & defined_w).any()), "packed codes differ where the reference result is defined"
        assert not bool((bad_v & defined_v).any()), "rescaled values differ where the reference result is defined"
        assert torch.equal(mr[:, :, S:], ours_mat[:, :, S:])
        racy_words.append(int(bad_w.sum()))
        racy_vals.append(int(bad_v.sum()))
    return racy_words, racy_vals


def unpack_codes(mat, bits, L):
    """codes int64 [C, L] of the first L columns of a packed cache int32 [H, hd/32*bits, max_len] (CPU tensor);
    bit layouts of KCU:1240-1244 (4 bit), 1395-1424 (3 bit), 2712-2716 (2 bit)"""
    Hh, W, _ = mat.shape
    w = (mat[:, :, :L].to(torch.int64) & 0xffffffff).reshape(Hh, W // bits, bits, L)      # 32-channel groups
    out = torch.empty((Hh, W // bits, 32, L), dtype=torch.int64)
    for i in range(32):
        if bits == 4:
            out[:, :, i] = (w[:, :, i // 8] >> (4 * (i % 8))) & 15
        elif bits == 2:
            out[:, :, i] = (w[:, :, i // 16] >> (2 * (i % 16))) & 3
        else:
            b = 3 * i
            lo, sh = b // 32, b % 32
            val = w[:, :, lo] >> sh
            if sh > 29:
                val = val | (w[:, :, lo + 1] << (32 - sh))
            out[:, :, i] = val & 7
    return out.reshape(-1, L)

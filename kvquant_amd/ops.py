"""Typed wrappers over the C ABI of libkvq.so (include/kvq.h): torch tensors in,
raw device pointers + torch's current stream out.  Argument checks raise
ValueError; a failing library call raises KvqError.  No CPU path."""
import ctypes

import torch

from . import _lib

_ws = {}  # (slot, device index, stream) -> workspace tensor


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if not torch.is_tensor(t):
        raise ValueError("%s: expected a tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s: expected a GPU tensor (got %s); kvquant_amd has no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s: expected dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: expected a contiguous tensor" % name)
    return t.data_ptr()


def _f(t, name):
    return _chk(t, torch.float32, name)


def _i(t, name):
    return _chk(t, torch.int32, name)


def _fo(t, name):
    """optional f32 tensor -> pointer / None (compact formats carry no value array: include/kvq.h)"""
    return None if t is None else _chk(t, torch.float32, name)


def _io(t, name):
    return None if t is None else _chk(t, torch.int32, name)


def _cache_dims(mat, bits):
    if mat.dim() != 3:
        raise ValueError("mat: expected [num_heads, head_dim/32*bits, max_len]")
    H, W, max_len = mat.shape
    if W % bits:
        raise ValueError("mat.shape[1]=%d is not a multiple of bits=%d" % (W, bits))
    return H, W // bits * 32, max_len


def _workspace(device, nbytes, slot="mix"):
    """Scratch buffers of the library calls (score tables + fp32 query, softmax partials, p.V slabs), one per
    (slot, device, STREAM): calls on one stream are ordered, so consecutive layers can share them; work on another
    stream of the same device (the reference runs a side stream, ML:1804-1820) gets its own and cannot clobber
    tables or partials that are still being read.  A buffer that has to grow is replaced on its own stream,
    where the caching allocator orders the reuse of the old one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (slot, idx, torch.cuda.current_stream(idx).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        # grow geometrically: the decode workspaces grow by a few hundred bytes per token, and a buffer sized exactly
        # would be replaced every other token once it is past the 1 MiB floor (allocator churn in the hot path)
        grown = 0 if w is None else w.numel() + (w.numel() >> 1)
        w = torch.empty(max(nbytes, grown, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


class _Dev:
    """device guard equivalent to the reference's OptionalCUDAGuard"""

    def __init__(self, t):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        if self.idx is not None and self.idx != torch.cuda.current_device():
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(self.idx)

    def __exit__(self, *a):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)




def _L():
    return _lib.lib()


def rope_freqs(theta, device=None):
    """theta_j = powf(theta, -2j/128), j < 64, as the score kernels evaluate them (on the device)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty(64, dtype=torch.float32, device=dev)
    with _Dev(out):
        _lib.check(_L().kvq_rope_freqs(float(theta), out.data_ptr(), _stream()), "kvq_rope_freqs")
    return out


def rope_q_f16(q, cos, sin):
    """q f16 [H, 128] (contiguous), cos / sin f16 [128] -> q * cos + rotate_half(q) * sin, f16 [H, 128], with torch's fp16
    roundings (kvq_rope_q_f16): the decode query's RoPE of the patched attention in one launch"""
    out = torch.empty_like(q)
    with _Dev(q):
        _lib.check(_L().kvq_rope_q_f16(_chk(q, torch.float16, "q"), _chk(cos, torch.float16, "cos"),
                                       _chk(sin, torch.float16, "sin"), out.data_ptr(), q.shape[0], q.shape[1], _stream()),
                   "kvq_rope_q_f16")
    return out


def append_k(bits, mat, lut, x, col):
    H, hd, max_len = _cache_dims(mat, bits)
    with _Dev(mat):
        _lib.check(_L().kvq_append_k(bits, _i(mat, "mat"), _f(lut, "lookup_table"), _f(x, "newvec"), H, hd,
                                     max_len, int(col), _stream()), "kvq_append_k")


def append_v(bits, mat, lut_rows, x, col):
    H, hd, max_len = _cache_dims(mat, bits)
    with _Dev(mat):
        _lib.check(_L().kvq_append_v(bits, _i(mat, "mat"), _f(lut_rows, "lookup_table"), _f(x, "newvec"), H, hd,
                                     max_len, int(col), _stream()), "kvq_append_v")


def append_k_sparse(bits, mat, lut, x, rescaled, lo, hi, col):
    H, hd, max_len = _cache_dims(mat, bits)
    with _Dev(mat):
        _lib.check(_L().kvq_append_k_sparse(bits, _i(mat, "mat"), _f(lut, "lookup_table"), _f(x, "newvec"),
                                            _f(rescaled, "outliers_rescaled"), _f(lo, "lower"), _f(hi, "upper"),
                                            H, hd, max_len, int(col), _stream()), "kvq_append_k_sparse")


def append_v_sparse(bits, mat, lut_rows, x, lo, hi, col):
    H, hd, max_len = _cache_dims(mat, bits)
    with _Dev(mat):
        _lib.check(_L().kvq_append_v_sparse(bits, _i(mat, "mat"), _f(lut_rows, "lookup_table"), _f(x, "newvec"),
                                            float(lo), float(hi), H, hd, max_len, int(col), _stream()),
                   "kvq_append_v_sparse")


def pack_k_sparse_parallel(bits, mat, lut, x, rescaled, lo, hi, col0=0):
    H, hd, max_len = _cache_dims(mat, bits)
    S = x.shape[-1]
    with _Dev(mat):
        _lib.check(_L().kvq_pack_k_sparse_parallel(bits, _i(mat, "mat"), _f(lut, "lookup_table"), _f(x, "newvec"),
                                                   _f(rescaled, "outliers_rescaled"), _f(lo, "lower"),
                                                   _f(hi, "upper"), H, hd, S, max_len, int(col0), _stream()),
                   "kvq_pack_k_sparse_parallel")


def pack_v_sparse_parallel(bits, mat, lut_rows, x, lo, hi, col0=0):
    H, hd, max_len = _cache_dims(mat, bits)
    S = x.shape[-1]
    with _Dev(mat):
        _lib.check(_L().kvq_pack_v_sparse_parallel(bits, _i(mat, "mat"), _f(lut_rows, "lookup_table"),
                                                   _f(x, "newvec"), _f(lo, "lower"), _f(hi, "upper"), H, hd, S,
                                                   max_len, int(col0), _stream()), "kvq_pack_v_sparse_parallel")


def score_k(bits, q, mat, mul, lut, L, theta, pos_offset, outliers=None, outlier_indices=None,
            accumulate=True):
    """q [q_len,H,128] f32, mul [q_len,H,L] f32 (in place)."""
    H, hd, max_len = _cache_dims(mat, bits)
    if q.dim() != 3 or mul.dim() != 3 or q.shape[0] != mul.shape[0]:
        raise ValueError("vec must be [q_len, H, head_dim] and mul [q_len, H, kcachelen]")
    if mul.shape[2] != L:
        raise ValueError("mul.shape[2] must equal kcachelen")
    n_out = 0 if outliers is None else outliers.shape[1]
    with _Dev(q):
        nbytes = _L().kvq_score_k_workspace_bytes(bits, q.shape[0], H)
        ws = _workspace(q.device, nbytes, slot="score")
        _lib.check(_L().kvq_score_k(
            bits, _f(q, "vec"), _i(mat, "mat"), _f(mul, "mul"), _f(lut, "lookup_table"), q.shape[0], H, hd,
            int(L), max_len, float(theta), int(pos_offset),
            None if outliers is None else _f(outliers, "outliers"),
            None if outliers is None else _i(outlier_indices, "outlier_indices"), n_out,
            1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()), "kvq_score_k")


def outlier_mirror_rows(outliers, outlier_indices, outliers_t, outlier_indices_t, t0, t1):
    """rows [t0, t1) of the reference's outlier rows [max_len, n_out] -> columns [t0, t1) of the token-contiguous
    mirror [n_out, max_len] (include/kvq.h: kvq_outlier_mirror_rows)."""
    max_len, n_out = outliers.shape
    if tuple(outlier_indices.shape) != (max_len, n_out) or tuple(outliers_t.shape) != (n_out, max_len) or \
            tuple(outlier_indices_t.shape) != (n_out, max_len):
        raise ValueError("outlier rows must be [max_len, n_out] and the mirror [n_out, max_len]")
    with _Dev(outliers):
        _lib.check(_L().kvq_outlier_mirror_rows(
            _f(outliers, "outliers"), _i(outlier_indices, "outlier_indices"), _f(outliers_t, "outliers_t"),
            _i(outlier_indices_t, "outlier_indices_t"), n_out, max_len, int(t0), int(t1), _stream()),
            "kvq_outlier_mirror_rows")


def score_k_mirror(bits, q, mat, mul, lut, L, theta, pos_offset, outliers_t, outlier_indices_t, accumulate=True):
    """score_k (q_len = 1) over the token-contiguous outlier mirror [n_out, max_len]: the decode kernel's variant behind
    the legacy call's semantics (include/kvq.h: kvq_score_k_mirror)."""
    H, hd, max_len = _cache_dims(mat, bits)
    if q.dim() != 3 or mul.dim() != 3 or q.shape[0] != 1 or mul.shape[0] != 1:
        raise ValueError("vec must be [1, H, head_dim] and mul [1, H, kcachelen]")
    if mul.shape[2] != L:
        raise ValueError("mul.shape[2] must equal kcachelen")
    n_out = outliers_t.shape[0]
    if tuple(outliers_t.shape) != (n_out, max_len) or tuple(outlier_indices_t.shape) != (n_out, max_len):
        raise ValueError("the outlier mirror must be [n_out, max_len]")
    with _Dev(q):
        nbytes = _L().kvq_score_k_workspace_bytes(bits, 1, H)
        ws = _workspace(q.device, nbytes, slot="score")
        _lib.check(_L().kvq_score_k_mirror(
            bits, _f(q, "vec"), _i(mat, "mat"), _f(mul, "mul"), _f(lut, "lookup_table"), H, hd, int(L), max_len,
            float(theta), int(pos_offset), _f(outliers_t, "outliers_t"), _i(outlier_indices_t, "outlier_indices_t"),
            n_out, 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()), "kvq_score_k_mirror")


def mix_v(bits, p, mat, mul, lut_rows, L, outliers=None, outlier_indices=None, accumulate=True):
    """p [q_len,H,L] f32, mul [q_len,H,128] f32 (in place)."""
    H, hd, max_len = _cache_dims(mat, bits)
    if p.dim() != 3 or mul.dim() != 3 or p.shape[0] != mul.shape[0]:
        raise ValueError("vec must be [q_len, H, vcachelen] and mul [q_len, H, head_dim]")
    if p.shape[2] != L:
        raise ValueError("vec.shape[2] must equal vcachelen")
    q_len = p.shape[0]
    # (outlier_indices without outliers: COMPACT rows, packed entries in the index array)
    n_out = 0 if outlier_indices is None else outlier_indices.shape[1]
    with _Dev(p):
        nbytes = _L().kvq_mix_v_workspace_bytes(bits, q_len, H, hd, int(L))
        ws = _workspace(p.device, nbytes)
        _lib.check(_L().kvq_mix_v(
            bits, _f(p, "vec"), _i(mat, "mat"), _f(mul, "mul"), _f(lut_rows, "lookup_table"), q_len, H, hd,
            int(L), max_len, _fo(outliers, "outliers"), _io(outlier_indices, "outlier_indices"), n_out,
            1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()), "kvq_mix_v")


def _mirror(outliers_t, outlier_indices_t, thr_k, max_len):
    """(ptr, ptr) of the optional token-contiguous K outlier mirror [2*thr_k, max_len], or (None, None)"""
    if outlier_indices_t is None:
        if outliers_t is not None:
            raise ValueError("the outlier mirror needs its index array")
        return None, None
    if tuple(outlier_indices_t.shape) != (2 * thr_k, max_len) or \
            (outliers_t is not None and tuple(outliers_t.shape) != (2 * thr_k, max_len)):
        raise ValueError("the outlier mirror must be [%d, %d]" % (2 * thr_k, max_len))
    # (index array alone: the COMPACT mirror, packed entries fp16 residual << 16 | channel)
    return _fo(outliers_t, "outliers_t"), _i(outlier_indices_t, "outlier_indices_t")


def append_k_fused(bits, mat, lut, lut_off, x, lo, hi, outliers, outlier_indices, thr_k, col, outliers_t=None,
                   outlier_indices_t=None):
    """pack + rescale + exact top-thr_k selection + outlier row, one launch (include/kvq.h)."""
    H, hd, max_len = _cache_dims(mat, bits)
    if outliers.shape[1] != 2 * thr_k or outlier_indices.shape[1] != 2 * thr_k:
        raise ValueError("outlier buffers must be %d wide" % (2 * thr_k))
    with _Dev(mat):
        _lib.check(_L().kvq_append_k_fused(bits, _i(mat, "mat"), _f(lut, "lookup_table"), _f(lut_off, "lut_off"),
                                           _f(x, "newvec"), _f(lo, "lower"), _f(hi, "upper"),
                                           _f(outliers, "outliers"), _i(outlier_indices, "outlier_indices"),
                                           int(thr_k), H, hd, max_len, int(col),
                                           *_mirror(outliers_t, outlier_indices_t, thr_k, max_len), _stream()),
                   "kvq_append_k_fused")


def _vnorm(norm):
    """norm: None or (lut_rows2 f32 [max_len, 2^bits] or None, normscale, normoffset, zp_from_rows2,
    reference_tie_quirk) -> kvq_vopts* / None"""
    if norm is None:
        return None
    rows2, ns, no, zp2, quirk = norm
    return ctypes.byref(_lib.VNorm(None if rows2 is None else _f(rows2, "lookup_table2"), float(ns), float(no),
                                   1 if zp2 else 0, 1 if quirk else 0))


def append_v_fused(bits, mat, lut_rows, lut_sorted, x, outliers, outlier_indices, thr_k, col, norm=None):
    """top-(thr_k+1) thresholds + codebook row + pack + outlier row, one launch."""
    H, hd, max_len = _cache_dims(mat, bits)
    if outliers.shape[1] != 2 * thr_k or outlier_indices.shape[1] != 2 * thr_k:
        raise ValueError("outlier buffers must be %d wide" % (2 * thr_k))
    if lut_sorted.numel() != 2 ** bits:
        raise ValueError("lut_sorted must have 2^bits entries")
    with _Dev(mat):
        _lib.check(_L().kvq_append_v_fused(bits, _i(mat, "mat"), _f(lut_rows, "lookup_table"),
                                           _f(lut_sorted, "lut"), _f(x, "newvec"), _f(outliers, "outliers"),
                                           _i(outlier_indices, "outlier_indices"), int(thr_k), H, hd, max_len,
                                           int(col), _vnorm(norm), _stream()), "kvq_append_v_fused")


def pack_k_fused(bits, mat, lut, lut_off, x, lo, hi, outliers, outlier_indices, thr_k, col0, outliers_t=None,
                 outlier_indices_t=None):
    """prefill: x f32 [H, hd, S] (channel-major prompt) -> columns col0.. of the cache + outlier rows, one launch."""
    H, hd, max_len = _cache_dims(mat, bits)
    S = x.shape[-1]
    if x.numel() != H * hd * S:
        raise ValueError("x must be [H, hd, S]")
    with _Dev(mat):
        _lib.check(_L().kvq_pack_k_fused(bits, _i(mat, "mat"), _f(lut, "lookup_table"), _f(lut_off, "lut_off"),
                                         _f(x, "newvec"), _f(lo, "lower"), _f(hi, "upper"),
                                         _fo(outliers, "outliers"), _io(outlier_indices, "outlier_indices"),
                                         int(thr_k), H, hd, max_len, int(col0), int(S),
                                         *_mirror(outliers_t, outlier_indices_t, thr_k, max_len), _stream()),
                   "kvq_pack_k_fused")


def pack_v_fused(bits, mat, lut_rows, lut_sorted, x, outliers, outlier_indices, thr_k, col0, norm=None):
    """prefill: x f32 [H, hd, S] -> cache columns, per-token codebook rows and outlier rows, one launch."""
    H, hd, max_len = _cache_dims(mat, bits)
    S = x.shape[-1]
    if x.numel() != H * hd * S:
        raise ValueError("x must be [H, hd, S]")
    with _Dev(mat):
        _lib.check(_L().kvq_pack_v_fused(bits, _i(mat, "mat"), _f(lut_rows, "lookup_table"), _f(lut_sorted, "lut"),
                                         _f(x, "newvec"), _fo(outliers, "outliers"),
                                         _i(outlier_indices, "outlier_indices"), int(thr_k), H, hd, max_len,
                                         int(col0), int(S), _vnorm(norm), _stream()), "kvq_pack_v_fused")


def softmax_scale(scores, inv_sqrt_hd, sink_scores=None):
    """scores: f32 [H, L] raw q.K^T; sink_scores: f16 [H, n_sink] already scaled, or None.
    Returns (probs f32 [H, L] holding fp16-rounded values, sink_probs f16 [H, n_sink] or None)."""
    if scores.dim() != 2:
        raise ValueError("scores must be [H, L]")
    H, L = scores.shape
    n_sink = 0 if sink_scores is None else sink_scores.shape[1]
    probs = torch.empty_like(scores)
    sink_probs = None if sink_scores is None else torch.empty_like(sink_scores)
    with _Dev(scores):
        nbytes = _L().kvq_softmax_workspace_bytes(H, L)
        ws = _workspace(scores.device, nbytes + (1 << 16), slot="softmax")
        _lib.check(_L().kvq_softmax_scale(
            _f(scores, "scores"), None if sink_scores is None else _chk(sink_scores, torch.float16, "sink_scores"),
            _f(probs, "probs"), None if sink_probs is None else sink_probs.data_ptr(), H, L, n_sink,
            float(inv_sqrt_hd), ws.data_ptr(), ws.numel(), _stream()), "kvq_softmax_scale")
    return probs, sink_probs


# ---- uncapped ("orig") CSR / CSC variants, 4 bit --------------------------------------------------
def _grow(ptr, minor, val, start, idx, v, pos):
    """the reference's host-side CSR/CSC growth (KCU:763-829, 1010-1060) on device tensors"""
    dev = idx.device
    cnt = idx.numel()
    if ptr.numel() == 0:
        ptr2 = torch.tensor([0, cnt], dtype=torch.int32, device=dev)
        minor2, val2 = idx, v
        nt = (cnt + 9) // 10
        start2 = torch.full((nt,), pos, dtype=torch.int32, device=dev)
    else:
        ptr2 = torch.cat((ptr.int(), torch.tensor([minor.numel() + cnt], dtype=torch.int32, device=dev)))
        if cnt > 0:
            minor2 = torch.cat((minor.int(), idx))
            val2 = torch.cat((val.float(), v))
            nt = (minor2.numel() + 9) // 10
            new_alloc = nt - start.numel()
            start2 = torch.cat((start.int(), torch.full((new_alloc,), pos, dtype=torch.int32, device=dev))) \
                if new_alloc > 0 else start
        else:
            minor2, val2, start2 = minor, val, start
            nt = (minor2.numel() + 9) // 10
    return ptr2, minor2, val2, start2, nt


def _append_orig(is_v, mat, lut, x, zeropoint, lo, hi, col):
    H, hd, max_len = _cache_dims(mat, 4)
    C = H * hd
    oi = torch.empty(C, dtype=torch.int32, device=mat.device)
    ov = torch.empty(C, dtype=torch.float32, device=mat.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=mat.device)
    with _Dev(mat):
        if is_v:
            _lib.check(_L().kvq_append_v_sparse_orig(_i(mat, "mat"), _f(lut, "lookup_table"), _f(x, "newvec"),
                                                     float(zeropoint), float(lo), float(hi), oi.data_ptr(),
                                                     ov.data_ptr(), cnt.data_ptr(), H, hd, max_len, int(col),
                                                     _stream()), "kvq_append_v_sparse_orig")
        else:
            _lib.check(_L().kvq_append_k_sparse_orig(_i(mat, "mat"), _f(lut, "lookup_table"), _f(x, "newvec"),
                                                     _f(zeropoint, "zeropoint"), _f(lo, "lower"), _f(hi, "upper"),
                                                     oi.data_ptr(), ov.data_ptr(), cnt.data_ptr(), H, hd, max_len,
                                                     int(col), _stream()), "kvq_append_k_sparse_orig")
    n = int(cnt.item())          # the reference synchronises here too (KCU:745-747)
    return oi[:n].clone(), ov[:n].clone(), cnt


def append_k_sparse_orig(mat, lut, x, zeropoint, row, col, val, start_rows, lo, hi, kcachelen):
    idx, v, cnt = _append_orig(False, mat, lut, x, zeropoint, lo, hi, kcachelen)
    rows, cols, vals, start, nt = _grow(row, col, val, start_rows, idx, v, kcachelen)
    return [rows, cols, vals, start, torch.tensor([nt], dtype=torch.int32), cnt]


def append_v_sparse_orig(mat, lut_rows, x, zeropoint, row, col, val, start_cols, lo, hi, vcachelen):
    idx, v, cnt = _append_orig(True, mat, lut_rows, x, zeropoint, lo, hi, vcachelen)
    cols, rows, vals, start, nt = _grow(col, row, val, start_cols, idx, v, vcachelen)
    return [rows, cols, vals, start, torch.tensor([nt], dtype=torch.int32), cnt]


def spmv_k_rope_csr(rowptr, cols, vals, q, mul, num_rows, L, theta, pos_offset):
    with _Dev(q):
        _lib.check(_L().kvq_spmv_k_rope_csr(_i(rowptr, "rows"), _i(cols, "cols") if cols.numel() else None,
                                            _f(vals, "vals") if vals.numel() else None, _f(q, "vec"), _f(mul, "mul"),
                                            int(num_rows), int(L), q.shape[2], float(theta), int(pos_offset),
                                            _stream()), "kvq_spmv_k_rope_csr")


def spmv_v_csc(colptr, rows, vals, p, mul, num_cols, L):
    H, hd = mul.shape[1], mul.shape[2]
    with _Dev(p):
        _lib.check(_L().kvq_spmv_v_csc(_i(colptr, "cols"), _i(rows, "rows") if rows.numel() else None,
                                       _f(vals, "vals") if vals.numel() else None, _f(p, "vec"), _f(mul, "mul"),
                                       int(num_cols), int(L), H, hd, _stream()), "kvq_spmv_v_csc")


# ---- one-launch decode prologue + prepared score ----------------------------------------------------
def _act(t, name):
    """activation vector: fp32 or fp16, contiguous, on the GPU -> (ptr, is_half)"""
    if t.dtype == torch.float16:
        return _chk(t, torch.float16, name), 1
    return _chk(t, torch.float32, name), 0


def decode_prologue(bits, kmat, klut, klut_off, k, lo, hi, koutl, kidx, kcol, vmat, vlut_rows, vlut_sorted, v,
                    voutl, vidx, vcol, q, thr_k, koutl_t=None, kidx_t=None, klut_ends=None, klut_score=None,
                    vnorm=None, sinks=None):
    """K fused append + V fused append + K codebook images for score_k_prepared in ONE launch.
    q [H,128] (RoPE'd), k, v [C]: all fp32 or all fp16.  klut_score: table the score images are built from
    (default klut).  sinks = (k_sink f16 [H, 128, n_sink], sink_scores f16 [H, n_sink] (out), inv_sqrt_hd): the head
    workgroups also write the scaled scores of the fp16 sink tokens.  Returns the score workspace tensor."""
    H, hd, max_len = _cache_dims(kmat, bits)
    sk = None
    if sinks is not None:
        k_sink, sink_scores, inv = sinks
        n_sink = sink_scores.shape[1]
        if tuple(k_sink.shape) != (H, 128, n_sink) or sink_scores.shape[0] != H:
            raise ValueError("sinks: k_sink [H, 128, n_sink] and sink_scores [H, n_sink] expected")
        sk = _lib.Sinks(_chk(k_sink, torch.float16, "k_sink"), _chk(sink_scores, torch.float16, "sink_scores"), n_sink,
                        float(inv))
    kp, kh = _act(k, "k")
    vp, vh = _act(v, "v")
    qp, qh = _act(q, "q")
    if not (kh == vh == qh):
        raise ValueError("q, k, v must share one dtype (fp32 or fp16)")
    with _Dev(kmat):
        nbytes = _L().kvq_score_k_workspace_bytes(bits, 1, H)
        ws = _workspace(kmat.device, nbytes, slot="score")
        _lib.check(_L().kvq_decode_prologue(
            bits, _i(kmat, "kcache"), _f(klut, "lookup_table"), _f(klut_off, "lut_off"), kp, _f(lo, "lower"),
            _f(hi, "upper"), _f(koutl, "outliers"), _i(kidx, "outlier_indices"), int(kcol), _i(vmat, "vcache"),
            _f(vlut_rows, "lookup_table"), _f(vlut_sorted, "lut"), vp, _f(voutl, "outliers"),
            _i(vidx, "outlier_indices"), int(vcol), qp, kh, int(thr_k), H, hd, max_len,
            *_mirror(koutl_t, kidx_t, thr_k, max_len), None if klut_ends is None else _f(klut_ends, "lut_ends"),
            None if klut_score is None else _f(klut_score, "lut_score"), _vnorm(vnorm),
            None if sk is None else ctypes.byref(sk), ws.data_ptr(), ws.numel(), _stream()), "kvq_decode_prologue")
    return ws


def score_k_prepared(bits, mat, mul, lut, L, theta, pos_offset, ws, outliers=None, outlier_indices=None,
                     accumulate=False):
    """score kernel only (q_len = 1); tables + q were written to `ws` by decode_prologue."""
    H, hd, max_len = _cache_dims(mat, bits)
    n_out = 0 if outliers is None else outliers.shape[1]
    with _Dev(mat):
        _lib.check(_L().kvq_score_k_prepared(
            bits, _i(mat, "mat"), _f(mul, "mul"), _f(lut, "lookup_table"), H, hd, int(L), max_len, float(theta),
            int(pos_offset), None if outliers is None else _f(outliers, "outliers"),
            None if outliers is None else _i(outlier_indices, "outlier_indices"), n_out, 1 if accumulate else 0,
            ws.data_ptr(), ws.numel(), _stream()), "kvq_score_k_prepared")


def score_k_prepared_softmax(bits, mat, mul, lut, L, theta, pos_offset, ws, outliers, outlier_indices, inv_sqrt_hd,
                             n_parts, outliers_t=None, outlier_indices_t=None, f16_pair=False):
    """sparse score kernel (tables already in `ws`, accumulate = 0) that also writes the per-(head, tile)
    softmax partials; returns the partials buffer (a workspace: consume it before the next call).
    f16_pair (3 bit, with the mirror): read the fp16 pair-sum tables (include/kvq.h: KVQ_SCORE_F16_PAIR_TABLES)."""
    H, hd, max_len = _cache_dims(mat, bits)
    with _Dev(mat):
        parts = _workspace(mat.device, H * n_parts * 8, slot="softmax")
        _lib.check(_L().kvq_score_k_prepared_softmax_ex(
            bits, _i(mat, "mat"), _f(mul, "mul"), _f(lut, "lookup_table"), H, hd, int(L), max_len, float(theta),
            int(pos_offset), _f(outliers, "outliers"), _i(outlier_indices, "outlier_indices"), outliers.shape[1],
            *_mirror(outliers_t, outlier_indices_t, outliers.shape[1] // 2, max_len),
            ws.data_ptr(), ws.numel(), float(inv_sqrt_hd), parts.data_ptr(), n_parts,
            _lib.SCORE_F16_PAIR_TABLES if f16_pair else 0, _stream()),
            "kvq_score_k_prepared_softmax_ex")
    return parts


def softmax_finish(scores, parts, n_parts, inv_sqrt_hd, sink_scores=None, v_sink=None, sink_out=None):
    """second softmax pass on partials written by score_k_prepared_softmax; scores f32 [H, L].
    v_sink (f16 [H, n_sink, 128]) + sink_out (f32 [1, H, 128]): the sink tokens' share of the attention output is
    written to sink_out by the same launch (mix_v then accumulates onto it)."""
    H, L = scores.shape
    n_sink = 0 if sink_scores is None else sink_scores.shape[1]
    probs = torch.empty_like(scores)
    sink_probs = None if n_sink == 0 else torch.empty_like(sink_scores)
    if v_sink is not None and (n_sink == 0 or sink_out is None or tuple(v_sink.shape) != (H, n_sink, 128)):
        raise ValueError("v_sink needs sink_scores, sink_out and the shape [H, n_sink, 128]")
    with _Dev(scores):
        _lib.check(_L().kvq_softmax_finish(
            _f(scores, "scores"), None if n_sink == 0 else _chk(sink_scores, torch.float16, "sink_scores"),
            parts.data_ptr(), n_parts, _f(probs, "probs"), None if n_sink == 0 else sink_probs.data_ptr(), H, L,
            n_sink, float(inv_sqrt_hd), None if v_sink is None else _chk(v_sink, torch.float16, "v_sink"),
            None if v_sink is None else _f(sink_out, "sink_out"), _stream()), "kvq_softmax_finish")
    return probs, sink_probs


def score_k_softmax(bits, mat, mul, lut, L, theta, pos_offset, ws, outliers, outlier_indices, inv_sqrt_hd,
                    sink_scores=None, outliers_t=None, outlier_indices_t=None, v_sink=None, sink_out=None):
    """q.K^T (tables already in `ws`) + softmax: `mul` [1, H, L] receives the raw scores; returns
    (probs f32 [H, L] holding fp16 values, sink_probs f16 [H, n_sink] or None).  Sparse caches take the
    score kernel with the first softmax pass fused in (2 launches), others score_k_prepared +
    softmax_scale (3 launches)."""
    n_parts = _L().kvq_score_k_softmax_parts(bits, int(L), 1 if outliers is not None else 0)
    if n_parts == 0:
        if v_sink is not None:
            raise ValueError("v_sink needs the fused softmax partials (sparse cache)")
        score_k_prepared(bits, mat, mul, lut, L, theta, pos_offset, ws, outliers, outlier_indices)
        return softmax_scale(mul[0], inv_sqrt_hd, sink_scores)
    parts = score_k_prepared_softmax(bits, mat, mul, lut, L, theta, pos_offset, ws, outliers, outlier_indices,
                                     inv_sqrt_hd, n_parts, outliers_t, outlier_indices_t)
    return softmax_finish(mul[0], parts, n_parts, inv_sqrt_hd, sink_scores, v_sink, sink_out)


def score_k_mix_v(bits, kmat, scores, klut, L, theta, pos_offset, ws, koutliers, koutlier_indices, inv_sqrt_hd,
                  vmat, out, vlut_rows, voutliers, voutlier_indices, sink_scores=None, koutliers_t=None,
                  koutlier_indices_t=None, v_sink=None, f16_pair=False):
    """q.K^T (tables already in `ws`) -> softmax -> p.V of one decode token in two streaming launches + the slab
    reduce: the score kernel writes raw scores and per-tile softmax partials, the p.V kernel normalises on the way
    (kvq_mix_v_softmax).  scores [1, H, L] scratch, out f32 [1, H, hd].  Returns sink_probs (f16 [H, n_sink]) or None."""
    n_parts = _L().kvq_score_k_softmax_parts(bits, int(L), 1 if koutliers is not None else 0)
    if n_parts == 0 or voutliers is None:
        probs, sink_probs = score_k_softmax(bits, kmat, scores, klut, L, theta, pos_offset, ws, koutliers,
                                            koutlier_indices, inv_sqrt_hd, sink_scores, koutliers_t,
                                            koutlier_indices_t, v_sink, out)
        mix_v(bits, probs.unsqueeze(0), vmat, out, vlut_rows, L, voutliers, voutlier_indices,
              accumulate=v_sink is not None)
        return sink_probs
    parts = score_k_prepared_softmax(bits, kmat, scores, klut, L, theta, pos_offset, ws, koutliers, koutlier_indices,
                                     inv_sqrt_hd, n_parts, koutliers_t, koutlier_indices_t, f16_pair=f16_pair)
    return mix_v_softmax(bits, scores, parts, n_parts, inv_sqrt_hd, vmat, out, vlut_rows, L, voutliers,
                         voutlier_indices, sink_scores, v_sink)


def mix_v_softmax(bits, scores, parts, n_parts, inv_sqrt_hd, mat, mul, lut_rows, L, outliers, outlier_indices,
                  sink_scores=None, v_sink=None):
    """kvq_mix_v_softmax: raw scores [1, H, L] + the score kernel's softmax partials -> mul f32 [1, H, hd]
    (overwritten; with v_sink f16 [H, n_sink, 128] it includes the sink tokens' share); returns sink_probs
    (f16 [H, n_sink]) or None."""
    H, hd, max_len = _cache_dims(mat, bits)
    n_sink = 0 if sink_scores is None else sink_scores.shape[1]
    sink_probs = None if n_sink == 0 else torch.empty_like(sink_scores)
    with _Dev(mat):
        nbytes = _L().kvq_mix_v_workspace_bytes(bits, 1, H, hd, int(L))
        wsv = _workspace(mat.device, nbytes)
        # room for the probabilities of the library's two-pass route: always handed over, so that the shapes its
        # streaming kernel does not take (unaligned rows or tables, H > 128, more than 2^31 packed words, ...) fall back
        # inside the library whatever its predicate is -- the two checks cannot drift apart (H * L * 4 bytes, cached)
        probs = _workspace(mat.device, H * int(L) * 4, slot="probs")
        _lib.check(_L().kvq_mix_v_softmax(
            bits, _f(scores, "scores"), parts.data_ptr(), n_parts, float(inv_sqrt_hd),
            None if n_sink == 0 else _chk(sink_scores, torch.float16, "sink_scores"),
            None if n_sink == 0 else sink_probs.data_ptr(), n_sink,
            None if v_sink is None else _chk(v_sink, torch.float16, "v_sink"),
            None if probs is None else probs.data_ptr(),
            _i(mat, "mat"), _f(mul, "mul"), _f(lut_rows, "lookup_table"), H, hd, int(L), max_len,
            _fo(outliers, "outliers"), _i(outlier_indices, "outlier_indices"), outlier_indices.shape[1], 0,
            wsv.data_ptr(), wsv.numel(), _stream()), "kvq_mix_v_softmax")
    return sink_probs


# ---- token-sharded single stream ---------------------------------------------------------------------------------
def _sinks_struct(sinks, H):
    """(k_sink f16 [H, 128, n_sink], sink_scores f16 [H, n_sink] (out), inv_sqrt_hd) -> struct kvq_sinks, or None"""
    if sinks is None:
        return None
    k_sink, sink_scores, inv = sinks
    n_sink = sink_scores.shape[1]
    if tuple(k_sink.shape) != (H, 128, n_sink) or sink_scores.shape[0] != H:
        raise ValueError("sinks: k_sink [H, 128, n_sink] and sink_scores [H, n_sink] expected")
    return _lib.Sinks(_chk(k_sink, torch.float16, "k_sink"), _chk(sink_scores, torch.float16, "sink_scores"), n_sink, float(inv))


def score_k_tables(bits, q, lut, H, sinks=None):
    """query-premultiplied K tables into the score workspace without an append (kvq_score_k_tables); returns the
    workspace tensor for score_k_prepared[_softmax].  sinks = (k_sink, sink_scores (out), inv_sqrt_hd): the scaled
    scores of the fp16 sink tokens this shard holds are written too."""
    qp, qh = _act(q, "q")
    sk = _sinks_struct(sinks, H)
    with _Dev(lut):
        nbytes = _L().kvq_score_k_workspace_bytes(bits, 1, H)
        ws = _workspace(lut.device, nbytes, slot="score")
        _lib.check(_L().kvq_score_k_tables(bits, qp, qh, _f(lut, "lookup_table"), H, 128,
                                           None if sk is None else ctypes.byref(sk), ws.data_ptr(), ws.numel(),
                                           _stream()), "kvq_score_k_tables")
    return ws


def softmax_stats(parts, n_parts, H, stats, sink_scores=None):
    """(max, normaliser) of every head's scaled scores from the score kernel's partials (+ the fp16 sink tokens' scaled
    scores f16 [H, n_sink] of the shard that holds them) -> stats f32 [H, 2]"""
    n_sink = 0 if sink_scores is None else sink_scores.shape[1]
    with _Dev(stats):
        _lib.check(_L().kvq_softmax_stats(parts.data_ptr(), int(n_parts),
                                          None if n_sink == 0 else _chk(sink_scores, torch.float16, "sink_scores"), n_sink,
                                          int(H), _f(stats, "stats"), _stream()),
                   "kvq_softmax_stats")


def combine_shards(packed, n_shards, H, hd, out):
    """exact merge of n_shards records [H*hd out | H x (M, Z)] (f32, contiguous) -> out f32 [1, H, hd]"""
    with _Dev(out):
        _lib.check(_L().kvq_combine_shards(_f(packed, "packed"), int(n_shards), int(H), int(hd), _f(out, "out"), _stream()),
                   "kvq_combine_shards")


def extract_heads(bits, h0, n_heads, src_k, src_v, dst_k, dst_v, src_col, dst_col, n):
    """columns [src_col, src_col + n) of the full-width caches src_k / src_v (QuantK / QuantV) -> columns [dst_col, ...)
    of the head shard dst_k / dst_v that holds heads [h0, h0 + n_heads): packed words, V codebook rows, the shard's share
    of the outlier rows (+ K mirror) -- kvq_extract_heads"""
    H, hd, src_max = _cache_dims(src_k.kcache, bits)
    Hs, _, dst_max = _cache_dims(dst_k.kcache, bits)
    if Hs != n_heads or _cache_dims(dst_v.vcache, bits)[0] != n_heads:
        raise ValueError("extract_heads: the shard caches hold %d heads, not %d" % (Hs, n_heads))
    sparse = src_k.include_sparse
    if sparse and (src_k.outliers is None or dst_k.outliers is None):
        raise NotImplementedError("extract_heads reads the reference outlier format (not compact caches)")
    n_out = int(src_k.num_outliers) if sparse else 0
    if sparse and (dst_k.num_outliers != n_out or dst_v.num_outliers != n_out):
        raise ValueError("extract_heads: the shard's outlier rows must be as wide as the full token's (%d)" % n_out)
    with _Dev(dst_k.kcache):
        _lib.check(_L().kvq_extract_heads(
            bits, H, hd, int(h0), int(n_heads), n_out, _i(src_k.kcache, "src kcache"), _i(src_v.vcache, "src vcache"),
            src_max, int(src_col), _fo(src_k.outliers if sparse else None, "src k outliers"),
            _io(src_k.outlier_indices if sparse else None, "src k outlier_indices"),
            _fo(src_v.outliers if sparse else None, "src v outliers"),
            _io(src_v.outlier_indices if sparse else None, "src v outlier_indices"),
            _f(src_v.lookup_table, "src v lookup_table"), _i(dst_k.kcache, "dst kcache"), _i(dst_v.vcache, "dst vcache"),
            dst_max, int(dst_col), _fo(dst_k.outliers if sparse else None, "dst k outliers"),
            _io(dst_k.outlier_indices if sparse else None, "dst k outlier_indices"),
            _fo(getattr(dst_k, "outliers_t", None) if sparse else None, "dst k outliers_t"),
            _io(getattr(dst_k, "outlier_indices_t", None) if sparse else None, "dst k outlier_indices_t"),
            _fo(dst_v.outliers if sparse else None, "dst v outliers"),
            _io(dst_v.outlier_indices if sparse else None, "dst v outlier_indices"),
            _f(dst_v.lookup_table, "dst v lookup_table"),
            None if src_v.lookup_table2 is None else _f(src_v.lookup_table2, "src v lookup_table2"),
            None if src_v.lookup_table2 is None else _f(dst_v.lookup_table2, "dst v lookup_table2"),
            int(n), _stream()), "kvq_extract_heads")


# ---- one decode token through one layer, one library call -----------------------------------------------------------
def make_layer(kc, vc, table, lut_off):
    """struct kvq_layer for a (QuantK, QuantV) pair: every pointer of the layer's compressed cache, built once and
    cached by the caller (the buffers are preallocated and never move).  Returns (struct, keep-alive tuple)."""
    H, hd, max_len = _cache_dims(kc.kcache, kc.bits)
    thr_k = kc.num_outliers // 2
    if _mirror(kc.outliers_t, kc.outlier_indices_t, thr_k, max_len)[1] is None:
        raise ValueError("decode_step needs the token-contiguous outlier mirror")
    vn = vc.vnorm_args()
    vstruct = None
    if vn is not None:
        rows2, ns, no, zp2, quirk = vn
        vstruct = _lib.VNorm(None if rows2 is None else _f(rows2, "lookup_table2"), float(ns), float(no), 1 if zp2 else 0,
                             1 if quirk else 0)
    mix = vc.mix_table()
    ly = _lib.Layer(
        kc.bits, H, hd, thr_k, max_len, float(kc.rope_theta), int(kc.first_few_fp16),
        _i(kc.kcache, "kcache"), _f(kc.lookup_table, "lookup_table"), _f(lut_off, "lut_off"),
        _f(kc.outlier_threshold_lower, "lower"), _f(kc.outlier_threshold_upper, "upper"), _fo(kc.outliers, "outliers"),
        _io(kc.outlier_indices, "outlier_indices"), _fo(kc.outliers_t, "outliers_t"), _i(kc.outlier_indices_t, "outlier_indices_t"),
        None if kc.lut_ends is None else _f(kc.lut_ends, "lut_ends"), None if table is kc.lookup_table else _f(table, "lut_score"),
        _i(vc.vcache, "vcache"), _f(vc.lookup_table, "lookup_table"), _f(vc.lut, "lut"), _fo(vc.outliers, "outliers"),
        _i(vc.outlier_indices, "outlier_indices"), None if vstruct is None else ctypes.pointer(vstruct),
        None if mix is vc.lookup_table else _f(mix, "lookup_table2"),
        (_lib.LAYER_SCORE_F16_PAIR if (kc.bits == 3 and getattr(kc, "score_f16_pair", False)) else 0) |
        (_lib.LAYER_SCORE_F32_PAIR if (kc.bits == 3 and getattr(kc, "score_f32_pair", False)) else 0))
    return ly, (vstruct, table, lut_off, mix)


def decode_step(layer, col, q, k, v, out, fuse_softmax, sinks=None, v_sink=None, sink_probs=None):
    """kvq_decode_step: prologue + q.K^T + softmax + p.V + reduce of one layer from ONE library call.  layer: struct
    from make_layer; col: append column (cached tokens before this one, sink tokens not counted); q [H, 128], k, v
    [C]: all fp16 or all fp32; out f32 [1, H, hd].  sinks = (k_sink, sink_scores (out), inv_sqrt_hd) with v_sink /
    sink_probs, or None."""
    kp, kh = _act(k, "k")
    vp, vh = _act(v, "v")
    qp, qh = _act(q, "q")
    if not (kh == vh == qh):
        raise ValueError("q, k, v must share one dtype (fp32 or fp16)")
    sk = None
    if sinks is not None:
        k_sink, sink_scores, inv = sinks
        sk = _lib.Sinks(_chk(k_sink, torch.float16, "k_sink"), _chk(sink_scores, torch.float16, "sink_scores"),
                        sink_scores.shape[1], float(inv))
    with _Dev(out):
        nbytes = _L().kvq_decode_step_workspace_bytes(layer.bits, layer.H, layer.hd, int(col) + 1)
        ws = _workspace(out.device, nbytes + 256, slot="step")
        base = (ws.data_ptr() + 255) & ~255
        _lib.check(_L().kvq_decode_step(
            ctypes.byref(layer), int(col), int(col), qp, kp, vp, kh, None if sk is None else ctypes.byref(sk),
            None if v_sink is None else _chk(v_sink, torch.float16, "v_sink"),
            None if sink_probs is None else sink_probs.data_ptr(), _f(out, "out"), int(fuse_softmax),
            base, ws.numel() - (base - ws.data_ptr()), _stream()), "kvq_decode_step")


# ---- prefill attention on the matrix cores ----------------------------------------------------------------------
def prefill_attention(q, k, v, softmax_scale=None):
    """causal attention of a prompt: q, k, v fp16 [H, S, 128] views (any strides with a contiguous last dimension,
    e.g. transposes of [S, H, 128]) -> fp16 [S, H*128] (token-major, ready for o_proj)."""
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float16 and t.dim() == 3 and t.stride(2) == 1):
            raise ValueError("%s: expected an fp16 GPU tensor [H, S, head_dim] with a contiguous last dimension" % name)
    H, S, hd = q.shape
    if k.shape != q.shape or v.shape != q.shape:
        raise ValueError("q, k, v must have the same shape")
    out = torch.empty((S, H, hd), dtype=torch.float16, device=q.device)
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / (hd ** 0.5)
    with _Dev(q):
        _lib.check(_L().kvq_prefill_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), H, S, hd,
                                             q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0),
                                             v.stride(1), out.stride(1), out.stride(0), scale, _stream()),
                   "kvq_prefill_attention")
    return out.view(S, H * hd)


def _step_ws(dev, bits, H, hd, L):
    nbytes = _L().kvq_decode_step_workspace_bytes(bits, H, hd, int(L))
    ws = _workspace(dev, nbytes + 256, slot="step")
    return (ws.data_ptr() + 255) & ~255, ws.numel() - 256, ws


def head_shard_step(full_layer, shard_layer, h0, col, q, k, v, out, fuse_softmax, sinks=None, v_sink=None, sink_probs=None):
    """kvq_head_shard_step: whole-token append into the staging cache -> extract of the shard's heads -> the shard's
    attention, ONE library call.  q [n_heads, 128] (the shard's heads), k, v [C] (the whole token): all fp16 or all fp32;
    out f32 [1, n_heads, hd]; sinks = (k_sink, sink_scores (out), inv_sqrt_hd) of the shard's heads."""
    kp, kh = _act(k, "k")
    vp, vh = _act(v, "v")
    qp, qh = _act(q, "q")
    if not (kh == vh == qh):
        raise ValueError("q, k, v must share one dtype (fp32 or fp16)")
    sk = _sinks_struct(sinks, shard_layer.H)
    with _Dev(out):
        base, nb, keep = _step_ws(out.device, shard_layer.bits, shard_layer.H, shard_layer.hd, int(col) + 1)
        _lib.check(_L().kvq_head_shard_step(
            ctypes.byref(full_layer), ctypes.byref(shard_layer), int(h0), int(col), qp, kp, vp, kh,
            None if sk is None else ctypes.byref(sk), None if v_sink is None else _chk(v_sink, torch.float16, "v_sink"),
            None if sink_probs is None else sink_probs.data_ptr(), _f(out, "out"), int(fuse_softmax), base, nb, _stream()),
            "kvq_head_shard_step")

"""Drop-in replacement of the reference's ``quant_cuda`` extension module:
the same 34 function names, positional arguments and in-place semantics
(/root/reference/deployment/kvquant/quant_cuda.cpp:401-436), implemented by
hand-written gfx950 kernels behind the C ABI of include/kvq.h.

Differences from the reference, all on the safe side:
  * kernels run on torch's CURRENT stream (the reference uses the null stream);
  * argument dtype / device / contiguity are checked and raise ValueError
    (the reference checks nothing and faults);
  * no atomics: results are deterministic run to run.
"""
import torch

from . import _lib

_ws = {}  # device index -> workspace tensor for the V matvec partial sums


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


def _cache_dims(mat, bits):
    if mat.dim() != 3:
        raise ValueError("mat: expected [num_heads, head_dim/32*bits, max_len]")
    H, W, max_len = mat.shape
    if W % bits:
        raise ValueError("mat.shape[1]=%d is not a multiple of bits=%d" % (W, bits))
    return H, W // bits * 32, max_len


def _workspace(device, nbytes):
    key = device.index if device.index is not None else torch.cuda.current_device()
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
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


def _make(bits):
    L = _lib.lib
    ns = {}

    def appendvecK(mat, lookup_table, newvec, kcachelen):
        H, hd, max_len = _cache_dims(mat, bits)
        with _Dev(mat):
            _lib.check(L().kvq_append_k(bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"),
                                        _f(newvec, "newvec"), H, hd, max_len, int(kcachelen), _stream()),
                       "vecquant%dappendvecK" % bits)

    def appendvecV(mat, lookup_table, newvec, vcachelen):
        H, hd, max_len = _cache_dims(mat, bits)
        with _Dev(mat):
            _lib.check(L().kvq_append_v(bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"),
                                        _f(newvec, "newvec"), H, hd, max_len, int(vcachelen), _stream()),
                       "vecquant%dappendvecV" % bits)

    def appendvecKsparse(mat, lookup_table, newvec, outliers_rescaled, outlier_threshold_lower,
                         outlier_threshold_upper, kcachelen):
        H, hd, max_len = _cache_dims(mat, bits)
        with _Dev(mat):
            _lib.check(L().kvq_append_k_sparse(
                bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"), _f(newvec, "newvec"),
                _f(outliers_rescaled, "outliers_rescaled"), _f(outlier_threshold_lower, "lower"),
                _f(outlier_threshold_upper, "upper"), H, hd, max_len, int(kcachelen), _stream()),
                "vecquant%dappendvecKsparse" % bits)

    def appendvecKsparseParallel(mat, lookup_table, newvec, outliers_rescaled, outlier_threshold_lower,
                                 outlier_threshold_upper):
        H, hd, max_len = _cache_dims(mat, bits)
        S = newvec.shape[-1]
        with _Dev(mat):
            _lib.check(L().kvq_pack_k_sparse_parallel(
                bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"), _f(newvec, "newvec"),
                _f(outliers_rescaled, "outliers_rescaled"), _f(outlier_threshold_lower, "lower"),
                _f(outlier_threshold_upper, "upper"), H, hd, S, max_len, 0, _stream()),
                "vecquant%dappendvecKsparseParallel" % bits)

    def appendvecVsparse(mat, lookup_table, newvec, zeropoint, outlier_threshold_lower,
                         outlier_threshold_upper, vcachelen):
        # `zeropoint` is accepted for signature compatibility; the reference kernel ignores it too
        H, hd, max_len = _cache_dims(mat, bits)
        with _Dev(mat):
            _lib.check(L().kvq_append_v_sparse(
                bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"), _f(newvec, "newvec"),
                float(outlier_threshold_lower), float(outlier_threshold_upper), H, hd, max_len,
                int(vcachelen), _stream()), "vecquant%dappendvecVsparse" % bits)

    def appendvecVsparseParallel(mat, lookup_table, newvec, outlier_threshold_lower, outlier_threshold_upper):
        H, hd, max_len = _cache_dims(mat, bits)
        S = newvec.shape[-1]
        with _Dev(mat):
            _lib.check(L().kvq_pack_v_sparse_parallel(
                bits, _i(mat, "mat"), _f(lookup_table, "lookup_table"), _f(newvec, "newvec"),
                _f(outlier_threshold_lower, "lower"), _f(outlier_threshold_upper, "upper"), H, hd, S,
                max_len, 0, _stream()), "vecquant%dappendvecVsparseParallel" % bits)

    def _score(vec, mat, mul, lookup_table, kcachelen, outliers, outlier_indices, theta, pos_offset, name):
        H, hd, max_len = _cache_dims(mat, bits)
        if vec.dim() != 3 or mul.dim() != 3 or vec.shape[0] != mul.shape[0]:
            raise ValueError("vec must be [q_len, H, head_dim] and mul [q_len, H, kcachelen]")
        if mul.shape[2] != kcachelen:
            raise ValueError("mul.shape[2] must equal kcachelen")
        n_out = 0 if outliers is None else outliers.shape[1]
        with _Dev(vec):
            _lib.check(L().kvq_score_k(
                bits, _f(vec, "vec"), _i(mat, "mat"), _f(mul, "mul"), _f(lookup_table, "lookup_table"),
                vec.shape[0], H, hd, int(kcachelen), max_len, float(theta), int(pos_offset),
                None if outliers is None else _f(outliers, "outliers"),
                None if outliers is None else _i(outlier_indices, "outlier_indices"), n_out, 1, _stream()),
                name)

    def k_opt(vec, mat, mul, lookup_table, kcachelen, theta, pos_offset):
        _score(vec, mat, mul, lookup_table, kcachelen, None, None, theta, pos_offset,
               "vecquant%dmatmul_..rope.._opt" % bits)

    def k_opt2(vec, mat, mul, lookup_table, kcachelen, outliers, outlier_indices, theta, pos_offset):
        _score(vec, mat, mul, lookup_table, kcachelen, outliers, outlier_indices, theta, pos_offset,
               "vecquant%dmatmul_..rope.._opt2" % bits)

    def _mix(vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices, name):
        H, hd, max_len = _cache_dims(mat, bits)
        if vec.dim() != 3 or mul.dim() != 3 or vec.shape[0] != mul.shape[0]:
            raise ValueError("vec must be [q_len, H, vcachelen] and mul [q_len, H, head_dim]")
        if vec.shape[2] != vcachelen:
            raise ValueError("vec.shape[2] must equal vcachelen")
        q_len = vec.shape[0]
        n_out = 0 if outliers is None else outliers.shape[1]
        with _Dev(vec):
            nbytes = L().kvq_mix_v_workspace_bytes(bits, q_len, H, hd, int(vcachelen))
            ws = _workspace(vec.device, nbytes)
            _lib.check(L().kvq_mix_v(
                bits, _f(vec, "vec"), _i(mat, "mat"), _f(mul, "mul"), _f(lookup_table, "lookup_table"),
                q_len, H, hd, int(vcachelen), max_len,
                None if outliers is None else _f(outliers, "outliers"),
                None if outliers is None else _i(outlier_indices, "outlier_indices"), n_out, 1,
                ws.data_ptr(), ws.numel(), _stream()), name)

    def v_opt(vec, mat, mul, lookup_table, vcachelen):
        _mix(vec, mat, mul, lookup_table, vcachelen, None, None, "vecquant%dmatmul_..mha.._opt" % bits)

    def v_opt2(vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices):
        _mix(vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices,
             "vecquant%dmatmul_..mha.._opt2" % bits)

    b = str(bits)
    ns["vecquant" + b + "appendvecK"] = appendvecK
    ns["vecquant" + b + "appendvecV"] = appendvecV
    ns["vecquant" + b + "appendvecKsparse"] = appendvecKsparse
    ns["vecquant" + b + "appendvecKsparseParallel"] = appendvecKsparseParallel
    ns["vecquant" + b + "appendvecVsparse"] = appendvecVsparse
    ns["vecquant" + b + "appendvecVsparseParallel"] = appendvecVsparseParallel
    ns["vecquant" + b + "matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt"] = k_opt
    ns["vecquant" + b + "matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2"] = k_opt2
    ns["vecquant" + b + "matmul_nuq_perchannel_transposed_mha_batched_fused_opt"] = v_opt
    ns["vecquant" + b + "matmul_nuq_perchannel_transposed_mha_batched_fused_opt2"] = v_opt2
    return ns


_NS = {}
for _b in (2, 3, 4):
    _NS.update(_make(_b))
globals().update(_NS)

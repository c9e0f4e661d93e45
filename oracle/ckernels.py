"""ctypes binding of oracle/kvq_oracle.c (CPU restatement of the reference kernels).

TEST INFRASTRUCTURE ONLY -- see the header of kvq_oracle.c.  Nothing under
``kvquant_amd/`` imports this module.  All functions take contiguous CPU
``torch`` tensors (int32 cache, float32 everything else) and mutate their
output arguments in place, exactly like the reference ops they restate.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkvq_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "kvq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.kvqo_rope_freq.restype = ctypes.c_float
        _lib.kvqo_rope_freq.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.kvqo_append_k_sparse_orig.restype = ctypes.c_int
        _lib.kvqo_append_v_sparse_orig.restype = ctypes.c_int
        _lib.kvqo_num_threads.restype = ctypes.c_int
    return _lib


def _p(t, dtype):
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle wants contiguous CPU tensors"
    assert t.dtype == dtype, (t.dtype, dtype)
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    return _p(t, torch.float32)


def _i(t):
    return _p(t, torch.int32)


_i64 = ctypes.c_int64
_cf = ctypes.c_float
_ci = ctypes.c_int


def n_rows(C, bits):
    return C // 32 * bits


def _dims(mat):
    """(C-independent) returns (rows_total, max_len) of a [H, W, max_len] cache."""
    assert mat.dim() == 3
    return mat.shape[0] * mat.shape[1], mat.shape[2]


def append_k(bits, mat, lut, x, col):
    C = x.numel()
    lib().kvqo_append_k(bits, _i(mat), _f(lut), _f(x), C, _i64(mat.shape[2]), _i64(col))


def append_v(bits, mat, lut_rows, x, col):
    C = x.numel()
    lib().kvqo_append_v(bits, _i(mat), _f(lut_rows), _f(x), C, _i64(mat.shape[2]), _i64(col))


def append_k_sparse(bits, mat, lut, x, rescaled, lo, hi, col):
    C = x.numel()
    lib().kvqo_append_k_sparse(bits, _i(mat), _f(lut), _f(x), _f(rescaled), _f(lo), _f(hi), C,
                               _i64(mat.shape[2]), _i64(col))


def append_v_sparse(bits, mat, lut_rows, x, lo, hi, col):
    C = x.numel()
    lib().kvqo_append_v_sparse(bits, _i(mat), _f(lut_rows), _f(x), _cf(lo), _cf(hi), C,
                               _i64(mat.shape[2]), _i64(col))


def pack_k_sparse_parallel(bits, mat, lut, x, rescaled, lo, hi, col0=0):
    C = x.shape[0] * x.shape[1]
    S = x.shape[2]
    lib().kvqo_pack_k_sparse_parallel(bits, _i(mat), _f(lut), _f(x), _f(rescaled), _f(lo), _f(hi), C,
                                      _i64(S), _i64(mat.shape[2]), _i64(col0))


def pack_v_sparse_parallel(bits, mat, lut_rows, x, lo, hi, col0=0):
    C = x.shape[0] * x.shape[1]
    S = x.shape[2]
    lib().kvqo_pack_v_sparse_parallel(bits, _i(mat), _f(lut_rows), _f(x), _f(lo), _f(hi), C, _i64(S),
                                      _i64(mat.shape[2]), _i64(col0))


def unpack_codes(bits, mat, C, L):
    codes = torch.empty((L, C), dtype=torch.uint8)
    lib().kvqo_unpack_codes(bits, _i(mat), ctypes.c_void_p(codes.data_ptr()), C, _i64(L),
                            _i64(mat.shape[2]))
    return codes


def set_rope_freqs(theta, table=None):
    """install the platform's theta_j table (float32 [hd/2], CPU) for `theta`, or clear it (None): see the
    comment at kvqo_rope_freq"""
    if table is None:
        lib().kvqo_set_rope_freqs(_cf(0.0), None, 0)
        return
    t = table.detach().cpu().float().contiguous()
    lib().kvqo_set_rope_freqs(_cf(theta), _f(t), t.numel())


def rope_freq(theta, k, hd):
    return lib().kvqo_rope_freq(_cf(theta), k, hd)


def score_k(bits, q, mat, mul, lut, L, theta, pos_offset):
    q_len, H, hd = q.shape
    lib().kvqo_score_k(bits, _f(q), _i(mat), _f(mul), _f(lut), q_len, H, hd, _i64(L),
                       _i64(mat.shape[2]), _cf(theta), _ci(pos_offset))


def spmv_k_rope(outliers, idx, q, mul, L, theta, pos_offset):
    q_len, H, hd = q.shape
    lib().kvqo_spmv_k_rope(_f(outliers), _i(idx), _f(q), _f(mul), _i64(L), H, hd,
                           outliers.shape[1], _cf(theta), _ci(pos_offset))


def mix_v(bits, p, mat, mul, lut_rows, L):
    q_len, H, _ = p.shape
    hd = mul.shape[2]
    assert p.shape[2] == L
    lib().kvqo_mix_v(bits, _f(p), _i(mat), _f(mul), _f(lut_rows), q_len, H, hd, _i64(L),
                     _i64(mat.shape[2]))


def spmv_v(outliers, idx, p, mul, L):
    q_len, H, _ = p.shape
    hd = mul.shape[2]
    lib().kvqo_spmv_v(_f(outliers), _i(idx), _f(p), _f(mul), _i64(L), H, hd, outliers.shape[1])


def append_k_sparse_orig(mat, lut, x, zeropoint, lo, hi, col):
    C = x.numel()
    oi = torch.empty(C, dtype=torch.int32)
    ov = torch.empty(C, dtype=torch.float32)
    n = lib().kvqo_append_k_sparse_orig(_i(mat), _f(lut), _f(x), _f(zeropoint), _f(lo), _f(hi),
                                        _i(oi), _f(ov), C, _i64(mat.shape[2]), _i64(col))
    return oi[:n].clone(), ov[:n].clone()


def append_v_sparse_orig(mat, lut_rows, x, zeropoint, lo, hi, col):
    C = x.numel()
    oi = torch.empty(C, dtype=torch.int32)
    ov = torch.empty(C, dtype=torch.float32)
    n = lib().kvqo_append_v_sparse_orig(_i(mat), _f(lut_rows), _f(x), _cf(zeropoint), _cf(lo),
                                        _cf(hi), _i(oi), _f(ov), C, _i64(mat.shape[2]), _i64(col))
    return oi[:n].clone(), ov[:n].clone()


def spmv_k_rope_csr(rowptr, cols, vals, q, mul, num_rows, L, theta, pos_offset):
    hd = q.shape[2]
    lib().kvqo_spmv_k_rope_csr(_i(rowptr), _i(cols), _f(vals), _f(q), _f(mul), _i64(num_rows),
                               _i64(L), hd, _cf(theta), _ci(pos_offset))


def spmv_v_csc(colptr, rows, vals, p, mul, num_cols, L):
    hd = mul.shape[2]
    lib().kvqo_spmv_v_csc(_i(colptr), _i(rows), _f(vals), _f(p), _f(mul), _i64(num_cols), _i64(L), hd)


def sim_decode_step(khat, vhat, q, H, hd, theta, pos_offset):
    L = khat.shape[0]
    out = torch.empty(H * hd, dtype=torch.float32)
    scores = torch.empty((H, L), dtype=torch.float32)
    lib().kvqo_sim_decode_step(_f(khat), _f(vhat), _f(q), _f(out), _f(scores), H, hd, _i64(L),
                               _cf(theta), _ci(pos_offset))
    return out


def num_threads():
    return lib().kvqo_num_threads()

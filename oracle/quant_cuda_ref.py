"""The reference's ``quant_cuda`` module surface (34 names, argument order of
/root/reference/deployment/kvquant/quant_cuda.cpp:401-436) implemented on CPU
tensors by the C oracle.

TEST INFRASTRUCTURE ONLY.  Used (a) by tests/golden/gen_golden.py to stand in
for the CUDA extension underneath the reference's own QuantK/QuantV Python
classes, and (b) by the tests as the checker for ``kvquant_amd.quant_cuda``.
"""
import sys
import types

import torch

from . import ckernels as ck


def _mk(bits):
    ns = {}

    def appendvecK(mat, lookup_table, newvec, kcachelen):
        ck.append_k(bits, mat, lookup_table, newvec, kcachelen)

    def appendvecV(mat, lookup_table, newvec, vcachelen):
        ck.append_v(bits, mat, lookup_table, newvec, vcachelen)

    def appendvecKsparse(mat, lookup_table, newvec, outliers_rescaled, lo, hi, kcachelen):
        ck.append_k_sparse(bits, mat, lookup_table, newvec, outliers_rescaled, lo, hi, kcachelen)

    def appendvecKsparseParallel(mat, lookup_table, newvec, outliers_rescaled, lo, hi):
        ck.pack_k_sparse_parallel(bits, mat, lookup_table, newvec, outliers_rescaled, lo, hi)

    def appendvecVsparse(mat, lookup_table, newvec, zeropoint, lo, hi, vcachelen):
        ck.append_v_sparse(bits, mat, lookup_table, newvec, float(lo), float(hi), vcachelen)

    def appendvecVsparseParallel(mat, lookup_table, newvec, lo, hi):
        ck.pack_v_sparse_parallel(bits, mat, lookup_table, newvec, lo.contiguous(), hi.contiguous())

    def k_opt(vec, mat, mul, lookup_table, kcachelen, theta, pos_offset):
        ck.score_k(bits, vec, mat, mul, lookup_table, kcachelen, theta, pos_offset)

    def k_opt2(vec, mat, mul, lookup_table, kcachelen, outliers, outlier_indices, theta, pos_offset):
        ck.score_k(bits, vec, mat, mul, lookup_table, kcachelen, theta, pos_offset)
        ck.spmv_k_rope(outliers, outlier_indices, vec, mul, kcachelen, theta, pos_offset)

    def v_opt(vec, mat, mul, lookup_table, vcachelen):
        ck.mix_v(bits, vec, mat, mul, lookup_table, vcachelen)

    def v_opt2(vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices):
        ck.mix_v(bits, vec, mat, mul, lookup_table, vcachelen)
        ck.spmv_v(outliers, outlier_indices, vec, mul, vcachelen)

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
    _NS.update(_mk(_b))


# ---- uncapped CSR/CSC ("orig") variants, 4-bit only ---------------------
def _start_rows(ptr, num_threads, nnz):
    """start row of every balanced thread (reference: KCU:797-823).  Thread i
    starts at nnz index i*per; its start row is the row containing it, or -1
    past the end."""
    per = (nnz + num_threads - 1) // num_threads if num_threads > 0 else 0
    out = torch.full((max(num_threads, 1),), -1, dtype=torch.int32)
    for i in range(num_threads):
        s = i * per
        if s < nnz:
            out[i] = int(torch.searchsorted(ptr, torch.tensor(s, dtype=ptr.dtype), right=True)) - 1
    return out


def vecquant4appendvecKsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_rows,
                                  lo, hi, kcachelen):
    """Returns [rows(ptr), cols, vals, start_rows, num_threads(cpu int[1]), outlier_count]
    like KCU:691-830: the CSR arrays grow by concatenation; 10 nnz per thread."""
    idx, v = ck.append_k_sparse_orig(mat, lookup_table, newvec, zeropoint, lo, hi, kcachelen)
    if row.numel() == 0:
        row = torch.zeros(1, dtype=torch.int32)
    rows = torch.cat((row.int(), (row[-1:].int() + idx.numel())))
    cols = torch.cat((col.int(), idx))
    vals = torch.cat((val.float(), v))
    nnz = int(rows[-1])
    num_threads = (nnz + 9) // 10
    start = _start_rows(rows, num_threads, nnz)
    return [rows, cols, vals, start, torch.tensor([num_threads], dtype=torch.int32),
            torch.tensor([idx.numel()], dtype=torch.int32)]


def vecquant4appendvecVsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_cols,
                                  lo, hi, vcachelen):
    idx, v = ck.append_v_sparse_orig(mat, lookup_table, newvec, float(zeropoint), float(lo), float(hi),
                                     vcachelen)
    if col.numel() == 0:
        col = torch.zeros(1, dtype=torch.int32)
    cols = torch.cat((col.int(), (col[-1:].int() + idx.numel())))
    rows = torch.cat((row.int(), idx))
    vals = torch.cat((val.float(), v))
    nnz = int(cols[-1])
    num_threads = (nnz + 9) // 10
    start = _start_rows(cols, num_threads, nnz)
    return [rows, cols, vals, start, torch.tensor([num_threads], dtype=torch.int32),
            torch.tensor([idx.numel()], dtype=torch.int32)]


def vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig(
        vec, mat, mul, lookup_table, kcachelen, rows, cols, startrows, spmat, num_rows, num_threads,
        nnz, rope_theta, pos_offset):
    ck.score_k(4, vec, mat, mul, lookup_table, kcachelen, rope_theta, pos_offset)
    if nnz > 0:
        ck.spmv_k_rope_csr(rows.int().contiguous(), cols.int().contiguous(), spmat, vec, mul, num_rows,
                           kcachelen, rope_theta, pos_offset)


def vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig(
        vec, mat, mul, lookup_table, vcachelen, rows, cols, startcols, spmat, num_rows, num_threads,
        nnz):
    ck.mix_v(4, vec, mat, mul, lookup_table, vcachelen)
    if nnz > 0:
        ck.spmv_v_csc(cols.int().contiguous(), rows.int().contiguous(), spmat, vec, mul, num_rows,
                      vcachelen)


_NS["vecquant4appendvecKsparseorig"] = vecquant4appendvecKsparseorig
_NS["vecquant4appendvecVsparseorig"] = vecquant4appendvecVsparseorig
_NS["vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig"] = \
    vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig
_NS["vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig"] = \
    vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig

globals().update(_NS)
NAMES = sorted(_NS)
assert len(NAMES) == 34, len(NAMES)


def as_module(name="quant_cuda"):
    """A module object exposing the 34 legacy names (for sys.modules injection
    in tests/golden/gen_golden.py)."""
    m = types.ModuleType(name)
    for k, v in _NS.items():
        setattr(m, k, v)
    return m

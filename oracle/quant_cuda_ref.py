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
def _grow(ptr, minor, val, start, idx, v, pos):
    """CSR/CSC growth of KCU:763-829 (K) / 1010-1060 (V): `ptr` gains one entry (running nnz),
    `minor`/`val` gain the new entries, `start` gains `pos` for every newly needed 10-nnz thread."""
    cnt = idx.numel()
    if ptr.numel() == 0:
        ptr2 = torch.tensor([0, cnt], dtype=torch.int32)
        minor2, val2 = idx, v
        nt = (cnt + 9) // 10
        start2 = torch.full((nt,), pos, dtype=torch.int32)
    else:
        ptr2 = torch.cat((ptr.int(), torch.tensor([minor.numel() + cnt], dtype=torch.int32)))
        if cnt > 0:
            minor2 = torch.cat((minor.int(), idx))
            val2 = torch.cat((val.float(), v))
            nt = (minor2.numel() + 9) // 10
            new_alloc = nt - start.numel()
            start2 = torch.cat((start.int(), torch.full((new_alloc,), pos, dtype=torch.int32))) if new_alloc > 0 else start
        else:
            minor2, val2, start2 = minor, val, start
            nt = (minor2.numel() + 9) // 10
    return ptr2, minor2, val2, start2, nt


def vecquant4appendvecKsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_rows,
                                  lo, hi, kcachelen):
    """returns [rows(ptr), cols, vals, start_rows, num_threads(cpu int[1]), outlier_count] (KCU:691-830)"""
    idx, v = ck.append_k_sparse_orig(mat, lookup_table, newvec, zeropoint, lo, hi, kcachelen)
    rows, cols, vals, start, nt = _grow(row, col, val, start_rows, idx, v, kcachelen)
    return [rows, cols, vals, start, torch.tensor([nt], dtype=torch.int32),
            torch.tensor([idx.numel()], dtype=torch.int32)]


def vecquant4appendvecVsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_cols,
                                  lo, hi, vcachelen):
    """returns [rows, cols(ptr), vals, start_cols, num_threads, outlier_count] (KCU:933-1066)"""
    idx, v = ck.append_v_sparse_orig(mat, lookup_table, newvec, float(zeropoint), float(lo), float(hi),
                                     vcachelen)
    cols, rows, vals, start, nt = _grow(col, row, val, start_cols, idx, v, vcachelen)
    return [rows, cols, vals, start, torch.tensor([nt], dtype=torch.int32),
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

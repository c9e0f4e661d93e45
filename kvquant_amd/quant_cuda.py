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
from . import ops


def _make(bits):
    ns = {}

    def appendvecK(mat, lookup_table, newvec, kcachelen):
        ops.append_k(bits, mat, lookup_table, newvec, kcachelen)

    def appendvecV(mat, lookup_table, newvec, vcachelen):
        ops.append_v(bits, mat, lookup_table, newvec, vcachelen)

    def appendvecKsparse(mat, lookup_table, newvec, outliers_rescaled, outlier_threshold_lower,
                         outlier_threshold_upper, kcachelen):
        ops.append_k_sparse(bits, mat, lookup_table, newvec, outliers_rescaled, outlier_threshold_lower,
                            outlier_threshold_upper, kcachelen)

    def appendvecKsparseParallel(mat, lookup_table, newvec, outliers_rescaled, outlier_threshold_lower,
                                 outlier_threshold_upper):
        ops.pack_k_sparse_parallel(bits, mat, lookup_table, newvec, outliers_rescaled,
                                   outlier_threshold_lower, outlier_threshold_upper, 0)

    def appendvecVsparse(mat, lookup_table, newvec, zeropoint, outlier_threshold_lower,
                         outlier_threshold_upper, vcachelen):
        # `zeropoint` is accepted for signature compatibility; the reference kernel ignores it too
        ops.append_v_sparse(bits, mat, lookup_table, newvec, outlier_threshold_lower,
                            outlier_threshold_upper, vcachelen)

    def appendvecVsparseParallel(mat, lookup_table, newvec, outlier_threshold_lower, outlier_threshold_upper):
        ops.pack_v_sparse_parallel(bits, mat, lookup_table, newvec, outlier_threshold_lower,
                                   outlier_threshold_upper, 0)

    def k_opt(vec, mat, mul, lookup_table, kcachelen, theta, pos_offset):
        ops.score_k(bits, vec, mat, mul, lookup_table, kcachelen, theta, pos_offset)

    def k_opt2(vec, mat, mul, lookup_table, kcachelen, outliers, outlier_indices, theta, pos_offset):
        ops.score_k(bits, vec, mat, mul, lookup_table, kcachelen, theta, pos_offset, outliers, outlier_indices)

    def v_opt(vec, mat, mul, lookup_table, vcachelen):
        ops.mix_v(bits, vec, mat, mul, lookup_table, vcachelen)

    def v_opt2(vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices):
        ops.mix_v(bits, vec, mat, mul, lookup_table, vcachelen, outliers, outlier_indices)

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


# ---- the four uncapped ("orig") entry points, 4 bit only (KCPP:381-399, 432-435) -------------------
def vecquant4appendvecKsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_rows,
                                  outlier_threshold_lower, outlier_threshold_upper, kcachelen):
    return ops.append_k_sparse_orig(mat, lookup_table, newvec, zeropoint, row, col, val, start_rows,
                                    outlier_threshold_lower, outlier_threshold_upper, kcachelen)


def vecquant4appendvecVsparseorig(mat, lookup_table, newvec, zeropoint, row, col, val, start_cols,
                                  outlier_threshold_lower, outlier_threshold_upper, vcachelen):
    return ops.append_v_sparse_orig(mat, lookup_table, newvec, zeropoint, row, col, val, start_cols,
                                    outlier_threshold_lower, outlier_threshold_upper, vcachelen)


def vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig(
        vec, mat, mul, lookup_table, kcachelen, rows, cols, startrows, spmat, num_rows, num_threads, nnz,
        rope_theta, pos_offset):
    """dense q.K^T then the CSR SpMV (KCU:5506-5596); startrows / num_threads only balance the
    reference's threads and do not enter the result."""
    ops.score_k(4, vec, mat, mul, lookup_table, kcachelen, rope_theta, pos_offset)
    if int(nnz) > 0:
        ops.spmv_k_rope_csr(rows.int().contiguous(), cols.int().contiguous(), spmat, vec, mul, num_rows,
                            kcachelen, rope_theta, pos_offset)


def vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig(
        vec, mat, mul, lookup_table, vcachelen, rows, cols, startcols, spmat, num_rows, num_threads, nnz):
    ops.mix_v(4, vec, mat, mul, lookup_table, vcachelen)
    if int(nnz) > 0:
        ops.spmv_v_csc(cols.int().contiguous(), rows.int().contiguous(), spmat, vec, mul, num_rows, vcachelen)


_NS["vecquant4appendvecKsparseorig"] = vecquant4appendvecKsparseorig
_NS["vecquant4appendvecVsparseorig"] = vecquant4appendvecVsparseorig
_NS["vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig"] = \
    vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig
_NS["vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig"] = \
    vecquant4matmul_nuq_perchannel_transposed_mha_batched_fused_opt2_orig
globals().update(_NS)

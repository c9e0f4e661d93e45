"""ctypes binding of libkvq.so (include/kvq.h).  There is NO fallback: if the
library is missing or a call fails this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkvq.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t

class VNorm(ctypes.Structure):
    """struct kvq_vopts of include/kvq.h"""
    _fields_ = [("lut_rows2", ctypes.c_void_p), ("normscale", ctypes.c_float), ("normoffset", ctypes.c_float),
                ("zp_from_rows2", ctypes.c_int), ("reference_tie_quirk", ctypes.c_int)]


_vn = ctypes.POINTER(VNorm)


class Sinks(ctypes.Structure):
    """struct kvq_sinks of include/kvq.h"""
    _fields_ = [("k_sink", ctypes.c_void_p), ("sink_scores", ctypes.c_void_p), ("n_sink", ctypes.c_int),
                ("inv_sqrt_hd", ctypes.c_float)]


_sk = ctypes.POINTER(Sinks)


class Layer(ctypes.Structure):
    """struct kvq_layer of include/kvq.h"""
    _fields_ = [("bits", ctypes.c_int), ("H", ctypes.c_int), ("hd", ctypes.c_int), ("thr_k", ctypes.c_int),
                ("max_len", ctypes.c_int64), ("rope_theta", ctypes.c_float), ("pos_offset", ctypes.c_int),
                ("kmat", ctypes.c_void_p), ("klut", ctypes.c_void_p), ("klut_off", ctypes.c_void_p),
                ("klo", ctypes.c_void_p), ("khi", ctypes.c_void_p), ("koutliers", ctypes.c_void_p),
                ("kidx", ctypes.c_void_p), ("koutliers_t", ctypes.c_void_p), ("kidx_t", ctypes.c_void_p),
                ("klut_ends", ctypes.c_void_p), ("klut_score", ctypes.c_void_p),
                ("vmat", ctypes.c_void_p), ("vlut_rows", ctypes.c_void_p), ("vlut_sorted", ctypes.c_void_p),
                ("voutliers", ctypes.c_void_p), ("vidx", ctypes.c_void_p), ("vnorm", ctypes.POINTER(VNorm)),
                ("v_mix_rows", ctypes.c_void_p), ("flags", ctypes.c_int)]


_ly = ctypes.POINTER(Layer)

# name -> (restype, argtypes); mirrors include/kvq.h one to one
SIGNATURES = {
    "kvq_version": (_i, []),
    "kvq_strerror": (ctypes.c_char_p, [_i]),
    "kvq_last_hip_error": (_i, []),
    "kvq_rope_freqs": (_i, [_f, _vp, _vp]),
    "kvq_append_k": (_i, [_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "kvq_append_v": (_i, [_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "kvq_append_k_sparse": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "kvq_append_v_sparse": (_i, [_i, _vp, _vp, _vp, _f, _f, _i, _i, _i64, _i64, _vp]),
    "kvq_pack_k_sparse_parallel": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i64, _vp]),
    "kvq_pack_v_sparse_parallel": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i64, _vp]),
    "kvq_score_k_workspace_bytes": (_sz, [_i, _i, _i]),
    "kvq_score_k": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "kvq_score_k_mirror": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "kvq_outlier_mirror_rows": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i64, _vp]),
    "kvq_mix_v_workspace_bytes": (_sz, [_i, _i, _i, _i, _i64]),
    "kvq_mix_v": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "kvq_append_k_fused": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _vp, _vp, _vp]),
    "kvq_append_v_fused": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _vn, _vp]),
    "kvq_pack_k_fused": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _vp, _vp, _vp]),
    "kvq_pack_v_fused": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _vn, _vp]),
    "kvq_softmax_workspace_bytes": (_sz, [_i, _i64]),
    "kvq_softmax_scale": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp, _sz, _vp]),
    "kvq_decode_prologue": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                 _vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vn, _sk, _vp, _sz, _vp]),
    "kvq_score_k_prepared": (_i, [_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "kvq_score_k_softmax_parts": (_i, [_i, _i64, _i]),
    "kvq_score_k_head_groups": (_i, [_i, _i64, _i, _i, _i]),
    "kvq_score_k_prepared_softmax": (_i, [_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                          _sz, _f, _vp, _i, _vp]),
    "kvq_score_k_prepared_softmax_ex": (_i, [_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                             _sz, _f, _vp, _i, _i, _vp]),
    "kvq_softmax_finish": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _f, _vp, _vp, _vp]),
    "kvq_mix_v_softmax": (_i, [_i, _vp, _vp, _i, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp, _vp, _i,
                               _i, _vp, _sz, _vp]),
    "kvq_fused_attend_supported": (_i, [_i, _i, _i, _i64, _i64, _i]),
    "kvq_fused_attend_workspace_bytes": (_sz, [_i, _i, _i, _i64]),
    "kvq_fused_attend": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i,
                              _vp, _vp, _vp, _sz, _vp]),
    "kvq_decode_step_workspace_bytes": (_sz, [_i, _i, _i, _i64]),
    "kvq_decode_step": (_i, [_ly, _i64, _i64, _vp, _vp, _vp, _i, _sk, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "kvq_decode_step_events": (_i, [_vp]),
    "kvq_decode_step_route": (_i, []),
    "kvq_decode_steps": (_i, [_i, _ly, _i64, _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "kvq_score_k_tables": (_i, [_i, _vp, _i, _vp, _i, _i, _sk, _vp, _sz, _vp]),
    "kvq_softmax_stats": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "kvq_combine_shards": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "kvq_rope_q_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "kvq_extract_heads": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "kvq_append_kv_fused": (_i, [_ly, _i64, _vp, _vp, _i, _vp]),
    "kvq_attend_step": (_i, [_ly, _i64, _vp, _i, _sk, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "kvq_head_shard_step": (_i, [_ly, _ly, _i, _i64, _vp, _vp, _vp, _i, _sk, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "kvq_prefill_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "kvq_append_k_sparse_orig": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "kvq_append_v_sparse_orig": (_i, [_vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "kvq_spmv_k_rope_csr": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _f, _i, _vp]),
    "kvq_spmv_v_csc": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _vp]),
}

_lib = None
ABI_MAJOR = 4      # include/kvq.h: KVQ_ABI_MAJOR
LAYER_SCORE_F16_PAIR = 1   # kvq_layer.flags
LAYER_SCORE_F32_PAIR = 2
SCORE_F16_PAIR_TABLES = 1  # kvq_score_k_prepared_softmax_ex flags


class KvqError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KvqError(
                "kvquant_amd: %s not found -- build it with `python -m kvquant_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.kvq_version() // 100 != ABI_MAJOR:
            raise KvqError("kvquant_amd: %s is ABI %d, this binding is written for major %d (include/kvq.h: kvq_version) "
                           "-- rebuild it" % (LIB_PATH, l.kvq_version(), ABI_MAJOR))
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        l = lib()
        msg = l.kvq_strerror(rc).decode()
        raise KvqError("%s failed: %s (code %d, hip error %d)" % (what, msg, rc, l.kvq_last_hip_error()))

"""QuantK / QuantV -- the cache-owning operator classes of the reference
(ML = /root/reference/deployment/transformers/src/transformers/models/llama/
modeling_llama.py:352-976 and 978-1385) on MI355X.

Same constructor arguments, attributes (kcache/vcache, outliers,
outlier_indices, lookup_table, klen/vlen, ...) and method signatures, so the
reference's patched LlamaAttention and drivers (deployment/llama.py:186-198)
work unchanged.  What changed underneath:
  * kernels are the gfx950 HIP kernels of libkvq.so, on torch's current stream;
  * NO host round trips: the reference copies the rescaled key / the value
    vector to the CPU and runs torch.topk there every layer every token
    (ML:707-714, 1812-1820); here selection stays on the GPU;
  * LUT construction is vectorised (the reference loops 4096 times in Python,
    ML:464-480) but evaluates the same fp16/fp32 expressions bit for bit;
  * `device` can be chosen per instance (the reference pins everything to
    cuda:0 with .cuda(), ML:392-397).
There is no CPU fallback: the tensors live on a GPU and every op raises if the
HIP library is missing.
"""
import os
import sys

import torch
import torch.nn as nn

from . import ops

ZERO_CODE = {4: 7, 3: 3, 2: 1}  # KCU:2084 / 2442 / 3020


def _threshold_k(sparsity_threshold, hidden_size):
    return int(((1 - sparsity_threshold) / 2) * hidden_size) + 1  # ML:706


def _pack_col0(length, sinks):
    """First compressed column of a parallel_pack.  The reference packs on an empty cache (columns 0..S-1) and
    counts its fp16 sink tokens AFTERWARDS (ML:1877-1888); chunked fills call it again with the sinks already in
    `length`.  Either way the next compressed column is the number of compressed tokens so far."""
    if length == 0:
        return 0
    if length < sinks:
        raise ValueError("parallel_pack while the fp16 sink cache is still filling (len %d < %d)" % (length, sinks))
    return length - sinks


def _default_device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("kvquant_amd.QuantK/QuantV need a GPU (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


class QuantK(nn.Module):
    """Compressed pre-RoPE key cache (ML:352-976)."""

    def __init__(self, bits=2, hidden_size=4096, num_heads=32, max_position_embeddings=-1,
                 include_sparse=False, sparsity_threshold=0.99, rope_theta=10000, use_orig_sparse=False,
                 first_few_fp16=0, device=None, compact=False, outlier_width=None):
        super().__init__()
        if bits not in (2, 3, 4):
            raise ValueError("bits must be 2, 3 or 4")
        if compact and not include_sparse:
            raise ValueError("compact is a format of the Dense-and-Sparse cache (include_sparse)")
        if use_orig_sparse and bits != 4:
            raise ValueError("use_orig_sparse is 4-bit only (as the reference, ML:513)")
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.bits = bits
        self.lut = None
        self.lookup_table = None
        self.zeropoint = None
        self.sparsity_threshold = sparsity_threshold
        self.include_sparse = include_sparse
        self.outlier_threshold_upper = None
        self.outlier_threshold_lower = None
        self.max_len = max_position_embeddings
        self.klen = 0
        dev = _default_device(device)
        self.kcache = torch.zeros((num_heads, (self.head_dim // 32) * bits, self.max_len), dtype=torch.int32,
                                  device=dev)
        # the reference hard-codes 42 columns (ML:396); same value at 0.99 / 4096.  outlier_width: a HEAD SHARD of a wider
        # layer (HeadShard below) keeps rows as wide as the whole token's selection
        self.num_outliers = int(outlier_width) if outlier_width else 2 * _threshold_k(sparsity_threshold, hidden_size)
        # compact=True (opt-in, NOT the reference's format): the outliers of a token are kept ONLY as packed
        # entries fp16 residual << 16 | channel in the token-contiguous mirror -- 168 B per token instead of 336 (+336
        # for the reference-layout rows, which are not kept).  The residual is rounded to fp16 (relative 2^-11 of an
        # outlier's residual); everything else is bit-identical.  Only the GPU-resident decode path (decode_kv,
        # parallel_pack) reads this format.
        self.compact = bool(compact)
        if include_sparse and self.compact:
            self.outliers = self.outlier_indices = self.outliers_t = None
            self.outlier_indices_t = torch.zeros((self.num_outliers, self.max_len), dtype=torch.int32, device=dev)
        elif include_sparse:
            self.outliers = torch.zeros((self.max_len, self.num_outliers), dtype=torch.float32, device=dev)
            self.outlier_indices = torch.zeros((self.max_len, self.num_outliers), dtype=torch.int32, device=dev)
            # token-contiguous mirror of the two buffers above ([slot][token]); kept in step by every append
            # path of this class and read by the decode score kernel (a lane then owns its token's entries:
            # coalesced loads, no segmented scan).  +336 B per token next to the reference's layout.
            self.outliers_t = torch.zeros((self.num_outliers, self.max_len), dtype=torch.float32, device=dev)
            self.outlier_indices_t = torch.zeros((self.num_outliers, self.max_len), dtype=torch.int32, device=dev)
        self.rope_theta = rope_theta
        self.use_orig_sparse = use_orig_sparse
        self.first_few_fp16 = first_few_fp16
        self.norm = False
        self.lookup_table2 = None
        self.lut_ends = None
        # OPT-IN (3 bit, GPU-resident decode path: decode_kv): q.K^T through fp16 PAIR-SUM tables -- the two channels of a
        # rotation pair share (cos, sin), one table entry indexed by both codes holds their pre-added query-premultiplied
        # values, so one LDS look-up + one v_dot2_f32_f16 replace two look-ups and two packed FMAs (include/kvq.h:
        # KVQ_SCORE_F16_PAIR_TABLES): 128K nuq3 q.K^T 88.5 -> 70.6 us.  Entries and (cos, sin) are rounded to fp16, sums
        # accumulate in fp32: <= 4e-4 of a row's largest score against the reference's kernel (contract 1e-3) -- but that is
        # ~0.4 ulp of the fp16 value the reference rounds every score to (ML:873), so four scores in ten land on the
        # neighbouring fp16 value and the attention OUTPUT moves by 1.4e-3 .. 1.8e-3 (held to 4e-3 by tests/test_atsize_gpu.py).  Default False:
        # the fp32 tables, whose scores round like the reference's.  Env KVQ_SCORE_F16=1 turns it on for every 3-bit cache.
        # 3 bit, decode through kvq_decode_step, >= 16K cached tokens: the EXACT fp32 pair-sum tables (round 6; include/kvq.h
        # KVQ_SCORE_F32_PAIR_TABLES) -- one look-up and one packed FMA per two codes, the same fp32 arithmetic in another
        # order, one 1024-lane score workgroup per CU.  Opt-in (KVQ_SCORE_F32_PAIR=1): measured EQUAL to the per-channel
        # tables at 128K (81.7 vs 81.9 us), slower at 32K (42.2 vs 39.4), 3 % faster at 1M (profiles/r06_p_pair32_ab.txt):
        # the look-ups are not what holds the 3-bit kernel, and one workgroup per CU loses the overlap of two.
        self.score_f32_pair = bits == 3 and os.environ.get("KVQ_SCORE_F32_PAIR", "0") == "1"
        self.score_f16_pair = bits == 3 and os.environ.get("KVQ_SCORE_F16", "0") == "1"
        if self.score_f16_pair and not getattr(QuantK, "_f16_pair_logged", False):
            QuantK._f16_pair_logged = True
            print("kvquant_amd: KVQ_SCORE_F16=1 -- 3-bit q.K^T through the opt-in fp16 pair-sum tables: attention outputs "
                  "within 4e-3 of the reference's instead of 1e-3", file=sys.stderr)
        self._reset_csr(dev)

    @property
    def device(self):
        return self.kcache.device

    def _reset_csr(self, dev):
        # uncapped-outlier CSR state (ML:402-408)
        self.rows = torch.tensor([], device=dev)
        self.cols = torch.tensor([], device=dev)
        self.vals = torch.tensor([], device=dev)
        self.start_rows = torch.tensor([], device=dev)
        self.num_threads = -1
        self.num_nonzeros = 0

    def reset(self):
        """ML:416-434 (plain zero fill instead of a masked assignment)."""
        self.klen = 0
        self.kcache.zero_()
        if self.include_sparse:
            for t in (self.outliers, self.outlier_indices, self.outliers_t, self.outlier_indices_t):
                if t is not None:
                    t.zero_()
            self._reset_csr(self.device)

    def load_lookup_table(self, quantizer, include_sparse=True, sparsity_threshold=0.99, norm=False):
        """ML:437-501.  thresholds -> fp16; offset / range formed in fp16;
        LUT row = sorted(centroids) * float(range_c) + float(offset_c) in the
        centroid dtype, stored fp32 [H, hd, 2^bits]."""
        dev = self.device
        up16 = torch.as_tensor(quantizer[0]).to(dev).half().flatten()
        lo16 = torch.as_tensor(quantizer[1]).to(dev).half().flatten()
        lut = torch.as_tensor(quantizer[2][0]).squeeze(-1).to(dev)
        lut, _ = lut.sort()
        self.lut = lut
        self.include_sparse = include_sparse
        self.sparsity_threshold = sparsity_threshold
        offset = (up16 + lo16) / 2
        rangeval = (up16 - lo16) / 2
        sf = rangeval.to(lut.dtype)
        off = offset.to(lut.dtype)
        n = 2 ** self.bits
        table = lut.unsqueeze(0) * sf.unsqueeze(1) + off.unsqueeze(1)
        self.lookup_table = table.float().reshape(self.num_heads, self.head_dim, n).contiguous()
        self.norm = norm
        if norm:
            self.normscale = torch.as_tensor(quantizer[3]).to(dev)
            self.normoffset = torch.as_tensor(quantizer[4]).to(dev)
            t2 = (lut * self.normscale + self.normoffset).unsqueeze(0) * sf.unsqueeze(1) + off.unsqueeze(1)
            self.lookup_table2 = t2.float().reshape(self.num_heads, self.head_dim, n).contiguous()
        else:
            self.normscale = self.normoffset = None
            self.lookup_table2 = None
        self.zeropoint = ((up16 + lo16) / 2).float()
        # the two codebook values the outlier residuals refer to (ML:729-733), contiguous per channel
        t_off = self.lookup_table2 if norm else self.lookup_table
        self.lut_ends = torch.stack((t_off[..., 0], t_off[..., -1]), dim=-1).reshape(-1, 2).contiguous()
        self.outlier_threshold_upper = up16.float().contiguous()
        self.outlier_threshold_lower = lo16.float().contiguous()
        self._step_layer = None          # (cached kvq_layer of decode_kv: the tables just moved)

    # -- host glue kept on the GPU ------------------------------------------------
    def _outlier_rows(self, k_tok_major, resc_tok_major):
        """ML:706-751 / 928-972 for [S, C] inputs; returns (vals, idx) [S, 2*thr]."""
        n = 2 ** self.bits
        thr = _threshold_k(self.sparsity_threshold, self.hidden_size)
        lut = (self.lookup_table2 if self.norm else self.lookup_table).reshape(-1, n)
        up_tmp, up_idx = torch.topk(resc_tok_major, thr, dim=-1)
        lo_tmp, lo_idx = torch.topk(resc_tok_major, thr, dim=-1, largest=False)
        up = torch.gather(k_tok_major, 1, up_idx) - lut[up_idx, n - 1]
        lo = torch.gather(k_tok_major, 1, lo_idx) - lut[lo_idx, 0]
        zeros = torch.cat((up_tmp <= 1, lo_tmp >= -1), dim=-1)
        vals = torch.cat((up, lo), dim=-1)
        idx = torch.cat((up_idx, lo_idx), dim=-1)
        idx, order = idx.sort(dim=-1)
        vals = torch.gather(vals, 1, order)
        zeros = torch.gather(zeros, 1, order)
        vals = vals.masked_fill(zeros, 0.0)
        return vals, idx.int()

    def append_and_score(self, q, k):
        """The GPU-resident core of forward_fused_sparse: q f32 [q_len, H, hd] (post-RoPE),
        k f32 [C] (pre-RoPE).  Appends k (pack + outlier row, one launch) and returns the raw
        scores f32 [q_len, H, L] (one launch).  No host synchronisation."""
        if self.compact:
            raise NotImplementedError("compact caches are read by decode_kv only (the reference-format kernels need the rows)")
        pos = self.klen - self.first_few_fp16
        if self.include_sparse:
            lut_off = self.lookup_table2 if self.norm else self.lookup_table
            ops.append_k_fused(self.bits, self.kcache, self.lookup_table, lut_off, k,
                               self.outlier_threshold_lower, self.outlier_threshold_upper, self.outliers,
                               self.outlier_indices, self.num_outliers // 2, pos, self.outliers_t,
                               self.outlier_indices_t)
        else:
            ops.append_k(self.bits, self.kcache, self.lookup_table, k, pos)
        self.klen += 1
        L = self.klen - self.first_few_fp16
        mul = torch.empty((q.shape[0], q.shape[1], L), dtype=torch.float32, device=q.device)
        table = self.lookup_table2 if (self.norm and self.bits == 2) else self.lookup_table  # ML:811-815
        if self.include_sparse:
            ops.score_k(self.bits, q, self.kcache, mul, table, L, self.rope_theta, self.first_few_fp16,
                        self.outliers, self.outlier_indices, accumulate=False)
        else:
            ops.score_k(self.bits, q, self.kcache, mul, table, L, self.rope_theta, self.first_few_fp16,
                        accumulate=False)
        return mul

    def forward_fused_sparse(self, q, k):
        """ML:651-876.  q: [H, q_len, hd] post-RoPE query; k: the new pre-RoPE key
        (C values).  Appends k and returns half [H, q_len, L] raw scores."""
        k = k.flatten().float().contiguous()
        q = q.float().transpose(0, 1).contiguous()
        mul = self.append_and_score(q, k)
        return mul.transpose(0, 1).contiguous().half()

    def parallel_pack(self, k, fused=True):
        """ML:879-972.  k: [H, hd, S] pre-RoPE keys of the prompt; the cache must
        be empty (as in the reference, which writes columns 0..S-1).  fused=False keeps the reference's
        structure (pack kernel + torch top-k / gather / sort on the GPU)."""
        if not self.include_sparse:
            raise AssertionError("parallel_pack needs include_sparse (as the reference, ML:975)")
        k = k.float().contiguous()
        S = k.shape[-1]
        col0 = _pack_col0(self.klen, self.first_few_fp16)
        if self.compact and not fused:
            raise NotImplementedError("compact caches take the fused prefill pack only")
        if fused:
            # one launch: pack + exact top-k selection + outlier rows (+ mirror), one workgroup per token
            lut_off = self.lookup_table2 if self.norm else self.lookup_table
            ops.pack_k_fused(self.bits, self.kcache, self.lookup_table, lut_off, k, self.outlier_threshold_lower,
                             self.outlier_threshold_upper, self.outliers, self.outlier_indices,
                             self.num_outliers // 2, col0, self.outliers_t, self.outlier_indices_t)
            self.klen += S        # (only once the library call has succeeded)
            return
        resc = torch.empty_like(k)
        ops.pack_k_sparse_parallel(self.bits, self.kcache, self.lookup_table, k, resc,
                                   self.outlier_threshold_lower, self.outlier_threshold_upper, col0)
        resc_t = resc.reshape(-1, S).t().contiguous()
        k_t = k.reshape(-1, S).t().contiguous()
        vals, idx = self._outlier_rows(k_t, resc_t)
        self.outliers[col0:col0 + S] = vals
        self.outlier_indices[col0:col0 + S] = idx
        self.outliers_t[:, col0:col0 + S] = vals.t()
        self.outlier_indices_t[:, col0:col0 + S] = idx.t()
        self.klen += S

    # ---- uncapped outliers (use_orig_sparse=True), 4 bit only ---------------------------------------
    def forward_fused_sparse_orig(self, q, k):
        """ML:504-559: outliers = everything outside the thresholds (code 7, residual x - zeropoint) kept
        in a growing CSR matrix; dense scores + CSR SpMV."""
        assert self.include_sparse and self.bits == 4
        from . import quant_cuda as qc
        k = k.flatten().float().contiguous()
        q = q.float().transpose(0, 1).contiguous()
        pos = self.klen - self.first_few_fp16
        self.rows, self.cols, self.vals, self.start_rows, nt, _ = qc.vecquant4appendvecKsparseorig(
            self.kcache, self.lookup_table, k, self.zeropoint, self.rows, self.cols, self.vals, self.start_rows,
            self.outlier_threshold_lower, self.outlier_threshold_upper, pos)
        self.num_threads = int(nt[0])
        self.num_nonzeros = self.vals.shape[0]
        self.klen += 1
        L = self.klen - self.first_few_fp16
        mul = torch.zeros((q.shape[0], q.shape[1], L), dtype=torch.float32, device=q.device)
        qc.vecquant4matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2_orig(
            q, self.kcache, mul, self.lookup_table, L, self.rows, self.cols, self.start_rows, self.vals, L,
            self.num_threads, self.num_nonzeros, self.rope_theta, self.first_few_fp16)
        return mul.transpose(0, 1).contiguous().half()

    def parallel_pack_orig(self, k):
        """ML:563-649.  NOTE (reference quirk, replicated): prefill packs with the CAPPED kernel (outliers
        saturate to the end codes) and stores residuals to the END codebook values, while decode forces
        code 7 and stores x - zeropoint."""
        assert self.include_sparse and self.bits == 4
        k = k.float().contiguous()
        S = k.shape[-1]
        resc = torch.empty_like(k)
        ops.pack_k_sparse_parallel(4, self.kcache, self.lookup_table, k, resc, self.outlier_threshold_lower,
                                   self.outlier_threshold_upper, 0)
        self.klen += S
        k = k.reshape(-1, S).clone()
        lower = k < self.outlier_threshold_lower.unsqueeze(-1)
        above = k > self.outlier_threshold_upper.unsqueeze(-1)
        lut = self.lookup_table.reshape(-1, 16)
        k1 = k - lut[:, 0].unsqueeze(-1)
        k2 = k - lut[:, 15].unsqueeze(-1)
        k[lower] = k1[lower]
        k[above] = k2[above]
        k[~torch.logical_or(above, lower)] = 0
        csr = k.t().contiguous().to_sparse_csr()
        self.rows = csr.crow_indices().int()
        self.cols = csr.col_indices().int()
        self.vals = csr.values().float()
        if len(self.vals) > 0:
            self.num_threads = int((self.vals.shape[0] + 9) / 10)
            self.num_nonzeros = self.vals.shape[0]
            nt_pad = int((self.num_threads + 127) / 128) * 128
            # start row of every 10-nnz thread (ML:638-646), vectorised: first row whose end exceeds j*10
            ends = self.rows[1:].long()
            j10 = torch.arange(nt_pad, device=k.device) * 10
            start = torch.searchsorted(ends, j10, right=True).int()
            start[j10 >= ends[-1]] = -1
            self.start_rows = start

class QuantV(nn.Module):
    """Compressed value cache with per-token codebooks (ML:978-1385)."""

    def __init__(self, bits=2, hidden_size=4096, num_heads=32, max_position_embeddings=-1,
                 include_sparse=False, sparsity_threshold=0.99, first_few_fp16=0, device=None, compact=False,
                 outlier_width=None):
        super().__init__()
        if bits not in (2, 3, 4):
            raise ValueError("bits must be 2, 3 or 4")
        if compact and not include_sparse:
            raise ValueError("compact is a format of the Dense-and-Sparse cache (include_sparse)")
        self.compact = bool(compact)
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.bits = bits
        self.lut = None
        self.zeropoint = None
        self.sparsity_threshold = sparsity_threshold
        self.include_sparse = include_sparse
        self.max_len = max_position_embeddings
        dev = _default_device(device)
        self.lookup_table = torch.zeros((self.max_len, 2 ** bits), dtype=torch.float32, device=dev)
        self.vcache = torch.zeros((num_heads, (self.head_dim // 32) * bits, self.max_len), dtype=torch.int32,
                                  device=dev)
        self.vlen = 0
        self.num_outliers = int(outlier_width) if outlier_width else 2 * _threshold_k(sparsity_threshold, hidden_size)
        if include_sparse and self.compact:
            # compact=True (opt-in, see QuantK): rows of packed entries fp16 residual << 16 | channel in outlier_indices,
            # no value array: 168 B per token instead of 336
            self.outliers = None
            self.outlier_indices = torch.zeros((self.max_len, self.num_outliers), dtype=torch.int32, device=dev)
        elif include_sparse:
            self.outliers = torch.zeros((self.max_len, self.num_outliers), dtype=torch.float32, device=dev)
            self.outlier_indices = torch.zeros((self.max_len, self.num_outliers), dtype=torch.int32, device=dev)
        self.first_few_fp16 = first_few_fp16
        self.norm = False
        self.lookup_table2 = None
        # The reference's kernel clips a value to the zero-point code only STRICTLY outside its per-token thresholds
        # (KCU:2084) while its glue stores the 21 largest / smallest as residuals (ML:1093-1096): a selected value that
        # EQUALS the threshold (21st == 22nd largest, ~10 % of fp16 tokens) keeps its own code AND gets a residual, i.e.
        # is dequantised twice.  Default True: the GPU-resident appends pack exactly the codes of the
        # reference's CUDA path on every token, ties included (tests/test_ties_gpu.py runs them next to the reference's
        # own kernel).  False: clip exactly the stored values -- the semantics of the reference's SIMULATED path
        # (SQ:95-108), whose reconstruction of such a value is exact; differs from the CUDA path in one code per tie.
        self.reference_tie_quirk = True

    @property
    def device(self):
        return self.vcache.device

    def reset(self):
        """ML:1028-1042."""
        self.vlen = 0
        self.lookup_table.zero_()
        self.vcache.zero_()
        if self.include_sparse:
            if self.outliers is not None:
                self.outliers.zero_()
            self.outlier_indices.zero_()
        if self.lookup_table2 is not None:
            self.lookup_table2.zero_()

    def load_lookup_table(self, quantizer, include_sparse=True, sparsity_threshold=0.99, norm=False):
        """ML:1045-1066: only the sorted global centroids are kept; per-token
        rows are built at append time."""
        lut = torch.as_tensor(quantizer[2][0]).squeeze(-1).to(self.device).float()
        self.lut, _ = lut.sort()
        self.include_sparse = include_sparse
        self.sparsity_threshold = sparsity_threshold
        self.norm = norm
        if norm:
            self.normscale = torch.as_tensor(quantizer[3]).to(self.device)
            self.normoffset = torch.as_tensor(quantizer[4]).to(self.device)
            # the same two numbers as fp32 host scalars for the fused kernels (one sync, at load time only)
            self._ns = float(self.normscale.float())
            self._no = float(self.normoffset.float())
            self.lookup_table2 = torch.zeros((self.max_len, 2 ** self.bits), dtype=torch.float32,
                                             device=self.device)
        else:
            self.normscale = self.normoffset = None
            self.lookup_table2 = None
        self._tables_version = getattr(self, "_tables_version", 0) + 1

    def topk_inputs(self, v_tok_major):
        """The four tensors the reference's attention hands over (ML:1537-1545 on the
        CPU, 1552-1557 on the GPU): topk(thr+1) of each token's values, both sides.
        Here always on the GPU."""
        k = _threshold_k(self.sparsity_threshold, self.hidden_size) + 1
        uv, ui = torch.topk(v_tok_major, k, dim=-1)
        lv, li = torch.topk(v_tok_major, k, dim=-1, largest=False)
        return uv, ui, lv, li

    def vnorm_args(self, prefill=False):
        """Q-Norm argument of the fused V appends: second per-token row table + (scale, offset) + which table the
        sparse residuals refer to -- ALWAYS the one the p.V kernel dequantises with (mix_table): the Q-Norm row at
        2 bit (ML:1153-1156, 1237-1240), the plain row at 3 / 4 bit, in decode and prefill alike.  (The reference's
        prefill glue reads the zero points from lookup_table2 at every width, ML:1369-1375 -- rows it never writes;
        taking them from a table the kernel does not read would reconstruct every prompt outlier as
        x + lut1[z] - lut2[z].)  None without Q-Norm."""
        if not self.norm:
            return None if self.reference_tie_quirk else (None, 1.0, 0.0, False, False)     # (NULL options = the reference's quirk)
        return (self.lookup_table2, self._ns, self._no, self.bits == 2, self.reference_tie_quirk)

    def mix_table(self):
        """table the p.V kernel dequantises with: the Q-Norm rows at 2 bit (ML:1237-1240)"""
        return self.lookup_table2 if (self.norm and self.bits == 2) else self.lookup_table

    def append_and_mix(self, p, v):
        """The GPU-resident core of forward_fused_sparse: p f32 [q_len, H, L+1] probabilities,
        v f32 [C].  Appends v (thresholds, codebook row(s), pack, outlier row: one launch) and returns
        f32 [q_len, H, hd] (two launches).  No host synchronisation.  include_sparse only."""
        pos = self.vlen - self.first_few_fp16
        ops.append_v_fused(self.bits, self.vcache, self.lookup_table, self.lut, v, self.outliers,
                           self.outlier_indices, self.num_outliers // 2, pos, self.vnorm_args())
        self.vlen += 1
        L = self.vlen - self.first_few_fp16
        mul = torch.empty((p.shape[0], p.shape[1], self.head_dim), dtype=torch.float32, device=p.device)
        ops.mix_v(self.bits, p, self.vcache, mul, self.mix_table(), L, self.outliers, self.outlier_indices,
                  accumulate=False)
        return mul

    def forward_fused_sparse(self, score, v, upper_outlier_vals=None, upper_outlier_indices=None,
                             lower_outlier_vals=None, lower_outlier_indices=None):
        """ML:1069-1291.  score: [H, q_len, L+1] probabilities; v: the new value
        (C values).  With include_sparse the four top-(thr+1) tensors may be passed as
        in the reference (they are then used as given) or left None: selection then
        happens inside the fused GPU append."""
        if self.compact:
            raise NotImplementedError("compact caches are read by decode_kv only")
        score = score.float()
        v_in = v.flatten()
        v = v_in.float().contiguous()
        if self.include_sparse and upper_outlier_vals is None:
            mul = self.append_and_mix(score.transpose(0, 1).contiguous(), v)
            return mul.transpose(0, 1).contiguous().half()
        pos = self.vlen - self.first_few_fp16
        zc = ZERO_CODE[self.bits]
        if self.include_sparse:
            if upper_outlier_vals is None:
                uv, ui, lv, li = self.topk_inputs(v.unsqueeze(0))
                upper_outlier_vals, upper_outlier_indices = uv[0], ui[0]
                lower_outlier_vals, lower_outlier_indices = lv[0], li[0]
            dev = v.device
            upper_outlier_vals = upper_outlier_vals.to(dev)
            lower_outlier_vals = lower_outlier_vals.to(dev)
            maxval = upper_outlier_vals[-1]
            minval = lower_outlier_vals[-1]
            uv, lv = upper_outlier_vals[:-1], lower_outlier_vals[:-1]
            ui, li = upper_outlier_indices[:-1].to(dev), lower_outlier_indices[:-1].to(dev)
            offset = (maxval + minval) / 2
            sf = (maxval - minval) / 2
        else:
            # compute_lut (ML:318-349) on the un-widened vector: fp16 arithmetic for an fp16 v
            maxval, minval = v_in.max(), v_in.min()
            offset = (maxval + minval) / 2
            sf = (maxval - minval) / 2
        # ML:1113: lut * sf.item() + offset.item() in fp32; sf / offset / thresholds stay on the GPU as 0-d
        # tensors (the reference synchronises on .item() here)
        row = self.lut.float() * sf.float() + offset.float()
        self.lookup_table[pos] = row
        if self.norm:
            self.lookup_table2[pos] = (self.lut.float() * self.normscale + self.normoffset) * sf.float() + offset.float()
        score = score.transpose(0, 1).contiguous()
        if self.include_sparse:
            zp_row = self.lookup_table2[pos] if (self.norm and self.bits == 2) else row   # ML:1153-1156
            zeropoint = zp_row[zc]
            # vecquant{b}appendvecVsparse with the two thresholds read from device memory: the one-token case of the
            # parallel pack entry point (per-token threshold arrays), so nothing is copied to the host
            ops.pack_v_sparse_parallel(self.bits, self.vcache, self.lookup_table, v.view(self.num_heads, self.head_dim, 1),
                                       minval.float().reshape(1), maxval.float().reshape(1), pos)
            vals = torch.cat((uv, lv), dim=-1) - zeropoint
            idx = torch.cat((ui, li), dim=-1)
            idx, order = idx.sort()
            self.outliers[pos] = vals[order]
            self.outlier_indices[pos] = idx.int()
        else:
            ops.append_v(self.bits, self.vcache, self.lookup_table, v, pos)
        self.vlen += 1
        L = self.vlen - self.first_few_fp16
        mul = torch.empty((score.shape[0], score.shape[1], self.head_dim), dtype=torch.float32,
                          device=score.device)
        table = self.lookup_table2 if (self.norm and self.bits == 2) else self.lookup_table      # ML:1237-1240
        if self.include_sparse:
            ops.mix_v(self.bits, score, self.vcache, mul, table, L, self.outliers, self.outlier_indices,
                      accumulate=False)
        else:
            ops.mix_v(self.bits, score, self.vcache, mul, table, L, accumulate=False)
        return mul.transpose(0, 1).contiguous().half()

    def parallel_pack(self, v, upper_outlier_vals=None, upper_outlier_indices=None,
                      lower_outlier_vals=None, lower_outlier_indices=None):
        """ML:1294-1382.  v: [H, hd, S]; top-k tensors [S, thr+1] or None (computed
        here on the GPU)."""
        if not self.include_sparse:
            raise AssertionError("parallel_pack needs include_sparse (as the reference, ML:1322)")
        v = v.float().contiguous()
        S = v.shape[-1]
        col0 = _pack_col0(self.vlen, self.first_few_fp16)
        if self.compact and upper_outlier_vals is not None:
            raise NotImplementedError("compact caches take the fused prefill pack only")
        if upper_outlier_vals is None:
            # one launch: top-(k+1) selection, per-token codebook rows, pack, outlier rows
            ops.pack_v_fused(self.bits, self.vcache, self.lookup_table, self.lut, v, self.outliers,
                             self.outlier_indices, self.num_outliers // 2, col0, self.vnorm_args(prefill=True))
            self.vlen += S
            return
        if upper_outlier_vals is None:
            vt = v.reshape(-1, S).t().contiguous()
            upper_outlier_vals, upper_outlier_indices, lower_outlier_vals, lower_outlier_indices = \
                self.topk_inputs(vt)
        maxval = upper_outlier_vals[:, -1].float().contiguous()
        minval = lower_outlier_vals[:, -1].float().contiguous()
        uv, lv = upper_outlier_vals[:, :-1], lower_outlier_vals[:, :-1]
        ui, li = upper_outlier_indices[:, :-1], lower_outlier_indices[:, :-1]
        offset = (maxval + minval) / 2
        sf = (maxval - minval) / 2
        rows = self.lut.float().unsqueeze(0) * sf.unsqueeze(-1) + offset.unsqueeze(-1)
        self.lookup_table[col0:col0 + S] = rows
        if self.norm:
            # (the reference's prefill glue reads lookup_table2[:, zero code] for the residuals, ML:1369-1375, but
            # never fills those rows -- they are still zero from the constructor; the rows are written here, which
            # is what its decode path does for every later token, ML:1116-1118)
            self.lookup_table2[col0:col0 + S] = (self.lut.float() * self.normscale + self.normoffset).unsqueeze(0) \
                * sf.unsqueeze(-1) + offset.unsqueeze(-1)
        ops.pack_v_sparse_parallel(self.bits, self.vcache, self.lookup_table, v, minval, maxval, col0)
        zc = ZERO_CODE[self.bits]
        zp_src = self.mix_table()        # the table the p.V kernel reads (see vnorm_args)
        vals = torch.cat((uv, lv), dim=-1) - zp_src[col0:col0 + S, zc].unsqueeze(-1)
        idx = torch.cat((ui, li), dim=-1)
        idx, order = idx.sort(dim=-1)
        vals = torch.gather(vals, 1, order)
        self.outliers[col0:col0 + S] = vals
        self.outlier_indices[col0:col0 + S] = idx.int()
        self.vlen += S


ONE_CALL_PER_LAYER = os.environ.get("KVQ_DECODE_MULTICALL", "0") != "1"   # (env: A/B runs against the five-call path)
FUSE_SOFTMAX_INTO_MIX_V = False
# The p.V kernel normalises the raw scores itself (kvq_mix_v_softmax: one launch and one pass over the probabilities
# less).  Round 2 this lost beyond 32K tokens -- every probability was converted twice, by the workgroup that streams its
# head's rows and by the one that owned its token's outliers for ALL heads; since a workgroup takes the outliers of its
# own heads only, it wins at every length (profiles/r03_b_fused_softmax.txt: 128K 6.12 -> 5.98 ms/step).  Env: A/B runs.
FUSE_SOFTMAX_UP_TO = int(os.environ.get("KVQ_FUSE_SOFTMAX_UP_TO", str(1 << 62)))
# One kernel for q.K^T + softmax + p.V per 256-token tile and a merge (kvq_fused_decode.hip: kvq_fused_attend) instead of
# the score / p.V kernel pair.  Measured (profiles/r04_fused_decode.txt): slower at 32K (one tile per workgroup leaves
# half the chip idle), equal at 128K, 3 % slower at 256K, 3.7 % FASTER at 1M (several generations of workgroups: K phases
# overlap V phases; same box at 1M: nuq4 11.08 -> 10.72 ms/step, but nuq3 + sinks 10.87 -> 11.02 and nuq2 9.28 -> 9.81)
# -> rounds 4 - 5: the default at 4 bit from 512K cached tokens on.  Round 6: the kernel PAIR wins at every length again --
# its p.V is the one-workgroup-per-CU kernel with the outlier entries inside the loop (kvq_mix_v_wide.hip); same box,
# profiles/r06_k_misc.txt: 1M 10.70 -> 9.86 ms/step (8 layers), 512K 10.86 -> 9.98, 256K 5.71 -> 5.22 -- so the fused kernel is
# opt-in only: KVQ_FUSED_ATTEND=1 forces it on (0: off), KVQ_FUSED_ATTEND_FROM=<tokens> turns it on from a length.
FUSED_ATTEND = {"1": True, "0": False}.get(os.environ.get("KVQ_FUSED_ATTEND", ""), None)
FUSED_ATTEND_FROM = int(os.environ.get("KVQ_FUSED_ATTEND_FROM", str(1 << 62)))


def decode_kv(kc, vc, q, k, v, sink_scores=None, k_sink=None, v_sink=None):
    """One decode token through a layer's compressed KV path, GPU-resident, 5 launches:
    prologue (K append | V append | K tables) -> q.K^T (+ first softmax pass) -> softmax finish -> p.V -> slab
    reduce; by default 4 launches (+ a 4 us merge of the softmax partials beyond 64K tokens): the p.V kernel normalises
    the raw scores itself (kvq_mix_v_softmax, FUSE_SOFTMAX_UP_TO above).
    q: [H, hd] RoPE'd query, k, v: [C] pre-RoPE key / value, all fp16 or all fp32 (no conversion
    launches).  sink_scores: optional f16 [H, n_sink] already scaled scores of the fp16 sink tokens.
    k_sink (f16 [H, 128, n_sink], post-RoPE) / v_sink (f16 [H, n_sink, 128]) instead: the fp16 sink caches themselves --
    their scores are computed in the prologue launch and their share of the output in the softmax launch, so that
    `out` is the complete attention output (the reference's two fp16 matmuls, division and add: ML:1950-1962,
    1987-1995; five small launches per layer otherwise).
    Returns (out f32 [1, H, hd], sink_probs f16 [H, n_sink] or None).  Sparse (include_sparse) caches only."""
    if not (kc.include_sparse and vc.include_sparse):
        raise ValueError("decode_kv needs include_sparse caches")
    bits = kc.bits
    kpos = kc.klen - kc.first_few_fp16
    vpos = vc.vlen - vc.first_few_fp16
    lut_off = kc.lookup_table2 if kc.norm else kc.lookup_table
    # the score tables are built from the table the reference dequantises with: the Q-Norm one at 2 bit (ML:811-815)
    table = kc.lookup_table2 if (kc.norm and bits == 2) else kc.lookup_table
    inv = 1.0 / (kc.head_dim ** 0.5)
    sinks = None
    if k_sink is not None:
        if sink_scores is not None or v_sink is None:
            raise ValueError("pass either sink_scores or (k_sink, v_sink)")
        sink_scores = torch.empty((kc.num_heads, k_sink.shape[2]), dtype=torch.float16, device=kc.device)
        sinks = (k_sink, sink_scores, inv)
    H = kc.num_heads
    compact = getattr(kc, "compact", False) or getattr(vc, "compact", False)
    if compact and not (kpos == vpos and (sink_scores is None or sinks is not None)):
        raise ValueError("compact caches: decode_kv with k_sink / v_sink (or no sinks), K and V in step")
    if (ONE_CALL_PER_LAYER or compact) and kpos == vpos and (sink_scores is None or sinks is not None):
        # the whole launch sequence from one library call (kvq_decode_step): the five Python / ctypes round trips of
        # the path below cost ~78 us of host time per layer, as much as the GPU needs for a 4K-token cache
        key = (id(vc), getattr(vc, "_tables_version", 0), vc.reference_tie_quirk, vc.norm, kc.norm, kc.compact, vc.compact,
               kc.score_f16_pair, getattr(kc, "score_f32_pair", False))
        cached = getattr(kc, "_step_layer", None)
        if cached is None or cached[0] != key:
            cached = (key,) + ops.make_layer(kc, vc, table, lut_off)
            kc._step_layer = cached
        out = torch.empty((1, H, vc.head_dim), dtype=torch.float32, device=kc.device)
        sink_probs = None if sinks is None else torch.empty_like(sink_scores)
        L = kpos + 1
        fuse = FUSE_SOFTMAX_INTO_MIX_V or L <= FUSE_SOFTMAX_UP_TO
        fused = FUSED_ATTEND if FUSED_ATTEND is not None else (bits == 4 and L >= FUSED_ATTEND_FROM)
        mode = (3 if fused else 1) if fuse else 0
        ops.decode_step(cached[1], kpos, q, k, v, out, mode, sinks, v_sink, sink_probs)
        kc.klen += 1
        vc.vlen += 1
        return out, sink_probs
    ws = ops.decode_prologue(bits, kc.kcache, kc.lookup_table, lut_off, k, kc.outlier_threshold_lower,
                             kc.outlier_threshold_upper, kc.outliers, kc.outlier_indices, kpos, vc.vcache,
                             vc.lookup_table, vc.lut, v, vc.outliers, vc.outlier_indices, vpos, q,
                             kc.num_outliers // 2, kc.outliers_t, kc.outlier_indices_t, kc.lut_ends,
                             None if table is kc.lookup_table else table, vc.vnorm_args(), sinks)
    kc.klen += 1
    vc.vlen += 1
    L = kc.klen - kc.first_few_fp16
    scores = torch.empty((1, H, L), dtype=torch.float32, device=kc.device)
    out = torch.empty((1, H, vc.head_dim), dtype=torch.float32, device=kc.device)
    if FUSE_SOFTMAX_INTO_MIX_V or L <= FUSE_SOFTMAX_UP_TO:
        sink_probs = ops.score_k_mix_v(bits, kc.kcache, scores, table, L, kc.rope_theta, kc.first_few_fp16, ws,
                                       kc.outliers, kc.outlier_indices, inv, vc.vcache, out,
                                       vc.mix_table(), vc.outliers, vc.outlier_indices, sink_scores, kc.outliers_t,
                                       kc.outlier_indices_t, v_sink, f16_pair=kc.bits == 3 and kc.score_f16_pair)
        return out, sink_probs
    probs, sink_probs = ops.score_k_softmax(bits, kc.kcache, scores, table, L, kc.rope_theta, kc.first_few_fp16, ws,
                                            kc.outliers, kc.outlier_indices, inv, sink_scores,
                                            kc.outliers_t, kc.outlier_indices_t, v_sink, out)
    ops.mix_v(bits, probs.unsqueeze(0), vc.vcache, out, vc.mix_table(), L, vc.outliers, vc.outlier_indices,
              accumulate=v_sink is not None)
    return out, sink_probs


def shard_record_floats(H, hd):
    """floats of one shard's record: [H*hd: output normalised over the shard][H x (max, normaliser)]"""
    return H * hd + 2 * H


def shard_attention(kc, vc, q, k=None, v=None, pos_base=0, record=None, k_sink=None, v_sink=None):
    """Attention of one decode token over ONE SHARD of a context that is split along the token axis (SURVEY 8e asks
    for layer / head placement; this is the third cut, the one that speeds up a single long stream: every GPU holds
    L / N cached tokens of every layer and streams only those).  kc / vc hold the shard's tokens, whose positions
    start at `pos_base`; k, v (the new token) are appended to THIS shard when given -- the owner of the newest tokens
    -- and the other shards only score.  Everything stays in library launches (no torch arithmetic): table prep or
    prologue -> q.K^T + softmax partials -> softmax finish -> softmax statistics -> p.V -> slab reduce.
    k_sink (f16 [H, 128, n_sink], post-RoPE) / v_sink (f16 [H, n_sink, 128]): the fp16 attention-sink tokens
    (first_few_fp16, ML:1464-1466), handed to the shard that holds the START of the context (and only to it): their
    scores enter that shard's softmax and its (max, normaliser), their values its output -- the merge across shards is
    unchanged.  pos_base of that shard = the number of sink tokens (the position of its first compressed token).
    Returns (out f32 [1, H, hd] normalised over the shard, M f32 [H], Z f32 [H]) -- views of `record` (f32
    [shard_record_floats], allocated when None), the buffer that goes into the all-gather: the shard's softmax maximum
    and normaliser of the scaled fp16 scores, so that the shards merge exactly (flash-decoding across devices; the
    probabilities are rounded to fp16 per shard instead of over the whole row, a last-bit effect).  An EMPTY shard
    (no cached tokens, no new one, no sinks) returns zeros with (M, Z) = (-inf, 0).  Sparse caches with the outlier mirror."""
    if not (kc.include_sparse and vc.include_sparse):
        raise ValueError("shard_attention needs include_sparse caches")
    if (k_sink is None) != (v_sink is None):
        raise ValueError("shard_attention: k_sink and v_sink come together")
    if getattr(kc, "compact", False) or getattr(vc, "compact", False):
        raise NotImplementedError("shard_attention reads the reference outlier format; compact caches go through decode_kv")
    bits, H, hd = kc.bits, kc.num_heads, vc.head_dim
    nf = kc.first_few_fp16                     # (sink tokens counted in klen, as in the reference's attention)
    if kc.klen < nf or kc.lookup_table is None:
        raise ValueError("shard_attention: the cache counts %d sink tokens in klen = %d / has no tables loaded" % (nf, kc.klen))
    # a cache that counts sink tokens holds them OUTSIDE the compressed rows: without their fp16 keys / values they would
    # silently drop out of the softmax and of the output (HeadShard.attend makes the same check).  The shards that do not
    # hold the start of the context use caches built with first_few_fp16 = 0.
    if nf > 0 and k_sink is None:
        raise ValueError("shard_attention: this cache counts %d fp16 sink tokens -- pass their k_sink / v_sink "
                         "(the shard holding the start of the context), or build the other shards' caches with "
                         "first_few_fp16 = 0" % nf)
    if nf > 0 and int(k_sink.shape[2]) != nf:
        raise ValueError("shard_attention: k_sink holds %d tokens, the cache counts %d sink tokens" % (int(k_sink.shape[2]), nf))
    if record is None:
        record = torch.empty(shard_record_floats(H, hd), dtype=torch.float32, device=kc.device)
    out = record[:H * hd].view(1, H, hd)
    stats = record[H * hd:].view(H, 2)
    inv = 1.0 / (kc.head_dim ** 0.5)
    table = kc.lookup_table2 if (kc.norm and bits == 2) else kc.lookup_table
    pos_offset = int(pos_base)
    if k is None and kc.klen - nf == 0 and k_sink is None:
        out.zero_()
        stats[:, 0] = float("-inf")
        stats[:, 1] = 0.0
        return out, stats[:, 0], stats[:, 1]
    sinks = sink_scores = None
    if k_sink is not None:
        sink_scores = torch.empty((H, k_sink.shape[2]), dtype=torch.float16, device=kc.device)
        sinks = (k_sink, sink_scores, inv)
    if k is not None:
        kpos = kc.klen - nf
        vpos = vc.vlen - nf
        lut_off = kc.lookup_table2 if kc.norm else kc.lookup_table
        ws = ops.decode_prologue(bits, kc.kcache, kc.lookup_table, lut_off, k, kc.outlier_threshold_lower,
                                 kc.outlier_threshold_upper, kc.outliers, kc.outlier_indices, kpos, vc.vcache,
                                 vc.lookup_table, vc.lut, v, vc.outliers, vc.outlier_indices, vpos, q,
                                 kc.num_outliers // 2, kc.outliers_t, kc.outlier_indices_t, kc.lut_ends,
                                 None if table is kc.lookup_table else table, vc.vnorm_args(), sinks)
        kc.klen += 1
        vc.vlen += 1
    else:
        ws = ops.score_k_tables(bits, q, table, H, sinks)
    L = kc.klen - nf
    if L == 0:
        # only the sink tokens: their softmax and output, no compressed token to score
        probs = torch.softmax(sink_scores.float(), dim=-1)
        M = sink_scores.float().max(dim=-1).values
        stats[:, 0] = M
        stats[:, 1] = torch.exp(sink_scores.float() - M[:, None]).sum(dim=-1)
        out.copy_(torch.matmul(probs.half().view(H, 1, -1), v_sink).float().view(1, H, hd))
        return out, stats[:, 0], stats[:, 1]
    scores = torch.empty((1, H, L), dtype=torch.float32, device=kc.device)
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    if n_parts <= 0:
        raise ValueError("shard_attention: unsupported shape")
    parts = ops.score_k_prepared_softmax(bits, kc.kcache, scores, table, L, kc.rope_theta, pos_offset, ws,
                                         kc.outliers, kc.outlier_indices, inv, n_parts, kc.outliers_t,
                                         kc.outlier_indices_t)
    ops.softmax_stats(parts, n_parts, H, stats, sink_scores)
    probs, _ = ops.softmax_finish(scores[0], parts, n_parts, inv, sink_scores, v_sink, out if v_sink is not None else None)
    ops.mix_v(bits, probs.unsqueeze(0), vc.vcache, out, vc.mix_table(), L, vc.outliers, vc.outlier_indices,
              accumulate=v_sink is not None)
    return out, stats[:, 0], stats[:, 1]



class HeadShard:
    """Heads [h0, h0 + n_heads) of one layer's compressed KV cache, all tokens: the HEAD-sharded placement of SURVEY 8e
    ("by head"; the reference itself places whole layers, ML:2428-2453).  Heads are independent until o_proj, so a rank
    computes the complete attention output of its heads and ONE all-gather of [n_heads * hd] floats per layer assembles
    the layer's (sharding.head_sharded_step).  What is NOT per head is the quantiser: a token's 21 + 21 outliers are
    selected over all H * hd channels (ML:742, 1093-1096) and V's codebook row comes from the token's 22nd largest /
    smallest value (ML:1086-1119).  Every rank therefore sees the WHOLE new token (in a tensor-parallel model: one
    all-gather of the k / v slices) and appends it into a full-width staging column with the ordinary kernels --
    bit-identical selection and codes on every rank -- and kvq_extract_heads moves its heads' words, the codebook row(s) and
    its share of the outlier entries (channels rebased, foreign entries zeroed) into the shard's own cache, which the
    ordinary matvec kernels then read with H = n_heads.  Outlier rows stay 42 wide: a shard cannot know in advance how
    many of a token's outliers fall into its heads.  A decode token is ONE library call (kvq_head_shard_step: append ->
    extract -> tables -> q.K^T -> softmax -> p.V).

    Carries everything BASELINE config 3 needs (round 5): `first_few_fp16` sink tokens (they shard trivially by head: the
    caller hands over its heads' slices of the fp16 sink caches) and Q-Norm quantizers (the normalised V codebook rows
    travel with the extract).

    `full_k` / `full_v`: the staging caches (full width, `stage_len` columns; they also own the quantiser tables),
    `k` / `v`: the shard (QuantK / QuantV over n_heads heads).  staging = another HeadShard: share ITS staging buffers
    (all layers of a model can stage through one set: a decode token uses column 0 only, a prompt goes through in pieces;
    the tables stay per layer).  Reference outlier format."""

    def __init__(self, bits, hidden_size, num_heads, heads, max_position_embeddings, sparsity_threshold=0.99,
                 rope_theta=10000, device=None, stage_len=64, first_few_fp16=0, staging=None):
        h0, n = int(heads[0]), int(heads[1])
        if not (0 <= h0 and n > 0 and h0 + n <= num_heads):
            raise ValueError("HeadShard: heads (%d, %d) outside 0..%d" % (h0, n, num_heads))
        self.bits, self.h0, self.n_heads, self.num_heads = bits, h0, n, num_heads
        self.head_dim = hidden_size // num_heads
        self.first_few_fp16 = int(first_few_fp16)
        kw = dict(bits=bits, include_sparse=True, sparsity_threshold=sparsity_threshold, device=device)
        own_len = 64 if staging is not None else stage_len
        self.full_k = QuantK(hidden_size=hidden_size, num_heads=num_heads, max_position_embeddings=own_len,
                             rope_theta=rope_theta, **kw)
        self.full_v = QuantV(hidden_size=hidden_size, num_heads=num_heads, max_position_embeddings=own_len, **kw)
        self._staging = staging
        if staging is not None:
            if (staging.bits, staging.num_heads, staging.head_dim) != (bits, num_heads, self.head_dim):
                raise ValueError("HeadShard: the shared staging caches have another shape")
            stage_len = staging.stage_len
            self._adopt_staging()
        width = self.full_k.num_outliers
        self.k = QuantK(hidden_size=n * self.head_dim, num_heads=n, max_position_embeddings=max_position_embeddings,
                        rope_theta=rope_theta, outlier_width=width, first_few_fp16=first_few_fp16, **kw)
        self.v = QuantV(hidden_size=n * self.head_dim, num_heads=n, max_position_embeddings=max_position_embeddings,
                        outlier_width=width, first_few_fp16=first_few_fp16, **kw)
        # (the fp16 sink tokens count in klen / vlen as in the reference's attention, ML:1877-1888; the caller owns the
        #  sink caches and hands its heads' slices to attend())
        self.k.klen += self.first_few_fp16
        self.v.vlen += self.first_few_fp16
        self.stage_len = stage_len
        self.device = self.k.device
        self._layers = None

    def _adopt_staging(self):
        """point the staging caches' DATA buffers at the shared set (the tables stay this layer's own)"""
        sk, sv = self._staging.full_k, self._staging.full_v
        for name in ("kcache", "outliers", "outlier_indices", "outliers_t", "outlier_indices_t"):
            setattr(self.full_k, name, getattr(sk, name))
        self.full_k.max_len = sk.max_len
        for name in ("vcache", "lookup_table", "outliers", "outlier_indices"):
            setattr(self.full_v, name, getattr(sv, name))
        self.full_v.max_len = sv.max_len

    @property
    def klen(self):
        return self.k.klen

    def load_lookup_table(self, k_quantizer, v_quantizer, include_sparse=True, sparsity_threshold=0.99, norm=False):
        """the layer's quantizers (deployment/llama.py:186-198) -> the staging caches; the shard reads its heads' slices"""
        if not include_sparse:
            raise ValueError("HeadShard is a Dense-and-Sparse cache")
        self.full_k.load_lookup_table(k_quantizer, include_sparse, sparsity_threshold, norm)
        self.full_v.load_lookup_table(v_quantizer, include_sparse, sparsity_threshold, norm)
        if self._staging is not None and norm:
            # (the Q-Norm rows of the staged tokens are data like the plain rows: one shared buffer)
            sv = self._staging.full_v
            if getattr(sv, "lookup_table2", None) is None or sv.lookup_table2.shape[0] != sv.max_len:
                sv.lookup_table2 = torch.zeros((sv.max_len, 2 ** self.bits), dtype=torch.float32, device=sv.device)
            self.full_v.lookup_table2 = sv.lookup_table2
        fk, k = self.full_k, self.k
        lo, hi = self.h0, self.h0 + self.n_heads
        c0, c1 = lo * self.head_dim, hi * self.head_dim
        k.lut, k.norm, k.include_sparse, k.sparsity_threshold = fk.lut, fk.norm, True, sparsity_threshold
        k.lookup_table = fk.lookup_table[lo:hi].contiguous()
        k.lookup_table2 = fk.lookup_table2[lo:hi].contiguous() if fk.lookup_table2 is not None else None
        k.lut_ends = fk.lut_ends[c0:c1].contiguous()
        k.zeropoint = fk.zeropoint.flatten()[c0:c1].contiguous()
        k.outlier_threshold_upper = fk.outlier_threshold_upper.flatten()[c0:c1].contiguous()
        k.outlier_threshold_lower = fk.outlier_threshold_lower.flatten()[c0:c1].contiguous()
        k.normscale, k.normoffset = fk.normscale, fk.normoffset
        k._step_layer = None
        # the shard's V cache keeps the sorted codebook (and, with Q-Norm, its own rows of the normalised table)
        self.v.load_lookup_table(v_quantizer, include_sparse, sparsity_threshold, norm)
        self._layers = None
        return self

    def reset(self):
        for c in (self.full_k, self.full_v, self.k, self.v):
            c.reset()
        self.k.klen += self.first_few_fp16
        self.v.vlen += self.first_few_fp16

    def _extract(self, n):
        ops.extract_heads(self.bits, self.h0, self.n_heads, self.full_k, self.full_v, self.k, self.v, 0,
                          self.k.klen - self.first_few_fp16, n)
        self.k.klen += n
        self.v.vlen += n

    def pack(self, k, v):
        """prompt tokens (behind the fp16 sink tokens, if any): k, v [H, hd, S] (whole tokens, pre-RoPE keys) -> the shard's
        columns, through the staging caches in pieces of stage_len tokens (QuantK / QuantV.parallel_pack: one launch each
        per piece)"""
        S = k.shape[-1]
        for s0 in range(0, S, self.stage_len):
            n = min(self.stage_len, S - s0)
            self.full_k.klen = 0
            self.full_v.vlen = 0
            self.full_k.parallel_pack(k[..., s0:s0 + n])
            self.full_v.parallel_pack(v[..., s0:s0 + n])
            self._extract(n)

    def _layer_structs(self):
        if self._layers is None:
            fk, fv, k, v = self.full_k, self.full_v, self.k, self.v
            f_off = fk.lookup_table2 if fk.norm else fk.lookup_table
            f_tab = fk.lookup_table2 if (fk.norm and self.bits == 2) else fk.lookup_table
            s_off = k.lookup_table2 if k.norm else k.lookup_table
            s_tab = k.lookup_table2 if (k.norm and self.bits == 2) else k.lookup_table
            self._layers = (ops.make_layer(fk, fv, f_tab, f_off), ops.make_layer(k, v, s_tab, s_off))
        return self._layers[0][0], self._layers[1][0]

    def attend(self, q, k, v, record=None, k_sink=None, v_sink=None):
        """one decode token: q [H, hd] post-RoPE query (or this shard's [n_heads, hd] slice), k / v [H * hd] the WHOLE new
        token (fp16 or fp32, the same dtype as q).  k_sink (f16 [n_heads, 128, n_sink], post-RoPE) / v_sink (f16 [n_heads,
        n_sink, 128]): this shard's heads of the fp16 sink caches (first_few_fp16 > 0).  Appends the token (staging column 0
        -> extract) and returns f32 [1, n_heads, hd]: the complete, normalised attention output of this shard's heads over
        all cached tokens (a view of `record` when given).  ONE library call (kvq_head_shard_step)."""
        if (k_sink is None) != (self.first_few_fp16 == 0) or (k_sink is None) != (v_sink is None):
            raise ValueError("HeadShard.attend: pass this shard's k_sink / v_sink exactly when first_few_fp16 > 0")
        if q.shape[0] == self.num_heads and self.n_heads != self.num_heads:
            q = q[self.h0:self.h0 + self.n_heads]
        q = q.contiguous()
        k, v = k.flatten().contiguous(), v.flatten().contiguous()
        if record is None:
            out = torch.empty((1, self.n_heads, self.head_dim), dtype=torch.float32, device=self.device)
        else:
            out = record[:self.n_heads * self.head_dim].view(1, self.n_heads, self.head_dim)
        full, shard = self._layer_structs()
        sinks = sink_probs = None
        if k_sink is not None:
            scores = torch.empty((self.n_heads, k_sink.shape[2]), dtype=torch.float16, device=self.device)
            sinks = (k_sink, scores, 1.0 / (self.head_dim ** 0.5))
            sink_probs = torch.empty_like(scores)
        col = self.k.klen - self.first_few_fp16
        ops.head_shard_step(full, shard, self.h0, col, q, k, v, out, 1, sinks, v_sink, sink_probs)
        self.k.klen += 1
        self.v.vlen += 1
        return out


def combine_shards(outs, Ms, Zs):
    """exact merge of per-shard attention: out = sum_r w_r out_r / sum_r w_r with w_r = Z_r exp(M_r - max_r M_r).
    outs [R, 1, H, hd], Ms / Zs [R, H] (stacked over the shards, any device); shards without tokens carry (-inf, 0).
    Plain torch (CPU tests, reference for kvq_combine_shards -- the GPU path merges the gathered records with that
    kernel, see sharding.token_sharded_step)."""
    Mg = Ms.max(dim=0).values
    w = torch.where(torch.isfinite(Ms), Zs * torch.exp(Ms - Mg[None, :]), torch.zeros_like(Zs))      # [R, H]
    den = w.sum(dim=0)
    return (outs * w[:, None, :, None]).sum(dim=0) / torch.where(den > 0, den, torch.ones_like(den))[None, :, None]

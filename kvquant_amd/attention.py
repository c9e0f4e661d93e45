"""KVQuantAttention -- the reference's patched LlamaAttention / LlamaFlashAttention2
(ML = /root/reference/deployment/transformers/src/transformers/models/llama/
modeling_llama.py:1388-2011) around the MI355X QuantK/QuantV operators.

Same protocol: pre-RoPE keys are cached compressed, RoPE is applied to Q only
(the K kernel rotates the dequantised keys on the fly), the first
`first_few_fp16` tokens live in fp16 side caches with POST-RoPE keys ("attention
sinks", ML:1464-1466), prefill packs the whole prompt in parallel and decode
feeds one token at a time (batch 1, MHA, as the reference asserts at
ML:1408/1801).

Decode is GPU-resident: 6 kernel launches for the KV path, no host
synchronisation (the reference does two device->host round trips per layer per
token, ML:707-714 and 1803-1820).  Prefill attention is kvq_prefill_attention (MFMA
flash kernel, csrc/kvq_prefill_attn.hip) where the reference calls flash-attn.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .cache import QuantK, QuantV, decode_kv


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class RotaryDynamic(nn.Module):
    """LlamaRotaryEmbeddingDynamic (ML:159-177): cos/sin for positions
    [start, end) computed on the fly, in the activation dtype."""

    def __init__(self, dim, base=10000.0, device=None):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float().to(device) / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self.dim, self.base = dim, float(base)

    # The tables of the last positions asked for, per frequency table / device / dtype: every layer of a model asks for
    # the SAME positions one after the other, and at decode the seven small launches below were a sizeable share of a
    # layer's host time (profiles/r04_full_model.txt).  Same arithmetic, once per token instead of once per layer.
    _last = {}

    def forward(self, x, start_pos, end_pos):
        inv = self.inv_freq
        key = (self.dim, self.base, inv.dtype, inv.device, x.dtype)
        # (only decode-size requests are kept -- a prefill's [S, dim] tables would be pinned until the next call, 512 MB
        #  after a 1M-token prompt -- and per CUDA stream: the cached tensors were produced on that stream)
        small = end_pos - start_pos <= RotaryDynamic.CACHE_MAX_POSITIONS
        if small:
            key = key + (torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0,)
            hit = RotaryDynamic._last.get(key)
            if hit is not None and hit[0] == (start_pos, end_pos):
                return hit[1], hit[2]
        t = torch.arange(start_pos, end_pos, device=x.device, dtype=torch.int64).type_as(inv)
        freqs = torch.outer(t, inv)
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos().to(x.dtype), emb.sin().to(x.dtype)
        if small:
            RotaryDynamic._last[key] = ((start_pos, end_pos), cos, sin)
        return cos, sin

    CACHE_MAX_POSITIONS = 16

    @classmethod
    def clear_cache(cls):
        cls._last.clear()


class KVQuantAttention(nn.Module):
    fuse_sinks = True     # decode: the fp16 sink tokens' matmuls inside the decode launches (False: torch glue as in the reference)

    def __init__(self, hidden_size=4096, num_heads=32, abits=4, include_sparse=True, first_few_fp16=0,
                 maxseqlen=4096, rope_theta=10000.0, sparsity_threshold=0.99, device=None,
                 dtype=torch.float16, bias=False, make_proj=True, use_orig_sparse=False, compact=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.first_few_fp16 = first_few_fp16
        self.rope_theta = rope_theta
        kw = dict(device=device, dtype=dtype)
        if make_proj:     # (kvquant_amd.llama wraps a model's own projections and only uses attend())
            self.q_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
            self.k_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
            self.v_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
            self.o_proj = nn.Linear(hidden_size, hidden_size, bias=bias, **kw)
        self.rotary_emb = RotaryDynamic(self.head_dim, base=rope_theta, device=device)
        self.kcache = QuantK(bits=abits, hidden_size=hidden_size, num_heads=num_heads,
                             max_position_embeddings=maxseqlen, include_sparse=include_sparse,
                             sparsity_threshold=sparsity_threshold, rope_theta=rope_theta,
                             first_few_fp16=first_few_fp16, device=device, compact=compact)
        self.vcache = QuantV(bits=abits, hidden_size=hidden_size, num_heads=num_heads,
                             max_position_embeddings=maxseqlen, include_sparse=include_sparse,
                             sparsity_threshold=sparsity_threshold, first_few_fp16=first_few_fp16,
                             device=device, compact=compact)
        if first_few_fp16 > 0:
            self.kcache_fp16 = torch.zeros((1, num_heads, self.head_dim, first_few_fp16), dtype=dtype,
                                           device=self.kcache.device)
            self.vcache_fp16 = torch.zeros((1, num_heads, first_few_fp16, self.head_dim), dtype=dtype,
                                           device=self.kcache.device)

    def load_quantizers(self, k_quantizer, v_quantizer, include_sparse=True, sparsity_threshold=0.99, norm=False):
        """deployment/llama.py:186-198."""
        self.kcache.reset()
        self.vcache.reset()
        self.kcache.load_lookup_table(k_quantizer, include_sparse, sparsity_threshold, norm)
        self.vcache.load_lookup_table(v_quantizer, include_sparse, sparsity_threshold, norm)

    def reset(self):
        self.kcache.reset()
        self.vcache.reset()

    # ------------------------------------------------------------------ core (post-projection states)
    def attend(self, query_states, key_states, value_states):
        """query/key/value_states: [1, H, q_len, hd] (pre-RoPE), activation dtype.
        Returns attn_output [1, q_len, hidden] (before o_proj)."""
        bsz, H, q_len, hd = query_states.shape
        assert bsz == 1, "batch 1 only (ML:1801)"
        sinks = self.first_few_fp16
        key_states = key_states.half()
        value_states = value_states.half()
        start = self.kcache.klen
        cos, sin = self.rotary_emb(value_states, start, start + q_len)
        fused = self.kcache.include_sparse and self.vcache.include_sparse
        if (q_len == 1 and fused and start >= sinks and query_states.dtype == torch.float16 and hd == 128
                and (sinks == 0 or (self.fuse_sinks and self.kcache_fp16.dtype == torch.float16))):
            # ---- decode over the compressed cache (ML:1948-2006), GPU-resident: the query's RoPE in one launch
            # (kvq_rope_q_f16: torch's fp16 roundings), then one library call for the layer's KV path
            q_rope = ops.rope_q_f16(query_states.reshape(H, hd), cos[0], sin[0])
            if sinks > 0:
                out, _ = decode_kv(self.kcache, self.vcache, q_rope, key_states.flatten(), value_states.flatten(),
                                   k_sink=self.kcache_fp16[0], v_sink=self.vcache_fp16[0])
            else:
                out, _ = decode_kv(self.kcache, self.vcache, q_rope, key_states.flatten(), value_states.flatten())
            return out.half().view(bsz, q_len, self.hidden_size)
        query_rope = (query_states * cos) + (rotate_half(query_states) * sin)

        if q_len > 1 and self.kcache.klen == 0:
            # ---- prefill (ML:1861-1927) ------------------------------------------------------------
            key_rope = (key_states * cos) + (rotate_half(key_states) * sin)
            # causal attention of the prompt on the matrix cores (kvq_prefill_attention, hand-written MFMA flash
            # kernel; the reference calls flash-attn here, ML:1869-1873); output token-major, ready for o_proj
            attn_output = ops.prefill_attention(query_rope[0].half(), key_rope[0], value_states[0]) \
                .view(bsz, q_len, self.hidden_size).to(query_states.dtype)
            if sinks > 0:
                n = min(sinks, q_len)
                self.kcache_fp16[:, :, :, :n] = key_rope[:, :, :n, :].transpose(2, 3)
                self.vcache_fp16[:, :, :n, :] = value_states[:, :, :n, :]
                if q_len > sinks:
                    self.kcache.parallel_pack(key_states[0, :, sinks:, :].transpose(1, 2))
                    self.vcache.parallel_pack(value_states[0, :, sinks:, :].transpose(1, 2))
                self.kcache.klen += n
                self.vcache.vlen += n
            else:
                self.kcache.parallel_pack(key_states[0].transpose(1, 2))
                self.vcache.parallel_pack(value_states[0].transpose(1, 2))
            return attn_output

        assert q_len == 1, "decode feeds one token at a time (generation/utils.py kvquant hooks)"
        inv = 1.0 / math.sqrt(hd)
        if self.kcache.klen < sinks:
            # ---- still filling the fp16 sink caches (ML:1932-1946, 1981-1984) ------------------------
            key_rope = (key_states * cos) + (rotate_half(key_states) * sin)
            self.kcache_fp16[:, :, :, self.kcache.klen] = key_rope.transpose(2, 3).squeeze(-1)
            self.kcache.klen += 1
            w = torch.matmul(query_rope, self.kcache_fp16[:, :, :, :self.kcache.klen]) / math.sqrt(hd)
            w = F.softmax(w, dim=-1, dtype=torch.float32).to(query_rope.dtype)
            self.vcache_fp16[:, :, self.vcache.vlen, :] = value_states.squeeze(2)
            self.vcache.vlen += 1
            out = torch.matmul(w, self.vcache_fp16[:, :, :self.vcache.vlen, :])
            return out.transpose(1, 2).reshape(bsz, q_len, self.hidden_size)

        # ---- decode over the compressed cache (ML:1948-2006), GPU-resident --------------------------
        sink_scores = None
        fused = self.kcache.include_sparse and self.vcache.include_sparse
        fuse_sinks = fused and sinks > 0 and self.fuse_sinks and self.kcache_fp16.dtype == torch.float16
        if sinks > 0 and not fuse_sinks:
            sink_scores = (torch.matmul(query_rope, self.kcache_fp16) / math.sqrt(hd))[0, :, 0, :].contiguous()
        if fuse_sinks:
            # the sink tokens' two fp16 matmuls, division and add ride in the decode launches (decode_kv docstring;
            # the sum of the two parts is then rounded to fp16 once instead of twice)
            out, _ = decode_kv(self.kcache, self.vcache, query_rope[0, :, 0, :].contiguous(), key_states.flatten(),
                               value_states.flatten(), k_sink=self.kcache_fp16[0], v_sink=self.vcache_fp16[0])
            out = out.transpose(0, 1).half().unsqueeze(0)
            return out.transpose(1, 2).contiguous().reshape(bsz, q_len, self.hidden_size)
        if fused:
            out, sink_probs = decode_kv(self.kcache, self.vcache, query_rope[0, :, 0, :].contiguous(),
                                        key_states.flatten(), value_states.flatten(), sink_scores)
            out = out.transpose(0, 1).half()                                   # [H, 1, hd]
        else:
            q32 = query_rope[0].float().transpose(0, 1).contiguous()             # [1, H, hd]
            scores = self.kcache.append_and_score(q32, key_states.flatten().float())
            probs, sink_probs = ops.softmax_scale(scores[0], inv, sink_scores)
            out = self.vcache.forward_fused_sparse(probs.unsqueeze(1).half(), value_states)
        out = out.unsqueeze(0)
        if sinks > 0:
            out = out + torch.matmul(sink_probs.view(1, H, 1, sinks), self.vcache_fp16)
        return out.transpose(1, 2).contiguous().reshape(bsz, q_len, self.hidden_size)

    def forward(self, hidden_states):
        """hidden_states [1, q_len, hidden] -> [1, q_len, hidden]."""
        bsz, q_len, _ = hidden_states.shape
        shape = (bsz, q_len, self.num_heads, self.head_dim)
        q = self.q_proj(hidden_states).view(shape).transpose(1, 2)
        k = self.k_proj(hidden_states).view(shape).transpose(1, 2)
        v = self.v_proj(hidden_states).view(shape).transpose(1, 2)
        return self.o_proj(self.attend(q, k, v))

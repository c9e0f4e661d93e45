"""BASELINE config 1 as an executable harness: the reference's SIMULATED-quantisation path
(quant/llama_simquant.py evaluates a model whose k_proj / v_proj outputs are fake-quantised by QuantLinearSim,
SQ:563-795) against the KERNEL path (kvquant_amd.llama: packed cache + HIP kernels) on the SAME random-init Llama,
tokens and quantizers, compared by perplexity.  Real LLaMA-2-7B weights / wikitext-2 are not available offline, so
the model is a seeded random-init Llama with the 7B head shape (32 heads x 128) and a reduced depth, the tokens are
seeded random ids, and the quantizers are calibrated on the model's own activations (kvquant_amd.calibrate).

TEST INFRASTRUCTURE (imports the oracle); used by tests/test_llama_gpu.py and tools/ppl_delta.py.
"""
import copy

import torch
import torch.nn as nn


def make_model(layers=2, vocab=32000, hidden=4096, heads=32, inter=2048, seed=0, device="cuda", maxseqlen=4096,
               dtype=torch.float16, **knobs):
    from transformers import LlamaConfig, LlamaForCausalLM
    from kvquant_amd import llama as kl
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=heads, max_position_embeddings=maxseqlen,
                      attention_bias=False, tie_word_embeddings=False)
    cfg = kl.kvquant_config(cfg, maxseqlen=maxseqlen, **knobs)
    torch.manual_seed(seed)
    model = LlamaForCausalLM(cfg).to(dtype).to(device).eval()
    return model


class MarkovStream:
    """A synthetic language with a KNOWN entropy: an order-1 Markov chain over the vocabulary in which every token has
    `fan` possible successors with Zipf(`alpha`) probabilities (seeded).  fan = 8, alpha = 1.2: 1.70 nats per token, i.e.
    an ideal perplexity of 5.48 -- the regime of wikitext-2 on LLaMA-2-7B (5.47), so that "within 0.01 PPL" means what
    it means in BASELINE.json instead of being measured on a random-init model whose PPL is the vocabulary size."""

    def __init__(self, vocab, fan=8, alpha=1.2, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.vocab, self.g = vocab, g
        self.succ = torch.randint(0, vocab, (vocab, fan), generator=g)
        p = 1.0 / torch.arange(1, fan + 1, dtype=torch.float64) ** alpha
        self.p = (p / p.sum()).float()
        self.entropy = float(-(self.p.double() * self.p.double().log()).sum())

    def sample(self, batch, length):
        cur = torch.randint(0, self.vocab, (batch,), generator=self.g)
        out = [cur]
        ks = torch.multinomial(self.p.expand(batch, -1), length - 1, replacement=True, generator=self.g)   # [batch, length-1]
        for t in range(length - 1):
            cur = self.succ[cur, ks[:, t]]
            out.append(cur)
        return torch.stack(out, dim=1)


def train_markov(model, stream, steps=300, batch=8, length=512, lr=1e-3, warmup=20, log=None):
    """a few hundred AdamW steps (fp32 weights) on the stream: enough for a 2-layer model to learn the transition table to
    a perplexity of O(10)"""
    dev = next(model.parameters()).device
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.95), weight_decay=0.0)
    last = None
    for step in range(steps):
        for gq in opt.param_groups:
            gq["lr"] = lr * min(1.0, (step + 1) / warmup) * (0.1 + 0.9 * 0.5 * (1 + __import__("math").cos(3.141592653589793 * step / steps)))
        ids = stream.sample(batch, length).to(dev)
        loss = model(ids, labels=ids, use_cache=False).loss
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        last = float(loss.detach())
        if log is not None and (step % 50 == 0 or step == steps - 1):
            log("train step %d loss %.3f (ppl %.1f)" % (step, last, __import__("math").exp(last)))
    model.eval()
    return last


class FakeQuantLinear(nn.Module):
    """QuantLinearSim.forward (SQ:700-795) around an existing Linear: y = fake_quant(half(x) @ W), per-channel
    static (K) or per-token dynamic (V)."""

    def __init__(self, lin, quantizer, bits, is_k, sparsity_threshold, first_few_fp16, norm):
        super().__init__()
        self.lin, self.q, self.bits, self.is_k = lin, quantizer, bits, is_k
        self.thr, self.ff, self.norm = sparsity_threshold, first_few_fp16, norm

    def forward(self, x):
        from oracle import simquant as sq
        y = self.lin(x.half())
        shape = y.shape
        y2 = y.reshape(-1, shape[-1]).float()
        if self.is_k:
            out = sq.fake_quant_k(y2, self.q, self.bits, include_sparse=True, cap_outliers=21, first_few_fp16=self.ff,
                                  norm=self.norm)
        else:
            out = sq.fake_quant_v(y2, self.q, self.bits, include_sparse=True, sparsity_threshold=self.thr,
                                  first_few_fp16=self.ff, norm=self.norm)
        return out.reshape(shape)


class DeployRuleLinear(nn.Module):
    """k_proj / v_proj followed by the DEPLOYMENT classes' quantise -> dequantise of every token, from the oracle's
    restatement of their glue (oracle.glue: build_k_lut, k_outlier_rows, v_token_rows; ML:437-501, 706-751, 1086-1176):
    K: nearest code against the per-channel table of fp16-rounded thresholds, the 21 + 21 capped outliers of the
    RESCALED token kept as residuals to the end codes;  V: per-token thresholds = the 22nd largest / smallest value,
    codebook row lut * sf + off, values strictly outside clipped to the zero-point code, the 21 + 21 extreme values
    kept as residuals to it (the reference's tie quirk included).  This is what the kernels must reproduce to rounding
    -- unlike the simulated path, whose V outliers come from a quantile (SQ:95-108)."""

    def __init__(self, lin, quantizer, bits, is_k, heads, first_few_fp16, norm):
        super().__init__()
        self.lin, self.q, self.bits, self.is_k, self.heads, self.ff, self.norm = lin, quantizer, bits, is_k, heads, first_few_fp16, norm

    def forward(self, x):
        from oracle import glue
        y = self.lin(x.half())
        shape = y.shape
        y2 = y.reshape(-1, shape[-1]).float()
        dev, C, n, thr_k = y2.device, y2.shape[1], 2 ** self.bits, glue.threshold_k(0.99, y2.shape[1])
        if self.is_k:
            t = glue.build_k_lut(self.q, self.bits, self.heads, C // self.heads, norm=self.norm)
            lut = t["lookup_table"].reshape(C, n).to(dev)
            lut_off = (t["lookup_table2"] if self.norm else t["lookup_table"]).reshape(C, n).to(dev)
            lut_deq = (t["lookup_table2"] if (self.norm and self.bits == 2) else t["lookup_table"]).reshape(C, n).to(dev)
            up, lo = t["upper"].to(dev), t["lower"].to(dev)
            code = (lut.unsqueeze(0) - y2.unsqueeze(-1)).abs().argmin(dim=-1)            # first minimum (KCU:1230-1237)
            deq = torch.gather(lut_deq.unsqueeze(0).expand(y2.shape[0], -1, -1), 2, code.unsqueeze(-1)).squeeze(-1)
            resc = (y2 - (up + lo) / 2) / ((up - lo) / 2)                                # KCU:1759-1764
            vals, idx = glue.k_outlier_rows(y2.cpu(), resc.cpu(), lut_off.cpu(), self.bits, thr_k)
            out = deq.clone()
            out.scatter_add_(1, idx.long().to(dev), vals.to(dev))
        else:
            lut_sorted = torch.as_tensor(self.q[2][0]).squeeze(-1).float().sort().values.to(dev)
            r = glue.v_token_rows(y2.cpu(), lut_sorted.cpu(), self.bits, thr_k)
            rows = r["lut_rows"].to(dev)
            if self.norm:
                ns, no = torch.as_tensor(self.q[3]).float().to(dev), torch.as_tensor(self.q[4]).float().to(dev)
                sf, off = (r["maxval"] - r["minval"]).to(dev) / 2, (r["maxval"] + r["minval"]).to(dev) / 2
                rows2 = (lut_sorted * ns + no).unsqueeze(0) * sf.unsqueeze(-1) + off.unsqueeze(-1)
            deq_rows = rows2 if (self.norm and self.bits == 2) else rows
            z = glue.ZERO_CODE[self.bits]
            code = (rows.unsqueeze(1) - y2.unsqueeze(-1)).abs().argmin(dim=-1)
            clip = (y2 < r["minval"].to(dev).unsqueeze(-1)) | (y2 > r["maxval"].to(dev).unsqueeze(-1))      # KCU:2084 (strict)
            code = torch.where(clip, torch.full_like(code, z), code)
            out = torch.gather(deq_rows, 1, code).clone()
            vals = r["vals"].to(dev)
            if self.norm and self.bits == 2:      # residuals refer to the table the kernel dequantises with (ML:1153-1156)
                vals = vals + (rows[:, z] - rows2[:, z]).unsqueeze(-1)
            out.scatter_add_(1, r["idx"].long().to(dev), vals)
        if self.ff > 0:
            out[:self.ff] = y2[:self.ff]                                                  # fp16 sink tokens stay as they are
        return out.reshape(shape)


@torch.no_grad()
def deploy_rule_ppl(model, ids, quantizers, bits, first_few_fp16=-1, norm=False, n_prompt=0, scores64=False):
    """the deployment classes' OWN quantisation (oracle.glue restatement) + the deployment path's attention arithmetic, in
    plain torch: the comparator the kernel path has to match to rounding (bar 2e-4 relative perplexity)"""
    import types
    m = copy.deepcopy(model)
    cfg = m.config
    theta = getattr(cfg, "rope_theta", None) or (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
    for i, layer in enumerate(m.model.layers):
        at = layer.self_attn
        at.k_proj = DeployRuleLinear(at.k_proj, quantizers["model.layers.%d.self_attn.k_proj" % i], bits, True,
                                     cfg.num_attention_heads, first_few_fp16, norm)
        at.v_proj = DeployRuleLinear(at.v_proj, quantizers["model.layers.%d.self_attn.v_proj" % i], bits, False,
                                     cfg.num_attention_heads, first_few_fp16, norm)
        at.kvq_heads, at.kvq_hd, at.kvq_theta = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads, float(theta)
        at.kvq_ff = max(first_few_fp16, 0)
        at.kvq_nprompt = n_prompt
        at.kvq_scores64 = bool(scores64)
        at.forward = types.MethodType(_deploy_arith_forward, at)
    return ppl_full_sequence(m, ids)


@torch.no_grad()
def ppl_full_sequence(model, ids):
    """next-token perplexity of one full-sequence forward (what llama_simquant.py's llama_eval computes per sample)"""
    logits = model(ids, use_cache=False).logits[0, :-1].float()
    return float(torch.exp(nn.functional.cross_entropy(logits, ids[0, 1:])))


@torch.no_grad()
def sim_path_ppl(model, ids, quantizers, bits, sparsity_threshold=0.99, first_few_fp16=-1, norm=False):
    """the reference's simulated path: fake-quantised k_proj / v_proj outputs, stock attention"""
    m = copy.deepcopy(model)
    for i, layer in enumerate(m.model.layers):
        at = layer.self_attn
        at.k_proj = FakeQuantLinear(at.k_proj, quantizers["model.layers.%d.self_attn.k_proj" % i], bits, True,
                                    sparsity_threshold, first_few_fp16, norm)
        at.v_proj = FakeQuantLinear(at.v_proj, quantizers["model.layers.%d.self_attn.v_proj" % i], bits, False,
                                    sparsity_threshold, first_few_fp16, norm)
    return ppl_full_sequence(m, ids)


def _deploy_arith_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None,
                          **kwargs):
    """full-sequence attention over fake-quantised K / V in the DEPLOYMENT path's dtype order (ML:873-874,
    1948-1995): RoPE on the dequantised keys in fp32 with fl32(theta_j * pos) angles, fp32 q.K^T -> half -> / sqrt(d)
    in fp16 -> fp32 softmax -> half probabilities -> fp32 p.V -> half.  Plain torch, no kernels."""
    import math
    from kvquant_amd import ops
    bsz, T, _ = hidden_states.shape
    H, hd = self.kvq_heads, self.kvq_hd
    dev = hidden_states.device
    q = self.q_proj(hidden_states).view(T, H, hd).transpose(0, 1).half()
    k = self.k_proj(hidden_states).view(T, H, hd).transpose(0, 1).float()        # fake-quantised already
    v = self.v_proj(hidden_states).view(T, H, hd).transpose(0, 1).float()
    th = ops.rope_freqs(self.kvq_theta, dev)                                     # theta_j as the kernels evaluate them
    pos = torch.arange(T, device=dev, dtype=torch.float32)
    ang = (th.unsqueeze(0) * pos.unsqueeze(1))                                   # fl32(theta * pos)
    cos, sin = torch.cat((ang.cos(), ang.cos()), -1), torch.cat((ang.sin(), ang.sin()), -1)

    def rope(x):
        rot = torch.cat((-x[..., hd // 2:], x[..., :hd // 2]), dim=-1)
        return x * cos + rot * sin
    # the query is rotated in the activation dtype with fp16 cos / sin tables (attention.py: RotaryDynamic), the keys in fp32
    inv = 1.0 / (self.kvq_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float().to(dev) / hd))
    fr = torch.outer(pos, inv)
    emb = torch.cat((fr, fr), dim=-1)
    c16, s16 = emb.cos().half(), emb.sin().half()
    qr = q * c16 + torch.cat((-q[..., hd // 2:], q[..., :hd // 2]), dim=-1) * s16
    kr = rope(k)
    ff = self.kvq_ff
    if ff > 0:      # fp16 sink tokens: post-RoPE keys in fp16 (ML:1464-1466)
        kh = self.k_proj.lin(hidden_states.half()).view(T, H, hd).transpose(0, 1)
        khr = kh * c16 + torch.cat((-kh[..., hd // 2:], kh[..., :hd // 2]), dim=-1) * s16
        kr[:, :ff] = khr[:, :ff].float()
    if getattr(self, "kvq_scores64", False):
        # the same scores summed in fp64 and rounded once: how much of a perplexity difference is nothing but the order of
        # an fp32 summation in front of the half() of the next line (run() reports the spread)
        scores = torch.matmul(qr.double(), kr.double().transpose(1, 2)).float()
    else:
        scores = torch.matmul(qr.float(), kr.transpose(1, 2))                    # [H, T, T] fp32
    w = (scores.half().float() * (1.0 / math.sqrt(hd))).half().float()
    mask = torch.ones(T, T, device=dev, dtype=torch.bool).tril()
    w = w.masked_fill(~mask, float("-inf"))
    p = torch.softmax(w, dim=-1, dtype=torch.float32).half().float()
    out = torch.matmul(p, v).half()                                              # [H, T, hd]
    npr = getattr(self, "kvq_nprompt", 0)
    if npr > 0:
        # the prompt's own rows: a parallel prefill attends to the UNQUANTISED fp16 K / V (ML:1861-1874, flash attention)
        kh = self.k_proj.lin(hidden_states.half()).view(T, H, hd).transpose(0, 1)[:, :npr]
        vh = self.v_proj.lin(hidden_states.half()).view(T, H, hd).transpose(0, 1)[:, :npr]
        khr = kh * c16[:npr] + torch.cat((-kh[..., hd // 2:], kh[..., :hd // 2]), dim=-1) * s16[:npr]
        sp = torch.matmul(qr[:, :npr].float(), khr.float().transpose(1, 2)) * (1.0 / math.sqrt(hd))
        sp = sp.masked_fill(~mask[:npr, :npr], float("-inf"))
        pp = torch.softmax(sp, dim=-1, dtype=torch.float32).half().float()
        out[:, :npr] = torch.matmul(pp, vh.float()).half()
    out = out.transpose(0, 1).reshape(1, T, H * hd)
    return self.o_proj(out.to(hidden_states.dtype)), None


@torch.no_grad()
def sim_deploy_arith_ppl(model, ids, quantizers, bits, sparsity_threshold=0.99, first_few_fp16=-1, norm=False, n_prompt=0):
    """simulated quantisation of K / V (the reference's functions) + the deployment path's attention arithmetic"""
    import types
    m = copy.deepcopy(model)
    cfg = m.config
    theta = getattr(cfg, "rope_theta", None) or (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
    for i, layer in enumerate(m.model.layers):
        at = layer.self_attn
        at.k_proj = FakeQuantLinear(at.k_proj, quantizers["model.layers.%d.self_attn.k_proj" % i], bits, True,
                                    sparsity_threshold, first_few_fp16, norm)
        at.v_proj = FakeQuantLinear(at.v_proj, quantizers["model.layers.%d.self_attn.v_proj" % i], bits, False,
                                    sparsity_threshold, first_few_fp16, norm)
        at.kvq_heads, at.kvq_hd, at.kvq_theta = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads, float(theta)
        at.kvq_ff = max(first_few_fp16, 0)
        at.kvq_nprompt = n_prompt
        at.forward = types.MethodType(_deploy_arith_forward, at)
    return ppl_full_sequence(m, ids)


@torch.no_grad()
def kernel_path_ppl(model, ids, quantizers, sparsity_threshold=0.99, norm=False, n_prompt=0):
    """the deployment path: patched attention over the packed cache; token by token (deployment/llama.py --check) or,
    with n_prompt > 0, a parallel prefill of the prompt followed by token-by-token decode"""
    from kvquant_amd import llama as kl
    m = copy.deepcopy(model)
    kl.patch_llama(m, sparsity_threshold=sparsity_threshold)
    kl.load_quantizers(m, quantizers, include_sparse=True, sparsity_threshold=sparsity_threshold, norm=norm)
    if n_prompt:
        return kl.prefill_then_decode(m, ids, n_prompt)["ppl"]
    return kl.benchmark(m, ids, check=True)["ppl"]


_TRAINED = {}


def trained_model(layers, vocab, seed, device, maxseqlen, train_steps, hidden=4096, heads=32, inter=2048, log=None, **knobs):
    """the seeded model after `train_steps` steps on the seeded Markov stream (cached per process: the configurations of a
    test session share one training run), as an fp16 copy with the given cache knobs"""
    key = (layers, vocab, seed, str(device), train_steps, hidden, heads, inter)
    if key not in _TRAINED:
        m = make_model(layers=layers, vocab=vocab, hidden=hidden, heads=heads, inter=inter, seed=seed, device=device,
                       maxseqlen=8192, dtype=torch.float32)
        stream = MarkovStream(vocab, seed=seed)
        loss = train_markov(m, stream, steps=train_steps, log=log)
        _TRAINED[key] = (m.half().state_dict(), stream, loss)
        del m
        if str(device).startswith("cuda"):
            torch.cuda.empty_cache()
    sd, stream, loss = _TRAINED[key]
    model = make_model(layers=layers, vocab=vocab, hidden=hidden, heads=heads, inter=inter, seed=seed, device=device,
                       maxseqlen=maxseqlen, **knobs)
    model.load_state_dict(sd)
    return model, stream, loss


def run(layers=2, n_tokens=512, bits=4, first_few_fp16=0, vocab=32000, seed=0, n_prompt=0, norm=False, device="cuda",
        train_steps=0, hidden=4096, heads=32, inter=2048, log=None):
    from kvquant_amd import calibrate
    g = torch.Generator().manual_seed(seed)
    if train_steps > 0:
        model, stream, _ = trained_model(layers, vocab, seed, device, n_tokens + 64, train_steps, hidden=hidden, heads=heads,
                                         inter=inter, log=log, abits=bits, include_sparse=True, first_few_fp16=first_few_fp16)
        ids = stream.sample(1, n_tokens).to(device)
        calib = stream.sample(1, 2048).to(device)
    else:
        model = make_model(layers=layers, vocab=vocab, hidden=hidden, heads=heads, inter=inter, seed=seed, device=device,
                           maxseqlen=n_tokens + 64, abits=bits, include_sparse=True, first_few_fp16=first_few_fp16)
        ids = torch.randint(0, vocab, (1, n_tokens), generator=g).to(device)
        calib = torch.randint(0, vocab, (1, 2048), generator=g).to(device)
    quantizers = calibrate.calibrate_llama(model, calib, bits=bits, include_sparse=True, sparsity_threshold=0.99, norm=norm)
    base = ppl_full_sequence(model, ids)
    ff = first_few_fp16 if first_few_fp16 else -1
    sim = sim_path_ppl(model, ids, quantizers, bits, first_few_fp16=ff, norm=norm)
    simd = sim_deploy_arith_ppl(model, ids, quantizers, bits, first_few_fp16=ff, norm=norm, n_prompt=n_prompt)
    ker = kernel_path_ppl(model, ids, quantizers, norm=norm, n_prompt=n_prompt)
    try:
        dep = deploy_rule_ppl(model, ids, quantizers, bits, first_few_fp16=ff, norm=norm, n_prompt=n_prompt)
        dep64 = deploy_rule_ppl(model, ids, quantizers, bits, first_few_fp16=ff, norm=norm, n_prompt=n_prompt, scores64=True)
    except Exception as e:          # (never take the other comparators down with it)
        dep = dep64 = float("nan")
        if log:
            log("deploy_rule_ppl failed: %s: %s" % (type(e).__name__, e))
    return {"ppl_deploy_rule": dep, "rel_delta_vs_deploy_rule": (ker - dep) / dep,
            # the comparator against ITSELF with the scores summed in another order (fp64): the noise floor of any comparison
            # through the deployment path's half(score)
            "ppl_deploy_rule_scores_fp64": dep64, "rel_summation_order_noise": (dep64 - dep) / dep,"layers": layers, "tokens": n_tokens, "bits": bits, "first_few_fp16": first_few_fp16, "vocab": vocab,
            "n_prompt": n_prompt, "norm": norm, "train_steps": train_steps, "ppl_fp16": base, "ppl_sim": sim, "ppl_sim_deploy_arith": simd,
            "ppl_kernel": ker, "delta": ker - sim, "rel_delta": (ker - sim) / sim,
            "rel_delta_vs_deploy_arith": (ker - simd) / simd}

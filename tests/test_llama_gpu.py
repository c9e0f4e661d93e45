"""kvquant_amd.llama on a stock Hugging Face Llama: the reference's driver protocol (config knobs, quantizer pickle,
kcache / vcache attributes, token-by-token benchmark loop with --check) and the BASELINE config-1 acceptance --
perplexity of the kernel path against the reference's simulated-quantisation path on the same model."""
import math
import pickle

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests import util
    util.sync_oracle_freqs(10000.0)
    return torch.device("cuda:0")


NORTH_STAR = 0.01 / 5.47   # BASELINE.json north_star: wikitext-2 perplexity within 0.01 of the reference's 5.47 = 1.83e-3 relative

# BASELINE config 1 acceptance.  Real LLaMA-2-7B weights and wikitext-2 are not available offline; the stand-in is a
# 2-layer Llama with the 7B head shape TRAINED (600 AdamW steps, tests/ppl_harness.py) on a synthetic Markov language
# whose ideal perplexity is 5.48, evaluated on 2048 fresh tokens of it: a model that predicts (PPL ~ 9.3), so a relative
# perplexity difference means here what it means in the north star.  Three comparators on the same model, tokens and
# calibrated quantizers:
#   sim   the reference's simulated path (quant/llama_simquant.py: fake-quantised k_proj / v_proj, fp32 attention);
#   simd  the same fake-quantised K / V through the DEPLOYMENT path's arithmetic (fp16 score scaling and probabilities,
#         ML:1972-1976; with a prompt: the prompt's rows over the unquantised K / V, ML:1861-1874) -- what the kernels
#         have to reproduce;
#   fp16  the unquantised model;
#   rule  (round 5) the DEPLOYMENT classes' own quantise -> dequantise of every token, restated by the oracle
#         (oracle.glue: fp16-rounded thresholds, capped outliers, V thresholds from the 22nd largest / smallest value,
#         strict clipping) + the deployment arithmetic in torch: the same numbers the kernels produce, summed in another
#         order.  Two draws (profiles/r05_b_ppl_delta.jsonl): nuq4 -1.5e-4 / +4.8e-4, with prompt +4.9e-4 / -1.1e-4,
#         nuq3 + 5 sinks +1.7e-3 / -1.5e-5, nuq2 + Q-Norm -8.9e-4 / +8.3e-5 -- both signs, no bias; the comparator's
#         own perplexity moves by up to 2e-4 when ITS scores are summed in fp64 (the half() of every score turns a
#         last-bit difference into a 2.8e-3 step of a probability).
# Bars are FIXED (no data-dependent xfail): measured over three draws in round 4 (profiles/r04_ppl_delta.jsonl) --
#   nuq4 (+ prompt): |kernel - simd| <= 6.2e-4, |kernel - sim| <= 1.7e-3 (prompt: 4.3e-3, sim does not model the prefill);
#   nuq3 + 5 sinks:  <= 2.3e-3 / 5.1e-3 (the reference's two paths pick V outliers differently -- quantile vs top-21 --
#                    which moves a token's scale, and at 3 bit that shows);   nuq2 + Q-Norm: <= 3.7e-4 / 5.2e-3.
PPL_CASES = [
    # kw, bar vs simd, bar vs sim, bar vs rule
    (dict(bits=4), NORTH_STAR, 5e-3, NORTH_STAR),
    (dict(bits=4, n_prompt=1024), NORTH_STAR, 1e-2, NORTH_STAR),
    # (vs the deployment rule: five draws -- profiles/r05_b_ppl_delta.jsonl +1.7e-3 / -1.5e-5, r05_z +1.3e-4,
    #  profiles/r06_i_ppl_cfg3.jsonl +3.0e-5 / +6.3e-4 / -3.1e-5 -- both signs, largest 1.7e-3: bar = 1.5 x that)
    (dict(bits=3, first_few_fp16=5), 5e-3, 1e-2, 2.5e-3),
    (dict(bits=2, norm=True), NORTH_STAR, 1e-2, NORTH_STAR),
]


@pytest.mark.parametrize("kw,bar_simd,bar_sim,bar_rule", PPL_CASES,
                         ids=["nuq4", "nuq4-prompt1024", "nuq3-sink5", "nuq2-qnorm"])
def test_ppl_kernel_path_vs_reference_paths(gpu, kw, bar_simd, bar_sim, bar_rule):
    from tests import ppl_harness
    r = ppl_harness.run(layers=2, n_tokens=2048, vocab=4096, train_steps=600, **kw)
    print(r)
    assert math.isfinite(r["ppl_kernel"]) and math.isfinite(r["ppl_sim"]) and math.isfinite(r["ppl_fp16"])
    assert r["ppl_fp16"] < 20, "the stand-in model did not train (ideal perplexity of the stream: 5.48)"
    assert abs(r["rel_delta_vs_deploy_arith"]) < bar_simd, r
    assert abs(r["rel_delta"]) < bar_sim, r
    assert abs(r["rel_delta_vs_deploy_rule"]) < bar_rule, r
    # quantisation is not a no-op: the kernel path is NOT the fp16 model (nuq4 measured >= 1.2e-3 away, coarser ones more)
    assert abs(r["ppl_kernel"] - r["ppl_fp16"]) / r["ppl_fp16"] > 1e-4, r


def test_driver_protocol(gpu, tmp_path):
    """deployment/llama.py:165-216 against the patched stock model: knobs on the config, quantizers from a pickle,
    per-layer kcache / vcache objects with reset() / load_lookup_table(), benchmark(check=True); set_devices on the
    visible GPU(s)."""
    from kvquant_amd import calibrate, llama as kl
    from kvquant_amd.cache import QuantK, QuantV
    from tests import ppl_harness
    model = ppl_harness.make_model(layers=2, vocab=2048, inter=1024, maxseqlen=256, abits=4, include_sparse=True,
                                   first_few_fp16=2, device=gpu)
    assert model.config.abits == 4 and model.config.first_few_fp16 == 2 and model.config.dynamicrope is True
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 2048, (1, 64), generator=g)
    quant = calibrate.calibrate_llama(model, torch.randint(0, 2048, (1, 512), generator=g).to(gpu), bits=4)
    path = tmp_path / "quantizers.pickle"
    quant["model.layers.0.self_attn.k_proj.lut"] = "skipped like deployment/llama.py:188"
    with open(path, "wb") as f:
        pickle.dump(quant, f)
    placement = kl.set_devices(model)
    assert len(placement) == 2 and all(d.type == "cuda" for d in placement)
    kl.patch_llama(model)
    at = model.model.layers[1].self_attn
    assert isinstance(at.kcache, QuantK) and isinstance(at.vcache, QuantV) and at.kcache_fp16.shape == (1, 32, 128, 2)
    kl.load_quantizers(model, str(path), include_sparse=True, sparsity_threshold=0.99)
    r = kl.benchmark(model, ids, check=True)
    assert len(r["times"]) == 64 and math.isfinite(r["ppl"]) and r["median_s"] > 0
    assert at.kcache.klen == 64 and at.vcache.vlen == 64
    # a second pass after reset reproduces the first bit for bit (no atomics anywhere in the path)
    kl.reset_caches(model)
    assert at.kcache.klen == 0
    r2 = kl.benchmark(model, ids, check=True)
    assert r2["ppl"] == r["ppl"]


def test_command_line_driver_and_generate(gpu, tmp_path, capsys):
    """`python -m kvquant_amd.llama <model> <dataset> --abits 4 --include_sparse ... --benchmark N --check`
    (deployment/llama.py:100-216) on a small random-init Llama saved to disk, offline dataset `synthetic`; then the
    greedy generate(kvquant=True) loop (generation/utils.py:2325-2416): prompt through the model once, one token at a
    time afterwards; max_length / max_new_tokens stop it where the reference's loop stops; a second run from a fresh
    cache reproduces the tokens (the cache path is deterministic)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from kvquant_amd import calibrate, llama as kl
    torch.manual_seed(3)
    cfg = LlamaConfig(vocab_size=512, hidden_size=1024, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=256,
                      attention_bias=False, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).half()
    mdir = tmp_path / "tiny-llama"
    model.save_pretrained(str(mdir))
    model = model.to(gpu).eval()
    g = torch.Generator().manual_seed(4)
    quant = calibrate.calibrate_llama(model, [torch.randint(0, 512, (1, 256), generator=g) for _ in range(2)], bits=4,
                                      cap_outliers=True, first_few_fp16=1)
    qpath = tmp_path / "quantizers.pickle"
    with open(qpath, "wb") as f:
        pickle.dump(quant, f)
    del model
    rc = kl.main([str(mdir), "synthetic", "--abits", "4", "--include_sparse", "--sparsity-threshold", "0.99",
                  "--first_few_fp16", "1", "--maxseqlen", "128", "--seqlen", "64", "--nsamples", "2",
                  "--quantizer-path", str(qpath), "--benchmark", "24", "--check", "--generate", "5"])
    out = capsys.readouterr().out
    assert rc == 0 and "Load quantizers." in out and "Benchmarking ..." in out and "Median:" in out
    ppl = float(out.split("PPL:")[1].split()[0])
    assert math.isfinite(ppl) and 100 < ppl < 5000          # random-init model: PPL ~ vocabulary size
    assert "generated:" in out

    # generate(): prefill + one token at a time equals feeding the same tokens token by token from an empty cache
    m2 = kl.get_model(str(mdir), 64, 128, 4, True, 1).to(gpu).eval()
    kl.patch_llama(m2)
    kl.load_quantizers(m2, str(qpath), True, 0.99)
    prompt = torch.randint(0, 512, (1, 12), generator=g)
    seq = kl.generate(m2, prompt, max_new_tokens=6)
    assert seq.shape == (1, 18) and torch.equal(seq[:, :12].cpu(), prompt)
    assert not seq.requires_grad                                   # generate() runs under no_grad (HF's does)
    with torch.enable_grad():
        m3 = _fresh(kl, mdir, qpath, gpu)
        hooked = []
        h = m3.lm_head.register_forward_hook(lambda mod, i, o: hooked.append(o.requires_grad))
        kl.generate(m3, prompt, max_new_tokens=2)
        h.remove()
    assert hooked and not any(hooked), "generate() built an autograd graph"
    assert m2.model.layers[0].self_attn.kcache.klen == 17      # the last generated token was never fed back
    seq2 = kl.generate(_fresh(kl, mdir, qpath, gpu), prompt, max_length=15)
    assert seq2.shape == (1, 15) and torch.equal(seq2, seq[:, :15])
    # sampling (lwm/llama_inference.py:124-131): top_k = 1 is the greedy path; a seeded sampler is reproducible
    seq3 = kl.generate(_fresh(kl, mdir, qpath, gpu), prompt, max_new_tokens=6, do_sample=True, top_k=1)
    assert torch.equal(seq3, seq)
    gs = [torch.Generator(device=gpu).manual_seed(11) for _ in range(2)]
    sa = kl.generate(_fresh(kl, mdir, qpath, gpu), prompt, max_new_tokens=8, do_sample=True, temperature=0.7, top_p=0.9, generator=gs[0])
    sb = kl.generate(_fresh(kl, mdir, qpath, gpu), prompt, max_new_tokens=8, do_sample=True, temperature=0.7, top_p=0.9, generator=gs[1])
    assert sa.shape == (1, 20) and torch.equal(sa, sb)
    # the long-context inference driver (lwm/llama_inference.py), prompt given as token ids (no tokenizer offline)
    rc = kl.inference_main([str(mdir), "--abits", "4", "--include_sparse", "--sparsity-threshold", "0.99",
                            "--first_few_fp16", "1", "--maxseqlen", "128", "--quantizer-path", str(qpath),
                            "--token-ids", ",".join(str(int(t)) for t in prompt[0]), "--min_length", "14",
                            "--max_length", "20", "--seed", "5"])
    out = capsys.readouterr().out
    assert rc == 0 and "Load quantizers." in out and "Time:" in out
    ids = [int(t) for t in out.split("token ids: [")[1].split("]")[0].split(",")]
    assert ids[:12] == prompt[0].tolist() and 14 <= len(ids) <= 20


def test_compact_outlier_format_through_the_model(gpu, tmp_path):
    """patch_llama(compact=True): the opt-in 4-byte outlier entries through the whole model path (parallel prefill pack,
    GPU-resident decode); PPL next to the reference format's -- the residuals are rounded to fp16, nothing else changes"""
    from transformers import LlamaConfig, LlamaForCausalLM
    from kvquant_amd import calibrate, llama as kl
    torch.manual_seed(5)
    cfg = LlamaConfig(vocab_size=512, hidden_size=1024, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=256,
                      attention_bias=False, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).half().to(gpu).eval()
    g = torch.Generator().manual_seed(6)
    quant = calibrate.calibrate_llama(model, [torch.randint(0, 512, (1, 256), generator=g) for _ in range(2)], bits=4)
    ids = torch.randint(0, 512, (1, 96), generator=g)
    res = {}
    for compact in (False, True):
        kl.kvquant_config(model.config, abits=4, include_sparse=True, maxseqlen=128, first_few_fp16=0)
        kl.patch_llama(model, sparsity_threshold=0.99, compact=compact)
        kl.load_quantizers(model, quant, True, 0.99)
        assert model.model.layers[0].self_attn.vcache.compact == compact
        ppl_tok = kl.benchmark(model, ids, check=True)["ppl"]
        kl.reset_caches(model)
        res[compact] = (ppl_tok, kl.prefill_then_decode(model, ids, 48, check=True)["ppl"])
    for a, b in zip(res[False], res[True]):
        assert math.isfinite(b) and abs(b - a) / a < 5e-3, res


def _fresh(kl, mdir, qpath, gpu):
    m = kl.get_model(str(mdir), 64, 128, 4, True, 1).to(gpu).eval()
    kl.patch_llama(m)
    kl.load_quantizers(m, str(qpath), True, 0.99)
    return m

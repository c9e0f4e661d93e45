"""The driver-facing contract of bench.py: one JSON line with the fields the task names, on a small configuration (the
default one is the 128K line; this runs the same code paths in seconds)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, cwd=ROOT,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--ctx", "2048", "--layers", "2", "--steps", "3", "--warmup", "1", "--cpu-sample-tokens", "256", "--no-configs")
    assert "configs" not in d
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "kernels", "full_model", "fp16_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 1e-6          # tokens/s x s/token (1 stream)
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0          # (kept only for the 128K line, and only while the kernels' code matches)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    fm = d["full_model"]
    assert "error" not in fm, fm
    assert fm["tokens_per_s"] > 0 and fm["layers"] == 2


def test_bench_line_carries_every_baseline_config():
    """The default 1-GPU line carries the other BASELINE configurations as short legs under `configs`:
    config 2 at 4K and 32K, config 3 (nuq3 + 5 sinks at 128K), the 1M-token shape of config 5 on one GPU, config 4 (prefill)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--ctx", "2048", "--layers", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fp16-baseline",
             "--no-full-model")
    cfgs = d["configs"]
    assert len(cfgs) == 5, [c.get("label") for c in cfgs]
    for c in cfgs:
        assert "error" not in c, c
        assert c["value"] > 0 and c["ms_per_step"] > 0 and 0 < c["frac"] < 1, c
    labels = " | ".join(c["label"] for c in cfgs)
    for want in ("4K decode", "32K decode", "nuq3 + 5 fp16 sink tokens, 128K", "1M tokens", "prefill S=8192"):
        assert want in labels, (want, labels)
    decode = [c for c in cfgs if "decode" in c["label"] or "1M" in c["label"]]
    assert all(("score_k_us" in c and "mix_v_us" in c) or "fused_attend_us" in c for c in decode), decode
    pre = [c for c in cfgs if "prefill" in c["label"]][0]
    assert pre["pack_k_us"] > 0 and pre["pack_v_us"] > 0 and pre["prefill_attention_TFLOPs"] > 0


def test_bench_head_sharded_single_rank():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--shard", "heads", "--ctx", "2048", "--layers", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
             "--no-fp16-baseline")
    assert d["scaling"] == "strong" and d["value"] > 0 and "head-sharded" in d["config"]["parallelism"]


def test_bench_prefill_line_contract():
    """BASELINE config 4 as a bench line (`--prefill`): pack K / V and the MFMA attention of an 8192-token prompt, with the
    roofline arithmetic of the pack kernel written out."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--prefill", "--bits", "3")
    for key in ("metric", "value", "unit", "n_gpus", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config",
                "roofline", "kernels"):
        assert key in d, key
    assert d["config"]["S"] == 8192 and d["config"]["bits"] == 3 and d["value"] > 0 and d["vs_baseline"] is None
    k = d["kernels"]
    for key in ("pack_k_us", "pack_v_us", "prefill_attention_us", "prefill_attention_TFLOPs"):
        assert k[key] > 0, key
    assert abs(d["ms_per_step"] * 1000.0 - (k["pack_k_us"] + k["pack_v_us"])) < 1e-3 * d["ms_per_step"] * 1000.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]

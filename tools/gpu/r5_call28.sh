#!/bin/bash
# round 5, call 28: the sweep on the last state of the tree (every BASELINE configuration + the config-4 lines + the default line)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_y
timeout 1800 python bench.py --no-cpu-baseline --no-fp16-baseline --no-full-model --sweep 2> ${O}_sweep.err > ${O}_sweep.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05_y_sweep.jsonl"):
    if not l.startswith("{"): continue
    d = json.loads(l); c = d.get("config", {}); k = d.get("kernels", {})
    if "mix_v_us" in k:
        print(c.get("ctx"), c.get("bits"), c.get("sinks"), c.get("layers"), c.get("outlier_format", "")[:7], c.get("score_tables", "")[:5], "| %.3f ms  %.1f tok/s  frac %.3f step %.3f  K %.1f V %.1f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("roofline_step", {}).get("frac", 0), k.get("score_k_us") or 0, k["mix_v_us"]))
    elif "pack_k_us" in k:
        print(c.get("label"), "| K %.1f V %.1f attn %.1f us" % (k["pack_k_us"], k["pack_v_us"], k["prefill_attention_us"]))
    else:
        print(c.get("ctx"), c.get("bits"), c.get("layers"), "| %.3f ms %.1f tok/s" % (d["ms_per_step"], d["value"]), {kk: round(v, 1) for kk, v in k.items() if isinstance(v, float)})
PY

#!/bin/bash
# round 4, call 24: decode query RoPE in one launch + rotary tables once per token: tests, then the full-model leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c24
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_llama_gpu.py tests/test_decode_kv_gpu.py tests/test_compact_gpu.py -m gpu -q -x > ${O}_tests.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench.json 2> ${O}_err.txt
tail -4 ${O}_tests.txt; python -c "
import json; d=json.load(open('${O}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['full_model'])"

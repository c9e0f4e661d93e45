#!/bin/bash
# round 5, call 24: tests of the prefill pack with two tokens per workgroup + config-4 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_k
( timeout 1200 python -m pytest tests -m gpu -q -x -k "pack or prefill or atsize or ties or decode_kv or cache or attention or llama" 2>&1 | tail -6 ) > ${O}_tests.txt; cat ${O}_tests.txt
for bits in 4 3 2; do python bench.py --prefill --bits $bits 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('bits $bits | pack K %.1f us  V %.1f us  attention %.1f us | %.2f M tokens/s' % (k['pack_k_us'], k['pack_v_us'], k['prefill_attention_us'], d['value'] / 1e6))
"; done | tee ${O}_prefill.txt
python tools/prefill_bench.py 8192 4 2>&1 | tail -8

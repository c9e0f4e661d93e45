#!/bin/bash
# round 5, call 25: prefill pack with 1 token per workgroup against the default (2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_l
for rep in 1 2; do
for lib in default pk11 pk12; do
for bits in 4 3; do
  if [ $lib = default ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$lib.so; fi
  timeout 300 python bench.py --prefill --bits $bits 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('%-8s bits $bits | pack K %.1f us  V %.1f us  attention %.1f us' % ('$lib', k['pack_k_us'], k['pack_v_us'], k['prefill_attention_us']))
" >> ${O}_pack_tt.txt
done; done; done
cat ${O}_pack_tt.txt

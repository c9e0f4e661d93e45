#!/bin/bash
# round 6: prefill attention with the online softmax per 32-key block, branch-free tile bodies -- parity, A/B against the previous library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_prefill_attn_gpu.py tests/test_attention_gpu.py -x -q -m gpu 2>&1 | tail -4
out=gpurun_out/r06_af_attn_ab.txt; : > $out
for rep in 1 2 3; do
for lib in "" tools/abl/libkvq_prev.so; do
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('lib=$lib', {k: round(v, 1) for k, v in d.get('kernels', {}).items() if 'attention' in k})" >> $out
done
done
cat $out

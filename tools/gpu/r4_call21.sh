#!/bin/bash
# round 4, call 21: fused kernel vs the separate kernels at 1M for nuq3 + 5 sinks and nuq2 (same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c21
export TMPDIR=/tmp
for cfg in "--bits 3 --sinks 5" "--bits 2" "--bits 4"; do for v in 0 1; do
  KVQ_FUSED_ATTEND=$v timeout 400 python bench.py --ctx 1048576 --layers 8 $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('1M $cfg fused=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

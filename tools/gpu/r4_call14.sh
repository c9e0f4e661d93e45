#!/bin/bash
# round 4, GPU call 14: fused decode kernel vs the separate kernels where the grid has several generations (256K, 1M)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c14
export TMPDIR=/tmp
for cfg in "--ctx 262144 --layers 16" "--ctx 1048576 --layers 8"; do for v in 0 1; do
  KVQ_FUSED_ATTEND=$v timeout 400 python bench.py $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg fused=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

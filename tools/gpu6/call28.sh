#!/bin/bash
# round 6: score kernel, second-generation workgroups of a CU delayed before their head loop (KVQ_SCORE_STAGGER x 1024 cycles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_ad_stagger.txt; : > $out
for rep in 1 2; do
for st in 0 1 2 3 4; do
  KVQ_SCORE_STAGGER=$st timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('stagger $st | ms/step %.3f' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" >> $out
done
done
cat $out

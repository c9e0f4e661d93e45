#!/bin/bash
# round 6: 128- vs 256-token score tiles between 16K and 64K cached tokens (KVQ_SCORE_T8_FROM), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_aa_tile_ab.txt; : > $out
for rep in 1 2; do
for from in 16384 1000000; do
  for args in "--ctx 16384" "--ctx 24576" "--ctx 32768" "--ctx 49152" "--ctx 65536" "--ctx 32768 --bits 3 --sinks 5"; do
    KVQ_SCORE_T8_FROM=$from timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('T8_FROM=$from $args | ms/step %.3f' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" >> $out
  done
done
done
cat $out

#!/bin/bash
# round 5, call 10: how long do the riding appends take on an otherwise idle GPU (tiny context)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c10
export TMPDIR=/tmp
for ride in 1 0; do
  for cfg in "--ctx 256 --steps 20" "--ctx 1024 --steps 20" "--ctx 8192 --steps 20" "--ctx 16384 --steps 20" "--ctx 65536 --steps 10"; do
  KVQ_DECODE_RIDE=$ride timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg ride=$ride: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

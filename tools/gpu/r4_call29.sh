#!/bin/bash
# round 4, call 29: batched evaluation of the tail outlier entries, batches of 21 (new) vs 7 (evb7) vs HEAD (tb7 = sequential steps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c29
export TMPDIR=/tmp
for rep in 1 2 3; do for v in new evb7 tb7; do
  if [ $v = new ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  for cfg in "--ctx 4096 --steps 20" "--ctx 32768 --steps 20" "--ctx 131072 --steps 10"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

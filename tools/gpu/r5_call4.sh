#!/bin/bash
# round 5, call 4: p.V outlier phase variants, same box: old (round 4), C = windows of 32 (default build), A = all slots RB 24, B = all slots RB 32, D = windows of 36
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c4
export TMPDIR=/tmp
for rep in 1 2; do for v in old C A B D; do
  if [ $v = C ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_pv$v.so; fi
  for cfg in "--ctx 131072 --steps 10" "--ctx 32768 --steps 20" "--ctx 131072 --bits 3 --sinks 5 --steps 10"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg v=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

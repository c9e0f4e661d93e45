#!/bin/bash
# round 4, call 33: p.V plan -- workgroups the token ranges are cut for (512 = default), nuq3 + sinks and nuq4 at 32K / 128K
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c33
export TMPDIR=/tmp
for rep in 1 2; do for v in 512 256 384 1024; do
  if [ $v = 512 ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_wgs$v.so; fi
  for cfg in "--ctx 32768 --bits 3 --sinks 5 --steps 20" "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --steps 20" "--ctx 4096 --steps 20"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg wgs=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

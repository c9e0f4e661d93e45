#!/bin/bash
# round 5, call 8: ride-along appends with 4 histogram copies and 512-lane workgroups at every length
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c8
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_decode_kv_gpu.py tests/test_ties_gpu.py tests/test_fused_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > ${O}_tests.txt
for rep in 1 2; do for ride in 1 0; do
  for cfg in "--ctx 131072 --steps 10" "--ctx 32768 --steps 20" "--ctx 4096 --steps 20"; do
  KVQ_DECODE_RIDE=$ride timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg ride=$ride: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_tests.txt ${O}_ab.txt

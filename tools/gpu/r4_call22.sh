#!/bin/bash
# round 4, call 22: p.V outlier phase at 4 bit in ONE round trip per 512-token range (42 entries per lane) vs two (24), same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c22
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_atsize_gpu.py -m gpu -q -x > ${O}_tests.txt 2>&1
for rep in 1 2 3; do for v in 42 24 32; do
  if [ $v = 42 ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_rb$v.so; fi
  for cfg in "--ctx 131072" "--ctx 32768" "--ctx 1048576 --layers 8 --steps 5"; do
  timeout 300 python bench.py --steps 10 --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg rb=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
tail -3 ${O}_tests.txt; cat ${O}_ab.txt

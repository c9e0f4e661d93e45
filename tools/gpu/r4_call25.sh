#!/bin/bash
# round 4, call 25: the tail outlier entries of a small head group in ONE round trip (TB = 21) vs batches of 7, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c28
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_decode_gpu.py -m gpu -q -x > ${O}_tests.txt 2>&1
for rep in 1 2 3; do for v in 21 7; do  # (21 = the new build, 7 = HEAD)
  if [ $v = 21 ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_tb7.so; fi
  for cfg in "--ctx 4096 --steps 20" "--ctx 32768 --steps 20" "--ctx 131072 --steps 10" "--ctx 131072 --bits 3 --sinks 5 --steps 10"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg tb=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
tail -3 ${O}_tests.txt; cat ${O}_ab.txt

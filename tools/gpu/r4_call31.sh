#!/bin/bash
# round 4, call 31: fused-softmax p.V with the outlier phase of odd workgroups BEFORE their dense loop (new) vs every phase at the end (pvend = HEAD)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c31
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_atsize_gpu.py tests/test_fuzz_gpu.py tests/test_compact_gpu.py -m gpu -q -x > ${O}_tests.txt 2>&1
for rep in 1 2 3; do for v in new pvend; do
  if [ $v = new ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  for cfg in "--ctx 131072 --steps 10" "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --steps 20" "--ctx 4096 --steps 20"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
tail -3 ${O}_tests.txt; cat ${O}_ab.txt

#!/bin/bash
# round 5, call 11: p.V outlier entries split by tokens between the unit groups (4 bit, long caches): parity + A/B (KVQ_V_SPLIT=0 = rows phase)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c11
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_atsize_gpu.py tests/test_decode_kv_gpu.py tests/test_ref_gpu.py tests/test_compact_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > ${O}_tests.txt
for rep in 1 2; do for sp in 1 0; do
  for cfg in "--ctx 131072 --steps 10" "--ctx 262144 --layers 16 --steps 10" "--ctx 131072 --compact --steps 10"; do
  KVQ_V_SPLIT=$sp timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg split=$sp: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_tests.txt ${O}_ab.txt

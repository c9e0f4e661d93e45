#!/bin/bash
# round 4, GPU call 6: outlier entries as trailing workgroups of the p.V launch (no phase inside the dense workgroups)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c6
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mix_affine_gpu.py tests/test_decode_kv_gpu.py -q -x 2>&1 | tail -8 ) > ${O}_tests.txt
cat ${O}_tests.txt
for cfg in "--ctx 131072" "--ctx 32768" "--ctx 4096" "--ctx 131072 --bits 3 --sinks 5"; do for v in A A4 rows; do
  export KVQ_VA_DBG=0 KVQ_MIX_ROWS=0 KVQ_VA_CFG=A
  if [ $v = rows ]; then export KVQ_MIX_ROWS=1; fi
  if [ $v = A4 ]; then export KVQ_VA_DBG=4; fi
  timeout 600 python bench.py $cfg --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg cfg=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['frac'])"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

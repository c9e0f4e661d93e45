#!/bin/bash
# round 4, GPU call 7: first run of the fused decode kernel (kvq_fused_attend): correctness vs the separate kernels and the
# reference pipeline, then timing (KVQ_FUSED_ATTEND=1) against the separate kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c7
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fused_decode_gpu.py -x -q 2>&1 | tail -25 ) > ${O}_tests.txt
cat ${O}_tests.txt
for cfg in "--ctx 131072" "--ctx 32768" "--ctx 131072 --bits 3 --sinks 5"; do for v in 0 1; do
  KVQ_FUSED_ATTEND=$v timeout 600 python bench.py $cfg --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg fused=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['frac'])"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

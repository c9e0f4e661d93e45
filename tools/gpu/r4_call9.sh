#!/bin/bash
# round 4, GPU call 9: fused decode kernel with head parts (K(16) V(16) K(16) V(16)) and a start stagger of odd workgroups
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c9
export TMPDIR=/tmp
( KVQ_FUSED_PART=16 KVQ_FUSED_STAGGER=4 timeout 900 python -m pytest tests/test_fused_decode_gpu.py -q -x 2>&1 | tail -5 ) > ${O}_tests.txt
cat ${O}_tests.txt
for ps in "0 0" "0 5" "0 10" "16 0" "16 2" "16 4" "16 6" "16 9" "16 13"; do set -- $ps
  KVQ_FUSED_ATTEND=1 KVQ_FUSED_PART=$1 KVQ_FUSED_STAGGER=$2 timeout 300 python bench.py --ctx 131072 --steps 12 --warmup 3 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('part=$1 stagger=$2: %.3f ms/step' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

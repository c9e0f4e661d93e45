#!/bin/bash
# round 3, GPU call 30: number of p.V workgroups (KVQ_V_WGS: 512 = one generation of two per CU) with the new outlier phase
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c30
export TMPDIR=/tmp
for cfg in "--ctx 131072" "--ctx 32768 --steps 20" "--ctx 131072 --bits 3 --sinks 5"; do for v in "" w384 w768 w1024; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  timeout 900 python bench.py $cfg --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg lib=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_wgs.txt 2>&1
cat ${O}_wgs.txt

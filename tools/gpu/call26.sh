#!/bin/bash
# round 3, GPU call 26: softmax partials merged by a small kernel (default beyond 256 tiles) vs inside every p.V workgroup
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c26
export TMPDIR=/tmp
for rep in 1 2; do for cfg in "--ctx 131072" "--ctx 262144 --layers 16" "--ctx 1048576 --layers 8 --steps 5"; do for v in "" mp1024 mp4096; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  timeout 900 python bench.py $cfg --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg lib=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done; done > ${O}_merge.txt 2>&1
cat ${O}_merge.txt

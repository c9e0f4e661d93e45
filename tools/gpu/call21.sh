#!/bin/bash
# round 3, GPU call 21: where does the fused softmax (p.V normalises the raw scores itself: one launch less) stop paying?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c21
export TMPDIR=/tmp
for ctx in 16384 32768 49152 65536 98304; do for lim in 0 1000000; do
  KVQ_FUSE_SOFTMAX_UP_TO=$lim timeout 600 python bench.py --ctx $ctx --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ctx $ctx fuse_up_to $lim: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_fuse.txt 2>&1
cat ${O}_fuse.txt

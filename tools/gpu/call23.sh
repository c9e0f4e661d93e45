#!/bin/bash
# round 3, GPU call 23: fused softmax (p.V normalises the raw scores) vs the separate softmax launch at long contexts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c23
export TMPDIR=/tmp
for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 131072 --compact" "--ctx 262144 --layers 16" "--ctx 1048576 --layers 8 --steps 5"; do for lim in 0 100000000; do
  KVQ_FUSE_SOFTMAX_UP_TO=$lim timeout 900 python bench.py $cfg --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg fuse_up_to $lim: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_fuse.txt 2>&1
cat ${O}_fuse.txt

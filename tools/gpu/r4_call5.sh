#!/bin/bash
# round 4, GPU call 5: ablations of the affine p.V kernel (KVQ_VA_DBG: 1 no look-ups, 2 no DMA, 4 no outlier phase, 8 no
# table build) for variants A and D; replay of the reference attention fixtures with the prefill tolerance explained
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c5
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_attention_gpu.py -q 2>&1 | tail -8 ) > ${O}_tests.txt
cat ${O}_tests.txt
for v in A D; do for d in 0 1 2 4 5 6 7 12; do
  KVQ_VA_CFG=$v KVQ_VA_DBG=$d timeout 300 python bench.py --ctx 131072 --steps 12 --warmup 3 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg=$v dbg=$d: %.3f ms/step' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done; done > ${O}_abl.txt 2>&1
cat ${O}_abl.txt

#!/bin/bash
# round 6, call 14: the wide p.V kernel merging the softmax partials itself at 128K / 256K (no merge launch) vs the merge kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3"
for rep in 1 2 3; do
for m in 256 1024; do
  for cfg in "--ctx 131072 --steps 10" "--ctx 262144 --layers 16 --steps 6" "--ctx 65536 --steps 10"; do
    KVQ_W_MERGE_PARTS_RT=$m timeout 150 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('merge_parts=$m $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_n_merge.txt 2>&1
cat gpurun_out/r06_n_merge.txt
timeout 600 python -m pytest tests/test_atsize_gpu.py tests/test_decode_kv_gpu.py -x -q -m gpu 2>&1 | tail -2

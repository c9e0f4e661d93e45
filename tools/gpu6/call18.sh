#!/bin/bash
# round 6, call 18: in-kernel softmax merge of the wide p.V at 128K with all of a lane's partial loads in one round trip
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3"
for rep in 1 2 3; do
for v in "default 256" "default 1024" "w_mb16 1024" "w_mb32 1024"; do
  set -- $v
  lib=kvquant_amd/libkvq.so; [ $1 != default ] && lib=tools/abl/libkvq_$1.so
  for cfg in "--ctx 131072 --steps 10" "--ctx 262144 --layers 16 --steps 6"; do
    KVQ_LIB=$lib KVQ_W_MERGE_PARTS_RT=$2 timeout 150 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$1 merge_parts=$2 $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_r_merge_batch.txt 2>&1
cat gpurun_out/r06_r_merge_batch.txt

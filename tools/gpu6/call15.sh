#!/bin/bash
# round 6, call 15: wave priorities that alternate between the two quads of a sub-stage (every wave favoured half of the time)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3 --steps 10"
for rep in 1 2 3; do
for v in default w_prio3; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768"; do
    lib=kvquant_amd/libkvq.so; [ $v != default ] && lib=tools/abl/libkvq_$v.so
    KVQ_LIB=$lib timeout 120 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
echo "## trace, alternating priorities"
BITS=4 KVQ_LIB=tools/abl/libkvq_w_prio3t.so timeout 120 python tools/dbg/trace_vw.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_o_prio3.txt 2>&1
cat gpurun_out/r06_o_prio3.txt

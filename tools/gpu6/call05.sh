#!/bin/bash
# round 6, call 5: wave priorities by token slot in the wide p.V kernel + trimmed entry arithmetic (same box A/B)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --warmup 3 --steps 10"
for rep in 1 2; do
for v in default w_prio1 w_prio2; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768"; do
    lib=kvquant_amd/libkvq.so; [ $v != default ] && lib=tools/abl/libkvq_$v.so
    KVQ_V_WIDE_FROM=1 KVQ_LIB=$lib python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
echo "## trace, priority = slot"
BITS=4 KVQ_LIB=tools/abl/libkvq_w_trace_prio1.so python tools/dbg/trace_vw.py 2>&1 | grep -v amdgpu.ids
echo "## trace, default"
BITS=4 KVQ_LIB=tools/abl/libkvq_w_trace.so python tools/dbg/trace_vw.py 2>&1 | grep -v amdgpu.ids | head -12
} > gpurun_out/r06_e_wide_prio.txt 2>&1
cat gpurun_out/r06_e_wide_prio.txt

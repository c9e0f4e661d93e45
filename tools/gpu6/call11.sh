#!/bin/bash
# round 6, call 11: 1M tokens -- fused split-L kernel vs the kernel pair with the wide p.V; wide p.V at 4K / 8K with the in-kernel
# merge; trace of the wide kernel as it stands
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 2"
for rep in 1 2; do
for f in 1 0; do
  for cfg in "--ctx 1048576 --layers 8 --steps 4" "--ctx 524288 --layers 16 --steps 4" "--ctx 262144 --layers 16 --steps 6"; do
    KVQ_FUSED_ATTEND=$f timeout 200 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('fused=$f $cfg | ms/step %.3f' % d['ms_per_step'], {a: round(b,1) for a,b in k.items() if a.endswith('_us')})
"
  done
done
done
for rep in 1 2; do
for w in 0 1; do
  for cfg in "--ctx 4096 --steps 20" "--ctx 8192 --steps 20" "--ctx 12288 --steps 20" "--ctx 8192 --bits 3 --sinks 5 --steps 20"; do
    KVQ_V_WIDE=$w KVQ_V_WIDE_FROM=1 timeout 100 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('wide=$w $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
echo "## trace of the wide kernel (padded probability rows, entries + conversion in the loop)"
for b in 4 3; do echo "== bits $b"; BITS=$b KVQ_LIB=tools/abl/libkvq_w_trace.so timeout 120 python tools/dbg/trace_vw.py 2>&1 | grep -v amdgpu.ids | head -12; done
} > gpurun_out/r06_k_misc.txt 2>&1
cat gpurun_out/r06_k_misc.txt

#!/bin/bash
# round 6, call 6: from which length does the wide p.V geometry win? (same box, alternating) + the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --warmup 3 --steps 20"
for rep in 1 2; do
for ctx in 2048 4096 8192 16384 32768 49152; do
  for w in 0 1; do
   for cfg in "--ctx $ctx" "--ctx $ctx --bits 3 --sinks 5"; do
    KVQ_V_WIDE=$w KVQ_V_WIDE_FROM=1 python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('wide=$w $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
   done
  done
done
done
} > gpurun_out/r06_f_wide_from.txt 2>&1
cat gpurun_out/r06_f_wide_from.txt
KVQ_V_WIDE_FROM=4096 timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_f_gpu_tests.txt 2>&1
tail -5 gpurun_out/r06_f_gpu_tests.txt

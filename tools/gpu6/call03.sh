#!/bin/bash
# round 6, call 3: where does the wide p.V kernel spend its time?  trace + ablations + burst issue (same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "## trace (KVQ_TRACE build), 128K + 77 tokens"
for b in 4 3; do echo "== bits $b"; BITS=$b KVQ_LIB=tools/abl/libkvq_w_trace.so python tools/dbg/trace_vw.py; done
echo "## variants: bench.py lines (mix_v_us = p.V kernel + slab reduce between in-stream events)"
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --warmup 3 --steps 10"
for rep in 1 2; do
for v in default w_burst w_noent w_nomath w_nodma w_noentmath; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5"; do
    lib=kvquant_amd/libkvq.so; [ $v != default ] && lib=tools/abl/libkvq_$v.so
    KVQ_LIB=$lib python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_c_wide_trace.txt 2>&1
cat gpurun_out/r06_c_wide_trace.txt
bash tools/profile_bench.sh r06_c --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model > /dev/null 2>&1
cat gpurun_out/r06_c_kernel_stats.csv

#!/bin/bash
# round 5, call 18: p.V plan (workgroups aimed at) per bit width and length
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_e
B="--warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model"
run() { # label env... -- bench args
  local label="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('%-22s %-50s | ms/step %.3f score_k %.1f mix_v %.1f' % ('$label', '$*', d['ms_per_step'], k['score_k_us'], k['mix_v_us']))
" >> ${O}_pv_plan.txt
}
for rep in 1 2; do
for w in 512 256; do
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 131072 --bits 3 --sinks 5 --steps 10
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 131072 --bits 3 --sinks 0 --steps 10
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 32768 --bits 3 --sinks 5 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 4096 --bits 3 --sinks 5 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 32768 --bits 4 --sinks 0 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 4096 --bits 4 --sinks 0 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 32768 --bits 2 --sinks 0 --steps 20
done
run "wgs 256 + merge k." KVQ_V_WGS_RT=256 KVQ_V_MERGE_PARTS_RT=0 -- --ctx 32768 --bits 3 --sinks 0 --steps 20
run "wgs 320" KVQ_V_WGS_RT=320 -- --ctx 131072 --bits 3 --sinks 5 --steps 10
done
cat ${O}_pv_plan.txt

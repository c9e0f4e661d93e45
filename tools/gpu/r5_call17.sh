#!/bin/bash
# round 5, call 17: p.V with sink tokens through the merge kernel at every length; 3-bit plan at short lengths
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_d
B="--steps 20 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model"
run() { # label env... -- bench args
  local label="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('%-28s %-44s | ms/step %.3f score_k %.1f mix_v %.1f' % ('$label', '$*', d['ms_per_step'], k['score_k_us'], k['mix_v_us']))
" >> ${O}_sink_merge.txt
}
for rep in 1 2; do
run "default(merge kernel)" X=1 -- --ctx 32768 --bits 4 --sinks 5
run "in-kernel merge" KVQ_V_MERGE_PARTS_RT=256 -- --ctx 32768 --bits 4 --sinks 5
run "default(merge kernel)" X=1 -- --ctx 32768 --bits 3 --sinks 5
run "in-kernel merge" KVQ_V_MERGE_PARTS_RT=256 -- --ctx 32768 --bits 3 --sinks 5
run "default" X=1 -- --ctx 32768 --bits 3 --sinks 0
run "merge kernel" KVQ_V_MERGE_PARTS_RT=0 -- --ctx 32768 --bits 3 --sinks 0
run "256 wgs" KVQ_V_WGS_RT=256 -- --ctx 32768 --bits 3 --sinks 0
run "384 wgs" KVQ_V_WGS_RT=384 -- --ctx 32768 --bits 3 --sinks 0
run "default" X=1 -- --ctx 32768 --bits 4 --sinks 0
run "merge kernel" KVQ_V_MERGE_PARTS_RT=0 -- --ctx 32768 --bits 4 --sinks 0
run "default" X=1 -- --ctx 4096 --bits 3 --sinks 5
run "in-kernel merge" KVQ_V_MERGE_PARTS_RT=256 -- --ctx 4096 --bits 3 --sinks 5
done
cat ${O}_sink_merge.txt

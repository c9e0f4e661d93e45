#!/bin/bash
# round 5, call 19: the 3-bit plan's crossover (256 vs 512 workgroups) + p.V tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_f
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
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 65536 --bits 3 --sinks 5 --steps 10
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 49152 --bits 3 --sinks 5 --steps 10
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 16384 --bits 3 --sinks 5 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 8192 --bits 3 --sinks 5 --steps 20
done
run "default" X=1 -- --ctx 32768 --bits 3 --sinks 5 --steps 20
run "default" X=1 -- --ctx 32768 --bits 2 --sinks 0 --steps 20
done
cat ${O}_pv_plan.txt
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_decode_gpu.py tests/test_atsize_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > ${O}_tests.txt; cat ${O}_tests.txt

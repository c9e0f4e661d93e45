#!/bin/bash
# round 5, call 29: 3 / 2 bit p.V as 256-lane workgroups (four per CU): tests + config 3 / nuq2 bench lines per plan
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_p
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_decode_gpu.py tests/test_compact_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > ${O}_tests.txt; cat ${O}_tests.txt
B="--warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model"
run() { local label="$1"; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('%-12s %-46s | ms/step %.3f score_k %.1f mix_v %.1f' % ('$label', '$*', d['ms_per_step'], k['score_k_us'], k['mix_v_us']))
" >> ${O}_v256.txt
}
for rep in 1 2; do
for w in 1024 512; do
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 131072 --bits 3 --sinks 5 --steps 10
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 32768 --bits 3 --sinks 5 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 4096 --bits 3 --sinks 5 --steps 20
run "wgs $w" KVQ_V_WGS_RT=$w -- --ctx 32768 --bits 2 --steps 20
done; done
cat ${O}_v256.txt

#!/bin/bash
# round 6, call 16: exact fp32 pair-sum K tables at 3 bit (1024-lane score workgroup): parity, then A/B against the per-channel tables
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_atsize_gpu.py tests/test_decode_kv_gpu.py tests/test_cache_gpu.py tests/test_attention_gpu.py tests/test_fused_decode_gpu.py tests/test_compact_gpu.py -x -q -m gpu > gpurun_out/r06_p_tests.txt 2>&1
tail -6 gpurun_out/r06_p_tests.txt
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3"
for rep in 1 2 3; do
for v in 0 1; do
  for cfg in "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --bits 3 --sinks 5 --steps 20" "--ctx 131072 --bits 3 --compact --steps 10" "--ctx 1048576 --bits 3 --sinks 5 --layers 8 --steps 4"; do
    KVQ_SCORE_F32_PAIR=$v timeout 200 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('f32pair=$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_p_pair32_ab.txt 2>&1
cat gpurun_out/r06_p_pair32_ab.txt

#!/bin/bash
# round 6, call 10: q of the score kernel's outlier step as one ds_read_b64 (same-box A/B against the two-read form) + parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_atsize_gpu.py tests/test_fused_decode_gpu.py -x -q -m gpu > gpurun_out/r06_j_tests.txt 2>&1
tail -3 gpurun_out/r06_j_tests.txt
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3 --steps 10"
for rep in 1 2 3; do
for v in default ql0; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768" "--ctx 4096 --steps 20"; do
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
} > gpurun_out/r06_j_ql_ab.txt 2>&1
cat gpurun_out/r06_j_ql_ab.txt

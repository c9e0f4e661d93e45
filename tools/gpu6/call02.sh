#!/bin/bash
# round 6, call 2: the 1024-lane p.V kernel (kvq_mix_v_wide.hip): parity with the wide geometry forced at every length, then its speed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KVQ_V_WIDE_FROM=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_compact_gpu.py tests/test_ties_gpu.py tests/test_cache_gpu.py -x -q -m gpu > gpurun_out/r06_b_wide_forced_tests.txt 2>&1
tail -15 gpurun_out/r06_b_wide_forced_tests.txt
timeout 900 python -m pytest tests/test_atsize_gpu.py -x -q -m gpu > gpurun_out/r06_b_atsize_tests.txt 2>&1
tail -5 gpurun_out/r06_b_atsize_tests.txt
B="--no-cpu-baseline --no-fp16-baseline --no-full-model"
for v in 0 1 0 1; do
  for cfg in "--ctx 131072 --steps 10" "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 65536 --steps 10" "--ctx 32768 --steps 20"; do
    KVQ_V_WIDE=$v KVQ_V_WIDE_FROM=1 python bench.py $cfg --warmup 3 $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('wide=$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f frac %.3f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0), d['roofline']['frac']))
"
  done
done | tee gpurun_out/r06_b_wide_ab.txt

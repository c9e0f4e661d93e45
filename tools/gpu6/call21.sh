#!/bin/bash
# round 6: pruning select (kvq_select.h) in the decode append and the prefill pack -- parity suites, then A/B against the radix
# select (tools/abl/libkvq_radix.so = the same tree with -DKVQ_FAST_SELECT=0): prefill leg, 4K / 32K / 128K decode
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ties_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_gpu.py tests/test_cache_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_attention_gpu.py tests/test_ref_gpu.py -x -q -m gpu > gpurun_out/r06_u_tests.txt 2>&1; tail -5 gpurun_out/r06_u_tests.txt
out=gpurun_out/r06_u_select_ab.txt; : > $out
for rep in 1 2; do
for lib in "" tools/abl/libkvq_radix.so; do
  echo "== KVQ_LIB=$lib" >> $out
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('prefill', {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})" >> $out
  for ctx in 4096 32768 131072; do
    KVQ_LIB=$lib timeout 300 python bench.py --ctx $ctx --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ctx', d['config']['ctx'], 'ms/step %.3f' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" >> $out
  done
done
done
cat $out

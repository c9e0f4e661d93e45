#!/bin/bash
# round 6: V codes by candidate + verification (nearest_codes_shared_row) -- parity suites, then A/B of the prefill leg and the decode
# step against the previous commit's library (tools/abl/libkvq_prev.so: pruning select, 16-entry scan) and the radix build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_select_gpu.py tests/test_ties_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_gpu.py tests/test_cache_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_attention_gpu.py tests/test_ref_gpu.py tests/test_compact_gpu.py tests/test_head_shard_gpu.py -x -q -m gpu > gpurun_out/r06_ab_tests.txt 2>&1; tail -3 gpurun_out/r06_ab_tests.txt
out=gpurun_out/r06_ab_vcodes_ab.txt; : > $out
for rep in 1 2; do
for lib in "" tools/abl/libkvq_prev.so tools/abl/libkvq_radix.so; do
  echo "== KVQ_LIB=$lib" >> $out
  for b in 4 3; do
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --bits $b --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('prefill bits $b', {k: round(v, 1) for k, v in d.get('kernels', {}).items() if 'pack' in k})" >> $out
  done
  for ctx in 4096 131072; do
    KVQ_LIB=$lib timeout 300 python bench.py --ctx $ctx --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ctx', d['config']['ctx'], 'ms/step %.3f' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" >> $out
  done
done
done
cat $out

#!/bin/bash
# round 6: V codes by binary-search candidate + verification -- parity subset, A/B of the prefill leg against libkvq_prev.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_select_gpu.py tests/test_ties_gpu.py tests/test_decode_kv_gpu.py tests/test_cache_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_attention_gpu.py -x -q -m gpu > gpurun_out/r06_ac_tests.txt 2>&1; tail -3 gpurun_out/r06_ac_tests.txt
out=gpurun_out/r06_ac_vcodes_ab.txt; : > $out
for rep in 1 2 3; do
for lib in "" tools/abl/libkvq_prev.so; do
  echo "== KVQ_LIB=$lib" >> $out
  for b in 4 3 2; do
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --bits $b --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('prefill bits $b', {k: round(v, 1) for k, v in d.get('kernels', {}).items() if 'pack' in k and 'us' in k})" >> $out
  done
done
done
cat $out

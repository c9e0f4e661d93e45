#!/bin/bash
# round 6: pruning select with the bound exchange -- parity suites, phase trace, A/B against the radix select
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ties_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_gpu.py tests/test_cache_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_attention_gpu.py tests/test_ref_gpu.py -x -q -m gpu > gpurun_out/r06_w_tests.txt 2>&1; tail -3 gpurun_out/r06_w_tests.txt
out=gpurun_out/r06_w_select_ab.txt; : > $out
echo "== ptrace" >> $out; KVQ_LIB=tools/abl/libkvq_ptrace.so timeout 200 python tools/dbg/trace_prologue.py 2>/dev/null >> $out
for rep in 1 2; do
for lib in "" tools/abl/libkvq_radix.so; do
  echo "== KVQ_LIB=$lib" >> $out
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('prefill', {k: round(v, 1) for k, v in d.get('kernels', {}).items() if 'pack' in k})" >> $out
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

#!/bin/bash
# round 6: V append of the decode prologue split over four workgroups (same selection, a quarter of the codes each) -- parity suites,
# A/B against the previous commit's library (tools/abl/libkvq_prev.so): step times + rocprofv3 kernel stats of the prologue
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_select_gpu.py tests/test_ties_gpu.py tests/test_decode_kv_gpu.py tests/test_cache_gpu.py tests/test_fused_decode_gpu.py tests/test_attention_gpu.py tests/test_head_shard_gpu.py tests/test_compact_gpu.py tests/test_llama_gpu.py -x -q -m gpu > gpurun_out/r06_ae_tests.txt 2>&1; tail -3 gpurun_out/r06_ae_tests.txt
out=gpurun_out/r06_ae_vparts_ab.txt; : > $out
for rep in 1 2; do
for lib in "" tools/abl/libkvq_prev.so; do
  for args in "--ctx 4096" "--ctx 32768" "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5"; do
    KVQ_LIB=$lib timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib=$lib $args | ms/step %.3f' % d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" >> $out
  done
done
done
cat $out
for lib in "" tools/abl/libkvq_prev.so; do
  cd /tmp; rm -rf /tmp/prof_x
  KVQ_LIB=$([ -n "$lib" ] && echo $GRAFT_REPO_ROOT/$lib) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/bench.py --ctx 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > /tmp/prof_x.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats at 4K, lib=$lib" | tee -a $out; grep "prologue\|score_k\|mix_v" "$f" | cut -d, -f1-4 | tee -a $out
done

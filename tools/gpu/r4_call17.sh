#!/bin/bash
# round 4, call 17: 1.25-op nibble extraction in the 4-bit score kernel: A/B against the round-3 extraction (same box,
# alternating), then the whole GPU suite on the new library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out/r4c17
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py -m gpu -q -x > ${O}_quick.txt 2>&1
for rep in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export KVQ_LIB=tools/abl/libkvq_oldcut.so; else unset KVQ_LIB; fi
  for cfg in "--ctx 131072" "--ctx 32768"; do
  timeout 300 python tools/bench_with_lib.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
  done
done; done > ${O}_ab.txt 2>&1
unset KVQ_LIB
timeout 1800 python -m pytest tests -m gpu -q -x > ${O}_tests.txt 2>&1
tail -3 ${O}_quick.txt; cat ${O}_ab.txt; tail -5 ${O}_tests.txt

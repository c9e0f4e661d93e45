#!/bin/bash
# round 5, call 2: p.V outlier phase in one round trip (32-slot windows, branch-free): parity + bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c2
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_ref_gpu.py tests/test_atsize_gpu.py tests/test_compact_gpu.py -m gpu -x -q 2>&1 | tail -8 ) > ${O}_tests.txt
for cfg in "--ctx 131072 --steps 10" "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --steps 20" "--ctx 4096 --steps 20" "--ctx 32768 --bits 3 --sinks 5 --steps 20"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
done > ${O}_bench.txt 2>&1
cat ${O}_tests.txt ${O}_bench.txt

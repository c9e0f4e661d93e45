#!/bin/bash
# round 5, call 6: cheaper exact 3-bit field extraction in the score kernel: parity + bench; the pair-table tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c6
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_atsize_gpu.py tests/test_ref_gpu.py tests/test_ops_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_decode_gpu.py -m gpu -q -s 2>&1 | grep -E "fp16 pair|end to end|passed|failed|Error|error|bits=3" | tail -40 ) > ${O}_tests.txt
for rep in 1 2; do for f16 in "" "--score-f16"; do
  for cfg in "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --bits 3 --sinks 5 --steps 20"; do
  timeout 300 python bench.py --warmup 3 $cfg $f16 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $f16: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_tests.txt ${O}_ab.txt

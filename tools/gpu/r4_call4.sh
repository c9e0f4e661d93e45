#!/bin/bash
# round 4, GPU call 4: affine p.V kernel with the steady-state loop (variants A, C, D) vs the per-row kernel; the replay of
# the reference's LlamaAttention fixtures; the tie tests with the new C-ABI default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c4
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mix_affine_gpu.py tests/test_attention_gpu.py tests/test_ties_gpu.py -q 2>&1 | tail -15 ) > ${O}_tests.txt
cat ${O}_tests.txt
for v in C D; do ( KVQ_VA_CFG=$v timeout 600 python -m pytest tests/test_mix_affine_gpu.py -x -q 2>&1 | tail -2 ); done >> ${O}_tests.txt 2>&1
for cfg in "--ctx 131072" "--ctx 32768" "--ctx 4096" "--ctx 131072 --bits 3 --sinks 5"; do for v in A C D rows; do
  if [ $v = rows ]; then export KVQ_MIX_ROWS=1; else export KVQ_MIX_ROWS=0 KVQ_VA_CFG=$v; fi
  timeout 600 python bench.py $cfg --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg cfg=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['frac'])"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

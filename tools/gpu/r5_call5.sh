#!/bin/bash
# round 5, call 5: 3-bit q.K^T through fp16 pair-sum tables: parity at size + bench A/B (KVQ_SCORE_F32=1 = the fp32 tables)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c5
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_atsize_gpu.py tests/test_decode_kv_gpu.py tests/test_ops_gpu.py -m gpu -x -q -s -k "pair or decode_kv or at_size or mix_v" 2>&1 | grep -v "^$" | tail -30 ) > ${O}_tests.txt
for rep in 1 2; do for f32 in 0 1; do
  for cfg in "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --bits 3 --sinks 5 --steps 20" "--ctx 131072 --steps 10"; do
  KVQ_SCORE_F32=$f32 timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg f32=$f32: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
cat ${O}_tests.txt ${O}_ab.txt

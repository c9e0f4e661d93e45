#!/bin/bash
# round 5, call 1: instruction-rate microbenchmarks (fp16-table / v_dot2 forms of the K look-up step) + the baseline of this box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c1
export TMPDIR=/tmp
( tools/ubench/lut_rate; tools/ubench/valu_rate2 ) > ${O}_ubench.txt 2>&1
for cfg in "--ctx 131072 --steps 10" "--ctx 131072 --bits 3 --sinks 5 --steps 10" "--ctx 32768 --steps 20"; do
  timeout 300 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, round(d['roofline']['frac'], 3))" || tail -3 ${O}_err.txt
done > ${O}_bench.txt 2>&1
cat ${O}_ubench.txt ${O}_bench.txt

#!/bin/bash
# round 3, GPU call 20: what do the timing events cost?  bench.py with every / every 3rd / every 8th layer timed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c20
export TMPDIR=/tmp
for ctx in 4096 32768 131072; do for ev in 1 3 8; do
  timeout 600 python bench.py --ctx $ctx --steps 20 --time-every $ev --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ctx $ctx every $ev: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['timed_launches'])"
done; done > ${O}_events.txt 2>&1
cat ${O}_events.txt

#!/bin/bash
# round 4, GPU call 3: geometry variants of the affine p.V kernel (KVQ_VA_CFG = A: 512 lanes, 16-token chunks, 8 copies, 2 wg/CU;
# B: 256 lanes one slot; C: 32-token chunks, 16 copies, 1 wg/CU; D: 32 copies (v_perm), 2 stages, 1 wg/CU) vs the per-row kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c3
export TMPDIR=/tmp
for v in A B C D; do
  echo "== variant $v"
  ( KVQ_VA_CFG=$v timeout 600 python -m pytest tests/test_mix_affine_gpu.py -x -q 2>&1 | tail -4 )
done > ${O}_tests.txt 2>&1
cat ${O}_tests.txt
for cfg in "--ctx 131072" "--ctx 32768"; do for v in A B C D rows; do
  if [ $v = rows ]; then export KVQ_MIX_ROWS=1; else export KVQ_MIX_ROWS=0 KVQ_VA_CFG=$v; fi
  timeout 600 python bench.py $cfg --steps 20 --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg cfg=$v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')}, d['roofline']['frac'])"
done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt

#!/bin/bash
# round 4, call 18: A/B of the 1.25-op nibble extraction (same box, alternating; KVQ_LIB = the round-3 extraction);
# where the ranks of the 4-process one-GPU smoke run are after 100 s (faulthandler dump)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out/r4c18
for rep in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export KVQ_LIB=tools/abl/libkvq_oldcut.so; else unset KVQ_LIB; fi
  for cfg in "--ctx 131072" "--ctx 32768"; do
  timeout 300 python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg $v: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})" || tail -3 ${O}_err.txt
  done
done; done > ${O}_ab.txt 2>&1
unset KVQ_LIB
export KVQ_BENCH_ONE_GPU=1 KVQ_BENCH_DUMP_AFTER=240
( echo "== layers, 4 ranks"; time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29504 bench.py --gpus 4 --ctx 16384 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*" | tail -60
  echo "== tokens, 4 ranks"; time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29604 bench.py --gpus 4 --shard tokens --ctx 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*" | tail -60 ) > ${O}_multirank.txt 2>&1
cat ${O}_ab.txt; cut -c1-200 ${O}_multirank.txt | tail -120

#!/bin/bash
# round 5, call 13: head / token shards with sinks + Q-Norm, one-call head shard step: tests + the N = 1 cost of --shard heads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5c13
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_head_shard_gpu.py tests/test_sharding_gpu.py tests/test_decode_kv_gpu.py -m gpu -q -x 2>&1 | tail -15 ) > ${O}_tests.txt
for cfg in "--ctx 32768 --steps 20" "--ctx 32768 --steps 20 --shard heads" "--ctx 32768 --steps 20 --bits 3 --sinks 5" "--ctx 32768 --steps 20 --bits 3 --sinks 5 --shard heads" "--ctx 131072 --steps 10 --shard heads"; do
  timeout 600 python bench.py --warmup 3 $cfg --no-cpu-baseline --no-fp16-baseline --no-full-model 2>${O}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']))" || tail -5 ${O}_err.txt
done > ${O}_bench.txt 2>&1
cat ${O}_tests.txt ${O}_bench.txt

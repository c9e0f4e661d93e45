#!/bin/bash
# round 4, call 19: head-sharded cache -- tests, and the rank logic of bench.py --shard heads on the one visible GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out/r4c19
timeout 900 python -m pytest tests/test_head_shard_gpu.py tests/test_sharding_gpu.py -m gpu -q -x > ${O}_tests.txt 2>&1
timeout 300 python bench.py --shard heads --ctx 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline > ${O}_w1.json 2> ${O}_w1.err
export KVQ_BENCH_ONE_GPU=1 KVQ_BENCH_DUMP_AFTER=240
( echo "== heads, 2 ranks"; time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29702 bench.py --gpus 2 --shard heads --ctx 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*\|socket.cpp\|^\[Gloo\]" | tail -30
  echo "== heads, 4 ranks"; time timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29704 bench.py --gpus 4 --shard heads --ctx 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*\|socket.cpp\|^\[Gloo\]" | tail -30 ) > ${O}_multirank.txt 2>&1
tail -30 ${O}_tests.txt; cut -c1-600 ${O}_w1.json; tail -5 ${O}_w1.err; cut -c1-700 ${O}_multirank.txt

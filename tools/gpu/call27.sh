#!/bin/bash
# round 3, GPU call 27: the multi-rank code paths of bench.py on the one visible GPU (every rank on cuda:0, gloo): smoke run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c27
export TMPDIR=/tmp KVQ_BENCH_ONE_GPU=1
for n in 2 4; do
  echo "== layers, $n ranks"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --ctx 16384 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*" | tail -3
  echo "== tokens, $n ranks"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --shard tokens --ctx 32768 --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-baseline 2>&1 | grep -v "amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*" | tail -3
done > ${O}_multirank.txt 2>&1
cut -c1-900 ${O}_multirank.txt

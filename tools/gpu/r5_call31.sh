#!/bin/bash
# round 5, call 31: the multi-rank code paths of bench.py on the one visible GPU (KVQ_BENCH_ONE_GPU=1: every rank on cuda:0, hand-overs
# over gloo) -- a smoke run of the rank logic after this round's bench changes, not a measurement
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_r_multirank_smoke.txt
export KVQ_BENCH_ONE_GPU=1 KVQ_BENCH_DUMP_AFTER=240
B="--ctx 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model"
for mode in "--layers 4" "--layers 4 --shard tokens" "--layers 4 --shard heads" "--layers 4 --shard heads --bits 3 --sinks 5"; do
  for n in 2 4; do
    echo "== bench.py --gpus $n $mode" >> $O
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n $B $mode 2>&1 | grep -E "^\{|Error|error|Traceback" | cut -c1-420 >> $O
  done
done
cat $O

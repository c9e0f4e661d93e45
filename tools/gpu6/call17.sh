#!/bin/bash
# round 6, call 17: multi-rank code paths of bench.py on the one GPU (smoke), with timeouts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "# round 6: multi-rank code paths of bench.py, all ranks on the ONE visible GPU (KVQ_BENCH_ONE_GPU=1, hand-overs over gloo): a smoke run of the rank logic after this round's changes (wide p.V kernel, configs legs off for N > 1); the times are those of 2 - 4 processes time-slicing one GPU and mean nothing"
p=29517
for cfg in "--gpus 2 --layers 4" "--gpus 4 --layers 4" "--gpus 2 --layers 4 --shard tokens" "--gpus 2 --layers 4 --shard heads" "--gpus 2 --layers 4 --shard heads --bits 3 --sinks 5"; do
  echo "== bench.py $cfg --ctx 16384"
  n=$(echo $cfg | awk '{print $2}')
  p=$((p+1))
  KVQ_BENCH_ONE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $p bench.py $cfg --ctx 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | grep '^{' | cut -c1-420
  echo "rc ${PIPESTATUS[0]}"
done
} > gpurun_out/r06_q_multirank_smoke.txt 2>&1
cut -c1-260 gpurun_out/r06_q_multirank_smoke.txt

cd $GRAFT_REPO_ROOT; export KVQ_BENCH_ONE_GPU=1 KVQ_BENCH_DUMP_AFTER=240
B="--ctx 4096 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --layers 4 --shard heads"
for cfg in "--bits 3 --sinks 5 --steps 3" "--bits 3 --sinks 5 --steps 20" "--bits 3 --sinks 0 --steps 20" "--bits 4 --sinks 5 --steps 20"; do
  echo "== $cfg"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 $B $cfg 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ms/step %.3f  %.1f tok/s' % (d['ms_per_step'], d['value']))"
done

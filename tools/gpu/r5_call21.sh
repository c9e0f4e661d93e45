#!/bin/bash
# round 5, call 21: prologue with sink tokens (one wave per sink score) and without the unused 3-bit pair image
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_h
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_decode_kv_gpu.py tests/test_fused_decode_gpu.py tests/test_head_shard_gpu.py tests/test_atsize_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > ${O}_tests.txt; cat ${O}_tests.txt
for c in 4096 32768 131072; do
  bash tools/profile_bench.sh r05_h_cfg3_ctx${c} --ctx $c --bits 3 --sinks 5 --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model > /dev/null 2>&1
  echo "== ctx $c nuq3 + 5 sinks"; python -c "
import json
for l in open('gpurun_out/r05_h_cfg3_ctx${c}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f -> %.1f us per layer' % (d['ms_per_step'], d['ms_per_step'] * 1000 / 32), d['kernels'])
"; grep -v pack_tiled gpurun_out/r05_h_cfg3_ctx${c}_kernel_stats.csv
done
KVQ_SCORE_F16=1 python bench.py --ctx 131072 --bits 3 --sinks 5 --score-f16 --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | cut -c1-400

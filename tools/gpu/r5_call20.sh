#!/bin/bash
# round 5, call 20: kernel stats of the short-context steps (where do the 54 / 85 us per layer go at 4K / 32K?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 4096 32768; do
  bash tools/profile_bench.sh r05_g_ctx${c} --ctx $c --steps 20 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model > /dev/null 2>&1
  echo "== ctx $c nuq4"; python -c "
import json
for l in open('gpurun_out/r05_g_ctx${c}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f -> %.1f us per layer' % (d['ms_per_step'], d['ms_per_step'] * 1000 / 32), d['kernels'])
"; cat gpurun_out/r05_g_ctx${c}_kernel_stats.csv
  bash tools/profile_bench.sh r05_g_cfg3_ctx${c} --ctx $c --bits 3 --sinks 5 --steps 20 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model > /dev/null 2>&1
  echo "== ctx $c nuq3 + 5 sinks"; python -c "
import json
for l in open('gpurun_out/r05_g_cfg3_ctx${c}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f -> %.1f us per layer' % (d['ms_per_step'], d['ms_per_step'] * 1000 / 32), d['kernels'])
"; cat gpurun_out/r05_g_cfg3_ctx${c}_kernel_stats.csv
done

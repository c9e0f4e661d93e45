#!/bin/bash
# round 6: (a) row-layout score kernel, 8-wave (spilling) vs 4-wave tiles at 32K / 128K; (b) per-kernel stats of the decode step at 4K / 32K
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_t_rows_ab.txt; : > $out
for w in 1 0 1 0; do
  echo "== KVQ_ROWS_8W=$w" >> $out
  KVQ_ROWS_8W=$w timeout 300 python tools/ref_bench.py 4 32768 131072 2>/dev/null | grep "legacy row layout" >> $out
  KVQ_ROWS_8W=$w timeout 300 python tools/ref_bench.py 3 131072 2>/dev/null | grep "legacy row layout" >> $out
done
cat $out
for ctx in 4096 32768; do
  cd /tmp; rm -rf /tmp/prof_$ctx
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$ctx -o p -- python $GRAFT_REPO_ROOT/bench.py --ctx $ctx --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > /tmp/prof_$ctx.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/prof_$ctx -name "*kernel_stats.csv" | head -1)
  echo "== ctx $ctx" | tee -a gpurun_out/r06_t_short_kernel_stats.txt
  head -12 "$f" | tee -a gpurun_out/r06_t_short_kernel_stats.txt
  grep '^{' /tmp/prof_$ctx.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['kernels'])" | tee -a gpurun_out/r06_t_short_kernel_stats.txt
done

#!/bin/bash
# round 3, GPU call 31: per-kernel durations of the decode step at 4K and 32K (rocprofv3 kernel stats)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_bench.sh c31_4k --ctx 4096 --steps 20 --no-cpu-baseline --no-fp16-baseline > /dev/null 2>&1
bash tools/profile_bench.sh c31_32k --ctx 32768 --steps 20 --no-cpu-baseline --no-fp16-baseline > /dev/null 2>&1
for t in c31_4k c31_32k; do echo "== $t"; python -c "
import json; d=json.load(open('gpurun_out/${t}_bench.json')); print('ms/step %.3f' % d['ms_per_step'], {k: round(v,1) for k,v in d['kernels'].items() if k.endswith('_us')})"; cat gpurun_out/${t}_kernel_stats.csv; done

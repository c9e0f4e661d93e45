#!/bin/bash
# round 6: phase trace of the decode prologue's selection workgroups, pruning select vs radix select; prefill leg A/B; kernel stats at 4K / 32K
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_v_select_trace.txt; : > $out
for lib in ptrace ptrace_radix; do echo "== $lib" >> $out; KVQ_LIB=tools/abl/libkvq_$lib.so timeout 200 python tools/dbg/trace_prologue.py >> $out 2>&1; done
for lib in "" tools/abl/libkvq_radix.so ""  tools/abl/libkvq_radix.so; do
  echo "== prefill KVQ_LIB=$lib" >> $out
  KVQ_LIB=$lib timeout 300 python bench.py --prefill --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if 'us' in k or 'GBps' in k or k == 'ms_per_step'}, {k: round(v, 1) for k, v in d.get('kernels', {}).items()})" >> $out
done
cat $out
for ctx in 4096 32768; do bash tools/profile_bench.sh r06_v_ctx$ctx --ctx $ctx --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs; done

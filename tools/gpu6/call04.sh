#!/bin/bash
# round 6, call 4: entries + score conversion riding inside the look-up loop: parity (wide forced), trace, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KVQ_V_WIDE_FROM=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_compact_gpu.py tests/test_ties_gpu.py -x -q -m gpu > gpurun_out/r06_d_wide_forced_tests.txt 2>&1
tail -4 gpurun_out/r06_d_wide_forced_tests.txt
timeout 900 python -m pytest tests/test_atsize_gpu.py -x -q -m gpu > gpurun_out/r06_d_atsize_tests.txt 2>&1
tail -3 gpurun_out/r06_d_atsize_tests.txt
{
echo "## trace (KVQ_TRACE build), 128K + 77 tokens"
for b in 4 3; do echo "== bits $b"; BITS=$b KVQ_LIB=tools/abl/libkvq_w_trace.so python tools/dbg/trace_vw.py 2>&1 | grep -v amdgpu.ids; done
echo "## variants: bench.py lines (mix_v_us = p.V kernel + slab reduce between in-stream events)"
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --warmup 3 --steps 10"
for rep in 1 2; do
for v in narrow default w_noent w_nomath; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768"; do
    lib=kvquant_amd/libkvq.so; [ $v != default ] && [ $v != narrow ] && lib=tools/abl/libkvq_$v.so
    w=1; [ $v = narrow ] && w=0
    KVQ_V_WIDE=$w KVQ_V_WIDE_FROM=1 KVQ_LIB=$lib python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_d_wide_trace.txt 2>&1
cat gpurun_out/r06_d_wide_trace.txt

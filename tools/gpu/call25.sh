#!/bin/bash
# round 3, GPU call 25: the profiles of the round (bench + rocprof kernel stats, PMC, sweep, prefill pack, reference kernels,
# PPL harness, host overhead, traces, full GPU test log)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=r03_b; O=gpurun_out/$T
export TMPDIR=/tmp
bash tools/profile_bench.sh $T > ${O}_profile.log 2>&1
PMC_OUT=/tmp bash tools/pmc_run.sh $T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline > ${O}_pmc_bench.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_traffic.py ${O}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json > /dev/null 2>&1
timeout 1200 python bench.py --no-cpu-baseline --no-fp16-baseline --sweep 2> ${O}_sweep.err | head -n 9 > ${O}_sweep.jsonl
( timeout 300 python tools/prefill_bench.py 8192 4; timeout 300 python tools/prefill_bench.py 8192 3; echo "--- per-token workgroups (KVQ_PACK_TILED=0):"; KVQ_LIB=tools/abl/libkvq_perq.so timeout 300 python tools/prefill_bench.py 8192 4 ) 2>&1 | grep -v amdgpu > ${O}_prefill_pack.txt
PMC_OUT=/tmp bash tools/pmc_run.sh ${T}_pack python tools/prefill_bench.py 8192 4 2>&1 | grep -A22 "pack_tiled_kernel<4" > ${O}_pmc_prefill_pack.txt
timeout 600 python tools/ref_bench.py 2>&1 | grep -v amdgpu > ${O}_ref_bench.jsonl
timeout 900 python tools/ppl_delta.py 2048 2>&1 | grep -v amdgpu > ${O}_ppl_delta.jsonl
( timeout 300 python tools/dbg/host_overhead.py 4096 2>&1 | head -3; KVQ_DECODE_MULTICALL=1 timeout 300 python tools/dbg/host_overhead.py 4096 2>&1 | head -3 ) | grep -v amdgpu > ${O}_host_overhead.txt
( echo "== score_k (default build + KVQ_TRACE)"; KVQ_LIB=tools/abl/libkvq_trk.so timeout 300 python tools/dbg/trace_k.py; echo "== score_k, round-2 scheme (KVQ_K_JIT=0)"; KVQ_LIB=tools/abl/libkvq_r2k.so KB_ONLY=score_k timeout 300 python tools/kbench2.py 4 131149; KB_ONLY=score_k timeout 300 python tools/kbench2.py 4 131149; echo "== mix_v (KVQ_TRACE)"; KVQ_LIB=tools/abl/libkvq_trv.so timeout 300 python tools/dbg/trace_v.py ) 2>&1 | grep -v amdgpu > ${O}_traces.txt
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > ${O}_gpu_tests.txt
cat ${O}_bench.json | head -c 3000; echo; cat ${O}_kernel_stats.csv; tail -4 ${O}_gpu_tests.txt; cat ${O}_host_overhead.txt; cat ${O}_prefill_pack.txt

#!/bin/bash
# round 3, GPU call 5: tiled prefill pack (4 tokens per workgroup): parity tests, timing, PMC traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c5
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ties_gpu.py tests/test_atsize_gpu.py tests/test_decode_kv_gpu.py tests/test_fused_gpu.py tests/test_cache_gpu.py tests/test_attention_gpu.py tests/test_simquant_gpu.py -m gpu -q 2>&1 | tail -15 ) > ${O}_tests.txt
timeout 300 python tools/prefill_bench.py 8192 4 2>&1 | grep -v amdgpu > ${O}_prefill.txt
timeout 300 python tools/prefill_bench.py 8192 3 2>&1 | grep -v amdgpu >> ${O}_prefill.txt
PMC_OUT=/tmp bash tools/pmc_run.sh c5 python tools/prefill_bench.py 8192 4 > ${O}_pmc_prefill.txt 2>&1
cat ${O}_tests.txt ${O}_prefill.txt; grep -E "pack_tiled|fused_pack|pack_parallel|kernel" ${O}_pmc_prefill.txt | head -30

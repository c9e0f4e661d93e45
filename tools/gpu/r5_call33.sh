#!/bin/bash
# round 5, call 33: PMC passes of the default bench on the final sources (a dead define left kvq_mix_v.hip: new source hash) -> pmc_traffic.json;
# p.V / decode tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=r05_x; O=gpurun_out/$T
export TMPDIR=/tmp
PMC_OUT=/tmp bash tools/pmc_run.sh $T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_pmc_bench.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_traffic.py ${O}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json > ${O}_pmc_traffic.log 2>&1
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decode_kv_gpu.py -m gpu -q -x 2>&1 | tail -3 ) > ${O}_tests.txt
python bench.py --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | cut -c1-1500 > ${O}_bench_line.txt
cat ${O}_tests.txt; tail -6 gpurun_out/pmc_traffic.json; grep -o '"traffic": [0-9a-z]*' ${O}_bench_line.txt; grep -o '"ms_per_step": [0-9.]*' ${O}_bench_line.txt

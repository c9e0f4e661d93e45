#!/bin/bash
# round 5, final call: the profiles of the round -- default bench (+ full model leg) and rocprofv3 kernel stats of the same command,
# PMC passes -> HBM traffic per launch stamped with the kernels' code hash, config 3 kernel stats + PMC, the sweep (+ config 4 lines),
# PPL deltas, the GPU test log, smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=r05_z; O=gpurun_out/$T
export TMPDIR=/tmp
bash tools/profile_bench.sh $T > ${O}_profile.log 2>&1
PMC_OUT=/tmp bash tools/pmc_run.sh $T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_pmc_bench.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_traffic.py ${O}_pmc_bench.txt 4 131072 gpurun_out/pmc_traffic.json > ${O}_pmc_traffic.log 2>&1
# config 3 (nuq3 + 5 sinks @128K): kernel stats + counters of its own
bash tools/profile_bench.sh ${T}_cfg3 --ctx 131072 --bits 3 --sinks 5 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_cfg3_profile.log 2>&1
PMC_OUT=/tmp bash tools/pmc_run.sh ${T}_cfg3 python bench.py --ctx 131072 --bits 3 --sinks 5 --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_cfg3_pmc_bench.txt 2>&1
timeout 1800 python bench.py --no-cpu-baseline --no-fp16-baseline --no-full-model --sweep 2> ${O}_sweep.err > ${O}_sweep.jsonl
timeout 1500 python tools/ppl_delta.py 2048 600 2 > ${O}_ppl_delta.jsonl 2> ${O}_ppl.err
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > ${O}_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.txt 2>&1
head -c 2500 ${O}_bench.json; echo; cat ${O}_kernel_stats.csv; cat ${O}_cfg3_kernel_stats.csv; tail -4 ${O}_gpu_tests.txt; cat gpurun_out/pmc_traffic.json; tail -1 ${O}_smoke.txt

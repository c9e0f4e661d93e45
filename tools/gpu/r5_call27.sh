#!/bin/bash
# round 5, call 27: the reference's own kernels (oracle/_ref, built for gfx950) next to libkvq on this box; PMC pass of the 32K config
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_n
export TMPDIR=/tmp
for a in "4 4096" "4 32768" "4 131072" "3 131072"; do timeout 600 python tools/ref_bench.py $a 2>/dev/null | grep "^{" >> ${O}_ref_bench.jsonl; done
PMC_OUT=/tmp bash tools/pmc_run.sh r05_n32k python bench.py --ctx 32768 --steps 4 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model > ${O}_pmc_bench_32k.txt 2>&1
cat ${O}_ref_bench.jsonl | cut -c1-700; tail -30 ${O}_pmc_bench_32k.txt | cut -c1-200

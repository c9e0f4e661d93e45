#!/bin/bash
# usage: tools/pmc_run.sh <tag> <python script + args...>   (run on the GPU box; one rocprofv3 pass per counter set;
# raw counter CSVs go to $PMC_OUT (default /tmp: gpurun_out is capped at 64 MiB), the per-kernel summary to stdout)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=(
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
"SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
"FETCH_SIZE GRBM_GUI_ACTIVE"
"WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d ${PMC_OUT:-/tmp}/pmc_${tag}_$i -o p -- "$@" > ${PMC_OUT:-/tmp}/pmc_${tag}_$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py ${PMC_OUT:-/tmp}/pmc_${tag}_ $i

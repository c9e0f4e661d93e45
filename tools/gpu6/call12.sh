#!/bin/bash
# round 6, call 12: PMC passes over the prefill leg (config 4): the MFMA attention's matrix-core utilisation, the pack kernels' LDS side
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*" | sort -u | head -20 > gpurun_out/r06_l_mfma_counters.txt
cat gpurun_out/r06_l_mfma_counters.txt
cd /tmp
i=0
for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  timeout 240 rocprofv3 --pmc $s --output-format csv -d /tmp/pmc_r06_pre_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --prefill > /tmp/pmc_r06_pre_$i.log 2>&1
  echo "pass $i rc $?"; tail -2 /tmp/pmc_r06_pre_$i.log
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc_r06_pre_ 4 > gpurun_out/r06_l_pmc_prefill.txt 2>&1
cat gpurun_out/r06_l_pmc_prefill.txt

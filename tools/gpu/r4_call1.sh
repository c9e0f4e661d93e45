#!/bin/bash
# round 4, GPU call 1: pair-table microbenchmark (replication 8/16/32, address forms, fmac vs pk_fma)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4c1
timeout 120 tools/ubench/pair_table2 > ${O}_pair.txt 2>&1
timeout 60 tools/ubench/lut_rate >> ${O}_pair.txt 2>&1
cat ${O}_pair.txt

#!/bin/bash
# round 5, call 16: instruction-rate ubench with the shader clock read out (s_memtime / s_memrealtime + rocm-smi samples while it
# runs), the 3-bit p.V pair-row forms; bench lines that separate "3 bit" from "sinks" in p.V
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_c
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3; sleep 0.4; done ) > ${O}_smi_clocks.txt 2>&1 &
SMI=$!
timeout 120 tools/ubench/lut_rate > ${O}_lut_rate.txt 2>&1
timeout 120 tools/ubench/lut_rate >> ${O}_lut_rate.txt 2>&1
wait $SMI
B="--steps 10 --warmup 3 --no-cpu-baseline --no-fp16-baseline --no-full-model"
for cfg in "--ctx 131072 --bits 3 --sinks 0" "--ctx 131072 --bits 4 --sinks 5" "--ctx 32768 --bits 3 --sinks 0" "--ctx 32768 --bits 4 --sinks 5"; do
  timeout 300 python bench.py $B $cfg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; k = d['kernels']
        print('$cfg', '| ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k['score_k_us'], k['mix_v_us']))
" >> ${O}_bits_vs_sinks.txt
done
cat ${O}_lut_rate.txt | head -40; cat ${O}_smi_clocks.txt | sort | uniq -c | head; cat ${O}_bits_vs_sinks.txt

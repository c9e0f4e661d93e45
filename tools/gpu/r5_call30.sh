#!/bin/bash
# round 5, call 30: run-to-run spread of the default line on ONE box (box-to-box: profiles/r05_{a,b,z,y}_bench.json)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_q
for i in 1 2 3 4; do python bench.py --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        print('run $i: %.3f ms/step  %.1f tok/s  roofline.frac %.3f  step.frac %.3f  score_k %.1f  mix_v %.1f  traffic %s' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_step']['frac'], k['score_k_us'], k['mix_v_us'], d['roofline']['traffic']))
"; done | tee ${O}_same_box.txt

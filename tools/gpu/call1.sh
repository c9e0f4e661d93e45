#!/bin/bash
# round 3, GPU call 1: correctness of the JIT score kernel + L2-touch p.V kernel, A/B timings, phase traces
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > ${O}_tests.txt
for v in "" r2base j42t3 j24t1 j0t4; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  timeout 300 python tools/kbench2.py 4 131149 32768 4096 2>&1 | grep -v Warn
done > ${O}_kbench.txt 2>&1
unset KVQ_LIB
for v in trk0 trk1; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_k.py 2>&1 | tail -12; done > ${O}_trace_k.txt
for v in trv0 trv2; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_v.py 2>&1 | tail -16; done > ${O}_trace_v.txt
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench.json 2> ${O}_bench.err
KVQ_LIB=tools/abl/libkvq_r2base.so timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench_r2base.json 2>> ${O}_bench.err
tail -3 ${O}_tests.txt; cat ${O}_kbench.txt; cat ${O}_trace_k.txt ${O}_trace_v.txt; cat ${O}_bench.json | python -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['kernels'])"

#!/bin/bash
# round 3, GPU call 6: woven dense section (KVQ_K_WEAVE): timing, trace, correctness
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c6
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in "" w2 w2q w2n4; do
  if [ -z "$v" ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$v.so; fi
  KB_ONLY=score_k KB_ITERS=200 timeout 300 python tools/kbench2.py 4 131149 2>&1 | grep -v "amdgpu.ids"
done; done > ${O}_kbench.txt 2>&1
unset KVQ_LIB
for v in trk_w2; do echo "== $v"; KVQ_LIB=tools/abl/libkvq_$v.so timeout 300 python tools/dbg/trace_k.py 2>&1 | grep -v amdgpu.ids | tail -34; done > ${O}_trace_k.txt
cp kvquant_amd/libkvq.so /tmp/libkvq_keep.so; cp tools/abl/libkvq_w2.so kvquant_amd/libkvq.so
( timeout 900 python -m pytest tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_fuzz_gpu.py tests/test_atsize_gpu.py tests/test_fullsize_gpu.py tests/test_ties_gpu.py -m gpu -q 2>&1 | tail -5 ) > ${O}_tests_w2.txt
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench_w2.json 2> ${O}_bench.err
cp /tmp/libkvq_keep.so kvquant_amd/libkvq.so
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench_default.json 2>> ${O}_bench.err
cat ${O}_kbench.txt ${O}_trace_k.txt ${O}_tests_w2.txt
python - <<'PY'
import json
for f in ("gpurun_out/c6_bench_w2.json", "gpurun_out/c6_bench_default.json"):
    try:
        d = json.load(open(f)); print(f, "tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items()})
    except Exception as e:
        print(f, e)
PY

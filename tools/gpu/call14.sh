#!/bin/bash
# round 3, GPU call 14: p.V with the outlier entries taken by the channel group's own workgroup (no outlier slabs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c14
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_fuzz_gpu.py tests/test_compact_gpu.py tests/test_ties_gpu.py tests/test_fullsize_gpu.py tests/test_atsize_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > ${O}_tests.txt
for rep in 1 2; do
  KB_ONLY=mix_v KB_ITERS=150 timeout 300 python tools/kbench2.py 4 131149 32768 4096 2>&1 | grep -v "amdgpu.ids"
  KB_ONLY=mix_v KB_ITERS=150 timeout 300 python tools/kbench2.py 3 131149 2>&1 | grep -v "amdgpu.ids"
done > ${O}_kbench.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-fp16-baseline > ${O}_bench.json 2> ${O}_bench.err
cat ${O}_tests.txt ${O}_kbench.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/c14_bench.json")); print("tok/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["kernels"].items()})
PY

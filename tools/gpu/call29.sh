#!/bin/bash
# round 3, GPU call 29: in-kernel merge of the softmax partials with batched loads, own heads only (up to 1024 tiles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c29
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_decode_kv_gpu.py tests/test_atsize_gpu.py tests/test_fuzz_gpu.py tests/test_compact_gpu.py tests/test_llama_gpu.py tests/test_sharding_gpu.py -m gpu -q 2>&1 | tail -4 ) > ${O}_tests.txt
for cfg in "--ctx 4096 --steps 20" "--ctx 32768 --steps 20" "--ctx 65536" "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 262144 --layers 16"; do
  timeout 900 python bench.py $cfg --no-cpu-baseline --no-fp16-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg: %.3f ms/step  %.1f tok/s' % (d['ms_per_step'], d['value']), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done > ${O}_bench.txt 2>&1
cat ${O}_tests.txt ${O}_bench.txt

#!/bin/bash
# round 6, call 9: non-temporal policy on the p.V tile rows (same-box A/B); PPL draws of config 3 (3 x nuq3 + 5 sinks)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3 --steps 10"
for rep in 1 2 3; do
for v in default w_nt; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768"; do
    lib=kvquant_amd/libkvq.so; [ $v != default ] && lib=tools/abl/libkvq_$v.so
    KVQ_LIB=$lib timeout 120 python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('$v $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_i_nt_ab.txt 2>&1
cat gpurun_out/r06_i_nt_ab.txt
timeout 900 python - > gpurun_out/r06_i_ppl_cfg3.jsonl 2> gpurun_out/r06_i_ppl_cfg3.err <<'PY'
import json, sys
sys.path.insert(0, ".")
from tests import ppl_harness
for rep in range(3):
    r = ppl_harness.run(layers=2, n_tokens=2048, vocab=4096, train_steps=600, bits=3, first_few_fp16=5, seed=100 + rep) if "seed" in ppl_harness.run.__code__.co_varnames else ppl_harness.run(layers=2, n_tokens=2048, vocab=4096, train_steps=600, bits=3, first_few_fp16=5)
    print(json.dumps(r), flush=True)
PY
python - <<'PY'
import json
for l in open("gpurun_out/r06_i_ppl_cfg3.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in d.items() if k.startswith("rel_") or k.startswith("ppl_")})
PY
tail -3 gpurun_out/r06_i_ppl_cfg3.err

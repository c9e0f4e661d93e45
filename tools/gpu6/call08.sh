#!/bin/bash
# round 6, call 8: padded probability rows in the wide p.V kernel (parity, same-box A/B vs the narrow kernel, PMC conflicts);
# three PPL draws of every configuration; PMC of the MFMA prefill attention; multi-rank smoke of bench.py on the one GPU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KVQ_V_WIDE_FROM=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu.py tests/test_decode_kv_gpu.py tests/test_compact_gpu.py tests/test_ties_gpu.py tests/test_atsize_gpu.py -x -q -m gpu > gpurun_out/r06_h_wide_forced_tests.txt 2>&1
tail -3 gpurun_out/r06_h_wide_forced_tests.txt
{
B="--no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs --warmup 3 --steps 10"
for rep in 1 2; do
for w in 0 1; do
  for cfg in "--ctx 131072" "--ctx 131072 --bits 3 --sinks 5" "--ctx 32768" "--ctx 131072 --bits 2"; do
    KVQ_V_WIDE=$w python bench.py $cfg $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('wide=$w $cfg | ms/step %.3f score_k %.1f mix_v %.1f' % (d['ms_per_step'], k.get('score_k_us',0), k.get('mix_v_us',0)))
"
  done
done
done
} > gpurun_out/r06_h_wide_ab.txt 2>&1
cat gpurun_out/r06_h_wide_ab.txt
PMC_OUT=/tmp bash tools/pmc_run.sh r06_h python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model --no-configs > gpurun_out/r06_h_pmc_bench.txt 2>&1
grep -A22 "mix_v_wide_kernel<4" gpurun_out/r06_h_pmc_bench.txt | grep "kvq\|FETCH\|WRITE\|CONFLICT\|IDX_ACTIVE\|INSTS_LDS\|WAIT_INST_LDS"
# PMC of the prefill (config 4): pack kernels + MFMA attention
cd /tmp
i=0
for s in "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_WAVE_CYCLES"; do
  rocprofv3 --pmc $s --output-format csv -d /tmp/pmc_r06_pre_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --prefill > /tmp/pmc_r06_pre_$i.log 2>&1
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc_r06_pre_ 3 > gpurun_out/r06_h_pmc_prefill.txt 2>&1
cat gpurun_out/r06_h_pmc_prefill.txt | head -60
# three PPL draws
python tools/ppl_delta.py 2048 600 3 > gpurun_out/r06_h_ppl_delta.jsonl 2> gpurun_out/r06_h_ppl_delta.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_h_ppl_delta.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in d.items() if k in ("bits", "first_few_fp16", "norm", "n_prompt", "ppl_kernel", "ppl_fp16", "rel_delta", "rel_delta_vs_deploy_arith", "rel_delta_vs_deploy_rule")})
PY
# multi-rank smoke on the one GPU
{
echo "# round 6: multi-rank code paths of bench.py, all ranks on the ONE visible GPU (KVQ_BENCH_ONE_GPU=1, hand-overs over gloo): a smoke run of the rank logic, the times mean nothing"
for cfg in "--gpus 2 --layers 4" "--gpus 4 --layers 4" "--gpus 2 --layers 4 --shard tokens" "--gpus 2 --layers 4 --shard heads" "--gpus 2 --layers 4 --shard heads --bits 3 --sinks 5"; do
  echo "== bench.py $cfg"
  KVQ_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $(echo $cfg | awk '{print $2}') --master-addr 127.0.0.1 --master-port 29517 bench.py $cfg --ctx 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-baseline --no-full-model 2>/dev/null | grep '^{' | cut -c1-400
done
} > gpurun_out/r06_h_multirank_smoke.txt 2>&1
cat gpurun_out/r06_h_multirank_smoke.txt | cut -c1-200

#!/bin/bash
# round-2 GPU call D: bf16x3 v2 (software-pipelined) parity + speed; clean HBM-cold K1 rocprofv3 trace
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -q --no-header -rf -k "mlp or c4" > gpurun_out/r02d_pytest_mlp.log 2>&1
echo "mlp rc=$?" >> gpurun_out/r02d_pytest_mlp.log
for E in 0 1; do
  MPPI_MLP_EXACT=$E timeout 300 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02d_bench_c4_exact$E.json 2>> gpurun_out/r02d_bench_c4.err
done
H=256 timeout 120 python tools/diag_mlp_mfma.py > gpurun_out/r02d_mlp_diag.txt 2>&1
REPO=$PWD
(cd /tmp && ONLY_COLD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_k1coldonly -o k1cold -- python $REPO/tools/k1_sweep.py > $REPO/gpurun_out/r02d_prof_k1cold.log 2>&1)
DB=$(find gpurun_out/prof_k1coldonly -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB gpurun_out/r02d_k1_hbm_cold_only_trace.txt > /dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c4 -o c4 -- python $REPO/bench.py --workload c4 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r02d_prof_c4.log 2>&1)
DB=$(find gpurun_out/prof_c4 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB gpurun_out/r02d_c4_kernel_trace.txt > /dev/null
tail -6 gpurun_out/r02d_pytest_mlp.log; cat gpurun_out/r02d_mlp_diag.txt | grep path; python - <<'PY'
import json
for e in (0,1):
    try:
        d=json.load(open(f'gpurun_out/r02d_bench_c4_exact{e}.json')); r=d['roofline']
        print('exact',e,'ms/step',d['ms_per_step'],'K1 us',r['avg_launch_us'],'TF',r['achieved'],'frac',r['frac'])
    except Exception as ex: print('bench c4', e, ex)
PY
grep -i "k1\|rollout" gpurun_out/r02d_prof_k1cold.log | head -3; head -5 gpurun_out/r02d_k1_hbm_cold_only_trace.txt; head -8 gpurun_out/r02d_c4_kernel_trace.txt

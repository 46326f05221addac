#!/bin/bash
# round-2 GPU call C: bf16x3 MLP kernel (parity + speed), K1 ring-depth A/B (HBM-cold), row-stride experiment
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -q --no-header -rf -k "mlp or c4" > gpurun_out/r02c_pytest_mlp.log 2>&1
echo "mlp rc=$?" >> gpurun_out/r02c_pytest_mlp.log
for E in 0 1; do
  MPPI_MLP_EXACT=$E timeout 300 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02c_bench_c4_exact$E.json 2>> gpurun_out/r02c_bench_c4.err
done
H=256 timeout 120 python tools/diag_mlp_mfma.py >> gpurun_out/r02c_mlp_diag.txt 2>&1
H=256 MPPI_MLP_EXACT=1 timeout 120 python tools/diag_mlp_mfma.py >> gpurun_out/r02c_mlp_diag.txt 2>&1
for rep in 1 2; do
  for V in "" _r12 _r15; do
    MPPI_LIB_SUFFIX=$V timeout 300 python tools/k1_sweep.py 2>&1 | grep "^\[" | grep K1 | sed "s/^\[/[ROWS${V:-_r9} /" >> gpurun_out/r02c_k1_rows.txt
  done
done
for K in 262144 262400 266240 278528 327680 393216; do
  K=$K ONLY_COLD=1 timeout 300 python tools/k1_sweep.py 2>&1 | grep "^\[" >> gpurun_out/r02c_k1_stride.txt
done
timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02c_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02c_pytest.log
tail -15 gpurun_out/r02c_pytest_mlp.log; cat gpurun_out/r02c_mlp_diag.txt; python - <<'PY'
import json
for e in (0,1):
    try:
        d=json.load(open(f'gpurun_out/r02c_bench_c4_exact{e}.json')); r=d['roofline']
        print('exact',e,'ms/step',d['ms_per_step'],'K1 us',r['avg_launch_us'],'TF',r['achieved'],'frac',r['frac'])
    except Exception as ex: print('bench c4', e, ex)
PY
cat gpurun_out/r02c_k1_rows.txt gpurun_out/r02c_k1_stride.txt; tail -3 gpurun_out/r02c_pytest.log

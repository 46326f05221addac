#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-e}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -q --no-header -rf -k "mlp or c4" > gpurun_out/r02${T}_pytest_mlp.log 2>&1
echo "mlp rc=$?" >> gpurun_out/r02${T}_pytest_mlp.log
MPPI_MLP_EXACT=0 timeout 300 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02${T}_bench_c4.json 2>> gpurun_out/r02${T}_bench_c4.err
H=256 timeout 120 python tools/diag_mlp_mfma.py > gpurun_out/r02${T}_mlp_diag.txt 2>&1
tail -4 gpurun_out/r02${T}_pytest_mlp.log; grep path gpurun_out/r02${T}_mlp_diag.txt; python - <<PY
import json
d=json.load(open('gpurun_out/r02${T}_bench_c4.json')); r=d['roofline']
print('ms/step',d['ms_per_step'],'K1 us',r['avg_launch_us'],'TF',r['achieved'],'frac',r['frac'])
PY

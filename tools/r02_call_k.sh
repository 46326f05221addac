#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02k_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02k_pytest.log
for K in 131072 262144 524288; do
  for D in 0 1; do
    K=$K DENSE_PITCH=$D ONLY_COLD=1 timeout 300 python tools/k1_sweep.py 2>&1 | grep "^\[" | sed "s/^\[/[DENSE_PITCH=$D /" >> gpurun_out/r02k_k1_pitch.txt
  done
done
timeout 300 python tools/k_sweep.py > gpurun_out/r02k_k_sweep.txt 2>&1
tail -4 gpurun_out/r02k_pytest.log; cat gpurun_out/r02k_k1_pitch.txt; grep float32 gpurun_out/r02k_k_sweep.txt

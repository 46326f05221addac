#!/bin/bash
# round-2 GPU call A: full gpu suite + K1 LDS-DMA A/B + default bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf -x --deselect tests/test_gpu_fullsize_parity.py > gpurun_out/r02a_pytest_main.log 2>&1
echo "main suite rc=$?" >> gpurun_out/r02a_pytest_main.log
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -q --no-header -rf > gpurun_out/r02a_pytest_fullsize.log 2>&1
echo "fullsize rc=$?" >> gpurun_out/r02a_pytest_fullsize.log
for K in 65536 262144; do
  for D in 0 30 15; do
    K=$K MPPI_K1_DMA=$D timeout 300 python tools/k1_sweep.py 2>&1 | grep "^\[" >> gpurun_out/r02a_k1_sweep.txt
  done
done
timeout 600 python bench.py > gpurun_out/r02a_bench_default.json 2> gpurun_out/r02a_bench_default.err
MPPI_K1_DMA=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02a_bench_dma0.json 2>> gpurun_out/r02a_bench_default.err
tail -3 gpurun_out/r02a_pytest_main.log; tail -5 gpurun_out/r02a_pytest_fullsize.log; cat gpurun_out/r02a_k1_sweep.txt

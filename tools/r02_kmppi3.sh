#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
tools/micro/kmppi_parts_0 2>&1 | grep KMPPI | tee gpurun_out/kmppi_parts_after.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/kmppi_bench.py philox 2>&1 | grep KMPPI | tee gpurun_out/kmppi_bench2.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -o t -- python /root/repo/tools/kmppi_bench.py philox > /dev/null 2>&1; DB=$(find /tmp/pm -name '*.db' | head -1); python /root/repo/tools/pmc_summary.py $DB > /root/repo/gpurun_out/kmppi_pmc_mfma.txt 2>&1)
head -20 gpurun_out/kmppi_pmc_mfma.txt | cut -c1-200

#!/bin/bash
# GPU call: fused-KMPPI tests, timing, trace
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "kmppi or KMPPI" 2>&1 | tail -15 > gpurun_out/kmppi_tests.txt
cat gpurun_out/kmppi_tests.txt
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/kmppi_bench.txt 2>&1
timeout 300 python tools/kmppi_bench.py torch >> gpurun_out/kmppi_bench.txt 2>&1
cat gpurun_out/kmppi_bench.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- python /root/repo/tools/kmppi_bench.py philox > /dev/null 2>&1; DB=$(find /tmp/pk -name '*.db' | head -1); python /root/repo/tools/prof_summary.py $DB /root/repo/gpurun_out/kmppi_trace.txt > /dev/null 2>&1)
head -14 gpurun_out/kmppi_trace.txt | cut -c1-220
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/all_tests.txt
cat gpurun_out/all_tests.txt

#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do timeout 300 python tools/kmppi_bench.py philox 2>&1 | grep KMPPI; done | tee gpurun_out/kmppi_bench3.txt
tools/micro/kmppi_parts_0 2>&1 | grep KMPPI
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- python /root/repo/tools/kmppi_bench.py philox > /dev/null 2>&1; DB=$(find /tmp/pk -name '*.db' | head -1); python /root/repo/tools/prof_summary.py $DB /root/repo/gpurun_out/kmppi_trace2.txt > /dev/null 2>&1)
head -8 gpurun_out/kmppi_trace2.txt | cut -c1-200

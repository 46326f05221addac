#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
for rep in 1 2; do for f in 1 0; do
  echo "MPPI_K3_FIN=$f"; MPPI_K3_FIN=$f timeout 300 python tools/variants_bench.py philox "MPPI" 2>&1 | grep "ms/command" | head -3
done; done | tee gpurun_out/k3fin_ab.txt
(cd /tmp && export TMPDIR=/tmp && MPPI_K3_FIN=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- python /root/repo/bench.py --no-extras --no-cpu-baseline > /dev/null 2>&1; DB=$(find /tmp/pk -name '*.db' | head -1); python /root/repo/tools/prof_summary.py $DB /root/repo/gpurun_out/k3fin_trace.txt > /dev/null 2>&1)
head -8 gpurun_out/k3fin_trace.txt | cut -c1-180

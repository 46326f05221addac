#!/bin/bash
# round 3, call C: on-chip retention prototype (tools/micro/k1ret_micro.hip) + the HBM-cold K1 launches under rocprofv3
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 300 ./tools/micro/k1ret_micro > gpurun_out/r03c_k1ret_micro.txt 2>&1
cat gpurun_out/r03c_k1ret_micro.txt
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r03_launches_c3cold.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c3cold -o c3cold -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r03c_bench_under_rocprof_c3cold.json 2> $REPO/gpurun_out/r03c_prof_c3cold.log)
DB=$(find gpurun_out/prof_c3cold -name "*.db" | head -1)
python tools/clock_calibration.py $DB gpurun_out/r03_launches_c3cold.json rollout_cost_kernel gpurun_out/r03c_clock_calibration_c3_cold.txt
rm -rf gpurun_out/prof_c3cold

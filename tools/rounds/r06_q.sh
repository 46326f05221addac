#!/bin/bash
# round 6: a kernel trace of the headline command in which most launches are steady-state ones (300 timed commands): the table's
# average itself is then comparable with the bench line's live launch time
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r06_q}
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/${P}_launches.json timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_q -o q -- python $REPO/bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_c3_steady.json 2> $REPO/gpurun_out/${P}_prof.log)
DB=$(find gpurun_out/prof_q -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/${P}_trace_c3_steady.txt > /dev/null
python tools/clock_calibration.py $DB gpurun_out/${P}_launches.json gpurun_out/${P}_clock_calibration_c3_steady.txt
rm -rf gpurun_out/prof_q
head -6 gpurun_out/${P}_trace_c3_steady.txt | cut -c1-200
grep "headline:" gpurun_out/${P}_clock_calibration_c3_steady.txt | cut -c1-260
python -c "
import json;d=json.loads(open('gpurun_out/${P}_bench_under_rocprof_c3_steady.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"

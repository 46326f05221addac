#!/bin/bash
# round 4, call C: evidence for profiles/ on the head -- C3 default bench under rocprofv3 (kernel table, idle gaps of a command,
# clock calibration on the same launches), PMC passes (VALU / traffic) of the on-chip command, C4 trace + matrix-pipe busy, KMPPI trace
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
run_prof() {  # name, extra bench args...
  name=$1; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r04c_launches_$name.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04c_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/r04c_prof_$name.log)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/r04c_trace_$name.txt > /dev/null
  python tools/clock_calibration.py $DB gpurun_out/r04c_launches_$name.json gpurun_out/r04c_clock_calibration_$name.txt
}
run_pmc() {  # name, counters, bench args...
  name=$1; ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04c_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/r04c_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
run_prof c3
DB=$(find gpurun_out/prof_c3 -name "*.db" | head -1)
python tools/timeline_gaps.py $DB rollout_onchip_kernel gpurun_out/r04c_timeline_gaps_c3.txt
rm -rf gpurun_out/prof_c3
run_prof c4 --workload c4
rm -rf gpurun_out/prof_c4
run_prof c2 --workload c2
rm -rf gpurun_out/prof_c2
run_pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run_pmc c3_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run_pmc c3_fetch "FETCH_SIZE"
run_pmc c3_write "WRITE_SIZE"
run_pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c4
rocprofv3 -L 2>/dev/null | grep -oE "SQ_INSTS_[A-Z_0-9]+|SQ_ACTIVE_INST_[A-Z_0-9]+" | sort -u > gpurun_out/r04c_sq_counter_names.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_kmppi -o kmppi -- python $REPO/tools/kmppi_bench.py philox > $REPO/gpurun_out/r04c_kmppi_bench_under_rocprof.txt 2>&1)
DB=$(find gpurun_out/prof_kmppi -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04c_trace_kmppi.txt > /dev/null
rm -rf gpurun_out/prof_kmppi
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04c_bench_default.json 2> gpurun_out/r04c_bench_default.err
for n in c3 c4 c2; do tail -3 gpurun_out/r04c_clock_calibration_$n.txt; head -8 gpurun_out/r04c_trace_$n.txt | cut -c1-200; done
head -9 gpurun_out/r04c_timeline_gaps_c3.txt
grep -E "rollout_onchip|rollout_mlp_split" gpurun_out/r04c_pmc_*.txt | cut -c1-160
head -12 gpurun_out/r04c_trace_kmppi.txt | cut -c1-200

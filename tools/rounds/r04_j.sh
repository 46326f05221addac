#!/bin/bash
# round 4, the head with the on-chip command's rows waiting in memory (ABI 20): suite, default bench (driver's protocol) with and without
# the array, the C3 line under rocprofv3 with the clock calibration on the same launches, SQ / memory counters of the on-chip K1
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=r04_spill
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -3 gpurun_out/${P}_pytest.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
MPPI_ONCHIP_SPILL=0 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default_generate_twice.json
name=c3
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/${P}_launches_$name.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/${P}_prof_$name.log)
DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
python tools/clock_calibration.py $DB gpurun_out/${P}_launches_$name.json gpurun_out/${P}_clock_calibration_$name.txt
python tools/timeline_gaps.py $DB rollout_onchip_kernel gpurun_out/${P}_timeline_gaps_$name.txt > /dev/null
rm -rf gpurun_out/prof_$name
run_pmc() {
  name=$1; ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
run_pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run_pmc c3_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run_pmc c3_classes "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
run_pmc c3_fetch "FETCH_SIZE"
run_pmc c3_write "WRITE_SIZE"
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
python - <<'PY'
import json
P="gpurun_out/r04_spill"
for n in ("default","default_generate_twice"):
    d=json.load(open(f"{P}_bench_{n}.json")); print(n, d["ms_per_step"], d["value"], d.get("onchip",{}).get("k1_avg_us"))
PY
grep "rollout_onchip" gpurun_out/${P}_pmc_c3_*.txt | cut -c1-150
head -8 gpurun_out/${P}_trace_c3.txt | cut -c1-160; head -3 gpurun_out/${P}_clock_calibration_c3.txt

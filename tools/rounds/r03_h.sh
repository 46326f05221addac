#!/bin/bash
# round 3, call H: HBM traffic of the on-chip command by PMC (separate FETCH_SIZE / WRITE_SIZE passes), and the VALU / busy
# counters of the on-chip K1
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
pmc() { # name, counters, command...
  local name=$1; local ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- "$@" > $REPO/gpurun_out/r03h_pmc_$name.log 2>&1)
  local DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/r03h_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
pmc c3_fetch FETCH_SIZE python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
pmc c3_write WRITE_SIZE python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
grep -h "onchip\|finalize_blocks\|rollout_cost_kernel\|noise_fill" gpurun_out/r03h_pmc_c3_fetch.txt gpurun_out/r03h_pmc_c3_write.txt gpurun_out/r03h_pmc_c3_valu.txt | cut -c1-200

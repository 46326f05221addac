#!/bin/bash
# round 6: LDS counters of the two-wave on-chip K1 (the hand-over and the kept rows go through LDS): bank conflicts against LDS-active time
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r06_o}
for ctr in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $ctr | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
  grep -h "rollout_onchip" gpurun_out/${P}_pmc_$name.txt | cut -c1-140
done

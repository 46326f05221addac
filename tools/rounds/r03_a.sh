#!/bin/bash
# round 3, call A: which event clock equals rocprofv3's kernel duration (tools/micro/event_clock.hip) + suite sanity
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
for ticks in 3300 800; do
  ./tools/micro/event_clock $ticks 40 > gpurun_out/r03_event_clock_plain_$ticks.txt 2>&1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_evclk_$ticks -o evclk -- $REPO/tools/micro/event_clock $ticks 40 > $REPO/gpurun_out/r03_event_clock_rocprof_$ticks.txt 2>&1)
  DB=$(find gpurun_out/prof_evclk_$ticks -name "*.db" | head -1)
  python - "$DB" >> gpurun_out/r03_event_clock_rocprof_$ticks.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ds = [d for d, in c.execute("select duration from kernels where name like '%spin_kernel%' order by start")]
ds = ds[3:]   # warm-ups
ev = sorted(ds[0::2]); od = sorted(ds[1::2])
print("rocprofv3 spin_kernel durations (us): triple launches median %.3f avg %.3f | pair launches median %.3f avg %.3f | first 8: %s" % (
    ev[len(ev)//2]/1e3, sum(ev)/len(ev)/1e3, od[len(od)//2]/1e3, sum(od)/len(od)/1e3, [round(d/1e3, 3) for d in ds[:8]]))
PY
  rm -rf gpurun_out/prof_evclk_$ticks
done
cat gpurun_out/r03_event_clock_plain_3300.txt | tail -3
tail -4 gpurun_out/r03_event_clock_rocprof_3300.txt
tail -4 gpurun_out/r03_event_clock_rocprof_800.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -x > gpurun_out/r03a_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03a_pytest.log
tail -5 gpurun_out/r03a_pytest.log

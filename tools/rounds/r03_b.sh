#!/bin/bash
# round 3, call B: the re-based parity suite (margins ledger), clock calibration under rocprofv3, default bench, self-spawn
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=15 > gpurun_out/r03b_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03b_pytest.log
tail -40 gpurun_out/r03b_pytest.log
./tools/micro/k1gen_micro > gpurun_out/r03b_k1gen_micro.txt 2>&1
cat gpurun_out/r03b_k1gen_micro.txt
cal() { # name, kernel pattern, bench args...
  local name=$1; local pat=$2; shift; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r03_launches_$name.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-extras --no-cpu-baseline > $REPO/gpurun_out/r03b_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/r03b_prof_$name.log)
  local DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python tools/prof_summary.py $DB gpurun_out/r03b_trace_$name.txt > /dev/null
    python tools/clock_calibration.py $DB gpurun_out/r03_launches_$name.json $pat gpurun_out/r03b_clock_calibration_$name.txt
  fi
  rm -rf gpurun_out/prof_$name
}
cal c3 rollout_cost_kernel
cal c4 rollout_mlp_split_kernel --workload c4
cal c2 rollout_cost_kernel --workload c2 --steps 200 --warmup 20
timeout 900 python bench.py > gpurun_out/r03b_bench_default.json 2> gpurun_out/r03b_bench_default.err
python - <<'PY'
import json
for n in ('bench_default', 'bench_under_rocprof_c3', 'bench_under_rocprof_c4', 'bench_under_rocprof_c2'):
    try:
        d = json.load(open('gpurun_out/r03b_%s.json' % n)); r = d['roofline'] or {}
        print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'K1 us %.2f (span %.2f, events %s)' % (r.get('avg_launch_us', 0), r.get('avg_launch_us_device_span', 0), r.get('avg_launch_us_hip_events')),
              'frac %.4f' % r.get('frac', 0), 'cold', r.get('median_launch_us_hbm_cold'), r.get('frac_hbm_cold'), 'synced', d.get('latency_ms_synced', {}).get('median_ms'))
    except Exception as e:
        print(n, 'ERR', e)
PY

#!/bin/bash
# round 3, call I: final evidence on the head -- the default bench under rocprofv3 (kernel table + clock calibration of its three
# regions incl. the HBM-cold launches located by their spans), then the plain default line with the CPU baseline
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r03i_launches_c3.json timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c3i -o c3i -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r03i_bench_under_rocprof_c3.json 2> $REPO/gpurun_out/r03i_prof_c3.log)
DB=$(find gpurun_out/prof_c3i -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r03i_trace_c3.txt > /dev/null
python tools/clock_calibration.py $DB gpurun_out/r03i_launches_c3.json gpurun_out/r03i_clock_calibration_c3.txt
rm -rf gpurun_out/prof_c3i
timeout 900 python bench.py > gpurun_out/r03i_bench_default.json 2> gpurun_out/r03i_bench_default.err
python - <<'PY'
import json
for n in ('bench_default', 'bench_under_rocprof_c3'):
    d = json.load(open('gpurun_out/r03i_%s.json' % n)); r = d['roofline'] or {}; o = d.get('onchip') or {}
    print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], d['config']['draw'], '| onchip K1 us %.2f' % o.get('avg_launch_us', 0), 'stream ms %.4f' % o.get('streaming_form_ms_per_step', 0),
          '| roofline K1 us %.2f frac %.4f cold %s %s' % (r.get('avg_launch_us', 0), r.get('frac', 0), r.get('median_launch_us_hbm_cold'), r.get('frac_hbm_cold')),
          'synced', d.get('latency_ms_synced', {}).get('median_ms'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY

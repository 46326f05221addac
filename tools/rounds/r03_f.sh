#!/bin/bash
# round 3, call F: on-chip tests again, the bench with the on-chip default under rocprofv3 (clock calibration of all its
# regions: headline = on-chip K1, streaming K1, HBM-cold K1), then the plain default line
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_onchip.py tests/test_gpu_sharding.py -m gpu -q --no-header -rf > gpurun_out/r03f_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r03f_pytest.log
tail -15 gpurun_out/r03f_pytest.log
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r03f_launches_c3.json timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c3f -o c3f -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r03f_bench_under_rocprof_c3.json 2> $REPO/gpurun_out/r03f_prof_c3.log)
DB=$(find gpurun_out/prof_c3f -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r03f_trace_c3.txt > /dev/null
python tools/clock_calibration.py $DB gpurun_out/r03f_launches_c3.json gpurun_out/r03f_clock_calibration_c3.txt
rm -rf gpurun_out/prof_c3f
timeout 900 python bench.py > gpurun_out/r03f_bench_default.json 2> gpurun_out/r03f_bench_default.err
tail -2 gpurun_out/r03f_bench_default.err
python - <<'PY'
import json
for n in ('bench_default', 'bench_under_rocprof_c3'):
    try:
        d = json.load(open('gpurun_out/r03f_%s.json' % n)); r = d['roofline'] or {}; o = d.get('onchip') or {}
        print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], d['config']['draw'], '| onchip K1 us', o.get('avg_launch_us'), 'stream ms', o.get('streaming_form_ms_per_step'),
              '| roofline K1 us %.2f frac %.4f cold %s %s' % (r.get('avg_launch_us', 0), r.get('frac', 0), r.get('median_launch_us_hbm_cold'), r.get('frac_hbm_cold')),
              'synced', d.get('latency_ms_synced', {}).get('median_ms'))
        print('   other modes', {k: round(v['ms_per_step'], 4) for k, v in (d.get('other_rng_modes') or {}).items()})
    except Exception as e:
        print(n, 'ERR', e)
PY

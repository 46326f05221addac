#!/bin/bash
# round 4, call A: the suite on the head (staleness guard, KMPPI scratch fix), KMPPI timing, on-chip variants (rows generated
# together), Philox multiply spellings, the default bench under rocprofv3 with the idle gaps of a command
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a_pytest.log
tail -5 gpurun_out/r04a_pytest.log
timeout 120 tools/micro/philox_mul_micro > gpurun_out/r04a_philox_mul_micro.txt 2>&1
for v in head pb12_nta5 pb6_nta5 pb3_nta5; do timeout 60 tools/micro/onchip_parts_$v >> gpurun_out/r04a_onchip_variants.txt 2>&1; done
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/r04a_kmppi_bench.txt 2>&1
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r04a_launches_c3.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c3a -o c3a -- python $REPO/bench.py --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04a_bench_under_rocprof_c3.json 2> $REPO/gpurun_out/r04a_prof_c3.log)
DB=$(find gpurun_out/prof_c3a -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04a_trace_c3.txt > /dev/null
python tools/timeline_gaps.py $DB rollout_onchip_kernel gpurun_out/r04a_timeline_gaps_c3.txt
python tools/clock_calibration.py $DB gpurun_out/r04a_launches_c3.json gpurun_out/r04a_clock_calibration_c3.txt
rm -rf gpurun_out/prof_c3a
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench_default.json 2> gpurun_out/r04a_bench_default.err
cat gpurun_out/r04a_philox_mul_micro.txt gpurun_out/r04a_onchip_variants.txt gpurun_out/r04a_kmppi_bench.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04a_bench_default.json'))
print('default ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), 'onchip K1 us', (d.get('onchip') or {}).get('avg_launch_us'),
      'roofline', {k: d['roofline'][k] for k in ('frac', 'avg_launch_us', 'frac_hbm_cold') if k in d['roofline']})
print('family', d.get('controller_family_on_c3_shape')); print('rng modes', {k: round(v['ms_per_step'], 4) for k, v in d.get('other_rng_modes', {}).items()})
print('others', {k: (round(v['ms_per_step'], 4), round(v['k1_avg_us'], 1)) for k, v in d.get('other_workloads', {}).items()})
PY

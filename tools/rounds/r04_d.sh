#!/bin/bash
# round 4, call D: the head after the on-chip one-code-path change and the matrix-core execution of traced networks -- suite, default
# bench (+ under rocprofv3 with the calibration on the same launches), instruction-class counters of the on-chip K1 and of the split
# MLP kernel, the learned-dynamics network at several K and its kernel table
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04d_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04d_pytest.log
tail -3 gpurun_out/r04d_pytest.log
run_prof() {
  name=$1; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r04d_launches_$name.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04d_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/r04d_prof_$name.log)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/r04d_trace_$name.txt > /dev/null
  python tools/clock_calibration.py $DB gpurun_out/r04d_launches_$name.json gpurun_out/r04d_clock_calibration_$name.txt
  rm -rf gpurun_out/prof_$name
}
run_pmc() {
  name=$1; ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04d_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/r04d_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
run_prof c3
run_prof c4 --workload c4
run_prof c2 --workload c2
run_pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run_pmc c3_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run_pmc c3_classes "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
run_pmc c4_classes "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU" --workload c4
run_pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" --workload c4
timeout 300 python tools/learned_bench.py > gpurun_out/r04d_learned_bench.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_learned -o learned -- python $REPO/tools/learned_trace_one.py > /dev/null 2>&1)
DB=$(find gpurun_out/prof_learned -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04d_trace_learned_8192x32.txt > /dev/null
python tools/timeline_gaps.py $DB noise_fill_philox gpurun_out/r04d_timeline_gaps_learned.txt > /dev/null
rm -rf gpurun_out/prof_learned
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/r04d_kmppi_bench.txt 2>&1
timeout 300 python tools/variants_bench.py philox > gpurun_out/r04d_variants_philox.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04d_bench_default.json 2> gpurun_out/r04d_bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --workload c4 > gpurun_out/r04d_bench_c4.json 2> gpurun_out/r04d_bench_c4.err
timeout 600 python bench.py --steps 100 --warmup 5 --workload c2 > gpurun_out/r04d_bench_c2.json 2> gpurun_out/r04d_bench_c2.err
for n in c3 c4 c2; do tail -2 gpurun_out/r04d_clock_calibration_$n.txt; done
grep -v amdgpu gpurun_out/r04d_learned_bench.txt; head -9 gpurun_out/r04d_trace_learned_8192x32.txt | cut -c1-190; head -8 gpurun_out/r04d_timeline_gaps_learned.txt
grep -E "rollout_onchip|rollout_mlp_split" gpurun_out/r04d_pmc_c3_classes.txt gpurun_out/r04d_pmc_c4_classes.txt gpurun_out/r04d_pmc_c4_mfma.txt | cut -c1-150
python - <<'PY'
import json
for n in ('default', 'c4', 'c2'):
    d = json.load(open('gpurun_out/r04d_bench_%s.json' % n)); r = d['roofline'] or {}
    print(n, 'ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), 'clock warm-up', d.get('clock_warmup_commands'), 'roofline frac %.3f K1 %.1f us' % (r.get('frac', 0), r.get('avg_launch_us', 0)),
          'cold', r.get('frac_hbm_cold'), 'onchip', (d.get('onchip') or {}).get('avg_launch_us'), ((d.get('onchip') or {}).get('roofline') or {}).get('frac'))
d = json.load(open('gpurun_out/r04d_bench_default.json'))
print('family', d.get('controller_family_on_c3_shape')); print('rng modes', {k: round(v['ms_per_step'], 4) for k, v in d.get('other_rng_modes', {}).items()})
print('others', {k: (round(v['ms_per_step'], 4), round(v['k1_avg_us'], 1)) for k, v in d.get('other_workloads', {}).items()})
PY
cat gpurun_out/r04d_variants_philox.txt | grep -v amdgpu | tail -12

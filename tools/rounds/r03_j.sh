#!/bin/bash
# round 3, call J: C4 evidence after the split kernel's VALU trims (v_fma_mix_f32 residual, layer-1 weights pinned in AGPRs):
# bench under rocprofv3 (kernel table + clock calibration), matrix-pipe busy PMC pass, clocks / power while it runs
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/r03j_smi.txt 2>&1 &
SMI=$!
(cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/r03j_launches_c4.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c4j -o c4j -- python $REPO/bench.py --workload c4 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r03j_bench_under_rocprof_c4.json 2> $REPO/gpurun_out/r03j_prof_c4.log)
DB=$(find gpurun_out/prof_c4j -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r03j_trace_c4.txt > /dev/null
python tools/clock_calibration.py $DB gpurun_out/r03j_launches_c4.json gpurun_out/r03j_clock_calibration_c4.txt
rm -rf gpurun_out/prof_c4j
timeout 300 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r03j_bench_c4.json 2> gpurun_out/r03j_bench_c4.err
kill $SMI
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc_c4j -o c4j -- python $REPO/bench.py --workload c4 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r03j_pmc_c4.log 2>&1)
DB=$(find gpurun_out/pmc_c4j -name "*.db" | head -1)
[ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/r03j_pmc_c4_mfma.txt > /dev/null
rm -rf gpurun_out/pmc_c4j
sort gpurun_out/r03j_smi.txt | uniq -c | sort -rn | head -6 > gpurun_out/r03j_smi_summary.txt
python - <<'PY'
import json
for n in ('bench_c4', 'bench_under_rocprof_c4'):
    d = json.load(open('gpurun_out/r03j_%s.json' % n)); r = d['roofline'] or {}
    print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], '| K1 us %.1f frac %.3f of %s, algorithmic %.1f TF = %.3f of fp32 MFMA peak' % (
        r.get('avg_launch_us', 0), r.get('frac', 0), r.get('peak'), r.get('algorithmic_tflops', 0), r.get('algorithmic_frac_of_fp32_mfma_peak', 0)),
        'synced', (d.get('latency_ms_synced') or {}).get('median_ms'))
PY
tail -3 gpurun_out/r03j_clock_calibration_c4.txt; grep -E "rollout_mlp" gpurun_out/r03j_trace_c4.txt | head -3; tail -5 gpurun_out/r03j_pmc_c4_mfma.txt; cat gpurun_out/r03j_smi_summary.txt

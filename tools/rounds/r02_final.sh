#!/bin/bash
# round-2 final measurements: suite, default bench, per-workload benches, rocprofv3 traces, PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
T=${TAG:-final}
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --durations=12 > gpurun_out/r02${T}_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02${T}_pytest.log
timeout 600 python bench.py > gpurun_out/r02${T}_bench_default.json 2> gpurun_out/r02${T}_bench_default.err
timeout 300 python bench.py --workload c4 --no-extras --no-cpu-baseline > gpurun_out/r02${T}_bench_c4.json 2>> gpurun_out/r02${T}_bench_default.err
MPPI_MLP_EXACT=1 timeout 300 python bench.py --workload c4 --no-extras --no-cpu-baseline > gpurun_out/r02${T}_bench_c4_exact.json 2>> gpurun_out/r02${T}_bench_default.err
timeout 300 python bench.py --workload c2 --steps 300 --warmup 30 --no-extras --no-cpu-baseline > gpurun_out/r02${T}_bench_c2.json 2>> gpurun_out/r02${T}_bench_default.err
prof() { # name, command...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- "$@" > $REPO/gpurun_out/r02${T}_prof_$name.log 2>&1)
  local DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/prof_summary.py $DB gpurun_out/r02${T}_trace_$name.txt > /dev/null
  rm -rf gpurun_out/prof_$name          # the .db files are tens of MB each: only the summaries travel back
}
prof c3 python $REPO/bench.py --no-extras --no-cpu-baseline
prof c4 python $REPO/bench.py --workload c4 --no-extras --no-cpu-baseline
prof c2 python $REPO/bench.py --workload c2 --steps 300 --warmup 30 --no-extras --no-cpu-baseline
ONLY_COLD=1 prof k1cold_65536 python $REPO/tools/k1_sweep.py
K=262144 ONLY_COLD=1 prof k1cold_262144 python $REPO/tools/k1_sweep.py
pmc() { # name, counters, command...
  local name=$1; local ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- "$@" > $REPO/gpurun_out/r02${T}_pmc_$name.log 2>&1)
  local DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/r02${T}_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
pmc c3_fetch FETCH_SIZE python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
pmc c3_write WRITE_SIZE python $REPO/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python $REPO/bench.py --workload c4 --steps 6 --warmup 2 --no-extras --no-cpu-baseline
timeout 300 python tools/variants_bench.py philox > gpurun_out/r02${T}_variants.txt 2>&1
timeout 300 python tools/variants_bench.py torch >> gpurun_out/r02${T}_variants.txt 2>&1
timeout 120 python tools/small_cmd_breakdown.py > gpurun_out/r02${T}_small.txt 2>&1
tail -22 gpurun_out/r02${T}_pytest.log | head -30
python - <<PY
import json
for n in ('bench_default','bench_c4','bench_c4_exact','bench_c2'):
    try:
        d=json.load(open('gpurun_out/r02${T}_%s.json' % n)); r=d['roofline'] or {}
        print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'K1 us %.1f' % r.get('avg_launch_us', 0), 'frac %.3f' % r.get('frac', 0), 'cold', r.get('avg_launch_us_hbm_cold'), r.get('frac_hbm_cold'))
    except Exception as e: print(n, 'ERR', e)
PY
for f in c3 c4 c2 k1cold_65536 k1cold_262144; do head -7 gpurun_out/r02${T}_trace_$f.txt | cut -c1-170; done
grep -v amdgpu gpurun_out/r02${T}_variants.txt | head -24

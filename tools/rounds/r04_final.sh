#!/bin/bash
# round 4, final evidence on the head: suite, default bench (driver's protocol), the same under rocprofv3 with the clock calibration on
# the same launches, rng="torch" bench + kernel table, C2 / C4 lines, KMPPI / family / learned-network tools
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=r04_final2
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -3 gpurun_out/${P}_pytest.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
run_prof() {
  name=$1; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/${P}_launches_$name.json timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/${P}_prof_$name.log)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/clock_calibration.py $DB gpurun_out/${P}_launches_$name.json gpurun_out/${P}_clock_calibration_$name.txt
  rm -rf gpurun_out/prof_$name
}
run_prof c3
run_prof torch --rng torch --steps 300 --warmup 30
python bench.py --rng torch --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/${P}_bench_torch.json
python bench.py --workload c4 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c4.json
python bench.py --workload c2 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c2.json
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/${P}_kmppi_bench.txt 2>&1
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
timeout 300 python tools/variants_bench.py torch > gpurun_out/${P}_variants_torch.txt 2>&1
timeout 300 python tools/learned_bench.py > gpurun_out/${P}_learned_bench.txt 2>&1
python - <<'PY'
import json
P="gpurun_out/r04_final2"
for n in ("default","torch","c4","c2"):
    d=json.load(open(f"{P}_bench_{n}.json")); print(n, d["ms_per_step"], d["value"])
d=json.load(open(f"{P}_bench_default.json"))
print("roofline", d["roofline"]["frac"], d["roofline"].get("frac_hbm_cold"), "onchip", d.get("onchip",{}).get("k1_avg_us"), d.get("onchip",{}).get("roofline"))
print("other", d.get("other_rng_modes"))
print({k:v for k,v in d.items() if k in ("family","other_workloads","latency_ms_synced","cpu_baseline")})
PY

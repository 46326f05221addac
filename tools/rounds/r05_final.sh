#!/bin/bash
# round 5, evidence on the head: the -m gpu suite, default bench (driver's protocol), the same under rocprofv3 with the clock calibration on
# the same launches (headline, streaming, HBM-cold regions), rng="torch" bench + kernel table + idle gaps, C2 / C4 lines, PMC passes of
# the headline command (VALU, waits, classes, FETCH_SIZE, WRITE_SIZE) and of the rng="torch" command, family / KMPPI / learned tools,
# bench --gpus 2 in one process (device group rig) and with self-started ranks
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r05_final}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -4 gpurun_out/${P}_pytest.log
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
run_prof() {
  name=$1; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/${P}_launches_$name.json timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/${P}_prof_$name.log)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/clock_calibration.py $DB gpurun_out/${P}_launches_$name.json gpurun_out/${P}_clock_calibration_$name.txt
  python tools/timeline_gaps.py $DB $TLK gpurun_out/${P}_timeline_gaps_$name.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name
}
TLK=rollout_onchip_kernel run_prof c3 --steps 20 --warmup 5 --no-extras --hbm-cold
TLK=rollout_cost_kernel run_prof torch --rng torch --steps 300 --warmup 30 --no-extras
python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/${P}_bench_torch.json
MPPI_DRAW_AHEAD=0 python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/${P}_bench_torch_no_draw_ahead.json
python bench.py --workload c4 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c4.json
python bench.py --workload c2 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c2.json
run_pmc() {
  name=$1; ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
run_pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run_pmc c3_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run_pmc c3_fetch "FETCH_SIZE"
run_pmc c3_write "WRITE_SIZE"
run_pmc torch_fetch "FETCH_SIZE" --rng torch
run_pmc torch_write "WRITE_SIZE" --rng torch
run_pmc torch_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --rng torch
run_pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c4
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/${P}_kmppi_bench.txt 2>&1
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
timeout 300 python tools/variants_bench.py torch > gpurun_out/${P}_variants_torch.txt 2>&1
timeout 300 python tools/learned_bench.py > gpurun_out/${P}_learned_bench.txt 2>&1
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_gpus2_one_process.json
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --process-model spawn 2>/dev/null | tail -1 > gpurun_out/${P}_bench_gpus2_spawn.json
python - <<PY
import json
P="gpurun_out/$P"
for n in ("default","torch","torch_no_draw_ahead","c4","c2","gpus2_one_process","gpus2_spawn"):
    try:
        d=json.load(open(f"{P}_bench_{n}.json")); print(n, d["ms_per_step"], d["value"])
    except Exception as e:
        print(n, "FAILED", e)
d=json.load(open(f"{P}_bench_default.json"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","bound","frac","avg_launch_us","hbm_equiv_frac_k1","hbm_equiv_frac_cmd")})
print("streaming", d["streaming"]["ms_per_step"], {k: d["streaming"]["roofline"].get(k) for k in ("frac","frac_hbm_cold","avg_launch_us","median_launch_us_hbm_cold")})
print("synced", d.get("value_synced"), d.get("latency_ms_synced",{}).get("median_ms"))
print("other", d.get("other_rng_modes"))
print("c4", {k: v for k, v in d.get("other_workloads",{}).get("c4",{}).items() if k.startswith("k1") or k in ("ms_per_step","mfma_busy","valu_issue_frac")})
print("fam", d.get("controller_family_on_c3_shape"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
grep -h "rollout_onchip\|weights_partial_rows\|rollout_mlp_split" gpurun_out/${P}_pmc_*.txt | cut -c1-170

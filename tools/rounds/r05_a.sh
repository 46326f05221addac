#!/bin/bash
# round 5, first GPU call: the draw-ahead K3 launch (ABI 21) -- tests, rng="torch" bench with / without it, R sweep, kernel table;
# Philox 10 vs 7 rounds in the product on-chip K1 (tools/micro/onchip_parts.hip); the default line
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=r05_a
timeout 900 python -m pytest tests/test_gpu_torch_stream.py tests/test_abi.py -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -15 gpurun_out/${P}_pytest.log
for mode in ahead noahead; do
  for R in 0 2 4 1; do
    if [ $mode = noahead ] && [ $R != 0 ]; then continue; fi
    export MPPI_K3_R=$R; [ $R = 0 ] && unset MPPI_K3_R
    if [ $mode = noahead ]; then export MPPI_DRAW_AHEAD=0; else unset MPPI_DRAW_AHEAD; fi
    python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/${P}_bench_torch_${mode}_R$R.json
    python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_torch_${mode}_R$R.json")); print("torch", "$mode", "R=$R", d["ms_per_step"], d["config"].get("draw"), d.get("latency_ms_synced",{}).get("median_ms"))
PY
  done
done
unset MPPI_K3_R MPPI_DRAW_AHEAD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_torch -o torch -- python $REPO/bench.py --rng torch --steps 300 --warmup 30 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_torch.json 2> $REPO/gpurun_out/${P}_prof_torch.log)
DB=$(find gpurun_out/prof_torch -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/${P}_trace_torch.txt > /dev/null
python tools/timeline_gaps.py $DB rollout_cost_kernel gpurun_out/${P}_timeline_gaps_torch.txt > /dev/null 2>&1
rm -rf gpurun_out/prof_torch
head -25 gpurun_out/${P}_trace_torch.txt
for r in 10 7; do
  MPPI_MICRO_SPILL=1 ./tools/micro/onchip_parts_r$r 65536 >> gpurun_out/${P}_philox_rounds.txt 2>&1
  ./tools/micro/onchip_parts_r$r 65536 >> gpurun_out/${P}_philox_rounds.txt 2>&1
done
cat gpurun_out/${P}_philox_rounds.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_a_bench_default.json"))
print("default", d["ms_per_step"], d["value"], "synced", d.get("value_synced"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","bound","frac","avg_launch_us","hbm_equiv_frac_k1","hbm_equiv_frac_cmd")})
print("streaming", d["streaming"]["ms_per_step"], {k: d["streaming"]["roofline"].get(k) for k in ("frac","frac_hbm_cold","avg_launch_us")})
print("other", d.get("other_rng_modes"))
print("c4", d.get("other_workloads",{}).get("c4"))
PY

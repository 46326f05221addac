#!/bin/bash
# round 5: the draw-ahead K3 launch after unifying the kernels + two Philox chains per generator wave
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r05_b}
timeout 900 python -m pytest tests/test_gpu_torch_stream.py -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -8 gpurun_out/${P}_pytest.log
python tools/diag_draw_ahead.py 2>&1 | grep -v "differ: \[\]" | tail -12
for X in 2 4 8 16 32; do
  MPPI_NEXT_GEN_PER_K3=$X python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('gen per k3 =', $X, 'R=4', d['ms_per_step'])"
  MPPI_K3_R=2 MPPI_NEXT_GEN_PER_K3=$X python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('gen per k3 =', $X, 'R=2', d['ms_per_step'])"
done
for mode in ahead noahead; do
  for R in 0 2; do
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
head -9 gpurun_out/${P}_trace_torch.txt | cut -c1-200

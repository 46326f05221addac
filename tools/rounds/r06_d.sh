#!/bin/bash
# round 6, fourth GPU session: the -m gpu suite (distribution-judged MLP shapes, re-armed device group, one-wave-per-copy M > 1),
# the device group's host share after the re-arm, M = 3 under rocprofv3 in both forms, bench --gpus 2 --process-model auto
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r06_d}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
grep -n "^FAILED\|passed\|failed" gpurun_out/${P}_pytest.log | tail -12 | cut -c1-300
timeout 900 python tools/group_host_issue.py gpurun_out/${P}_group_host_issue.txt 2>&1 | tail -14
trace() {
  name=$1; pat=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python "$@" > $REPO/gpurun_out/${P}_run_$name.log 2>&1)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/timeline_gaps.py $DB $pat gpurun_out/${P}_gaps_$name.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name
}
VARIANTS_N=100 trace m3_copies rollout_copies_kernel $REPO/tools/variants_bench.py philox "M=3"
MPPI_MULTI_COPIES=0 VARIANTS_N=100 trace m3_one_lane rollout_cost_kernel $REPO/tools/variants_bench.py philox "M=3"
for n in m3_copies m3_one_lane; do echo "== $n"; head -8 gpurun_out/${P}_trace_$n.txt | cut -c1-220; grep -v amdgpu gpurun_out/${P}_run_$n.log | tail -2; done
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --process-model auto 2>gpurun_out/${P}_bench_gpus2_auto.err | tail -1 > gpurun_out/${P}_bench_gpus2_auto.json
python - <<PY
import json
g=json.load(open("gpurun_out/${P}_bench_gpus2_auto.json")); print("gpus2 auto", g["ms_per_step"], g["value"], g["config"].get("process_model", "")[:100], g["config"].get("process_model_choice"))
PY

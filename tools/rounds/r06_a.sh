#!/bin/bash
# round 6, first GPU session: the device group's worker threads (tests + host-issue table), and where three commands' time goes
# (rng="torch" with bounds + null action against plain; KMPPI S=32 over 200 commands; M=3)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=r06_a
timeout 900 python -m pytest tests/test_gpu_devices.py tests/test_gpu_group_threads.py -x -q > gpurun_out/${P}_pytest_group.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest_group.log
tail -15 gpurun_out/${P}_pytest_group.log
timeout 900 python tools/group_host_issue.py gpurun_out/${P}_group_host_issue.txt 2>&1 | tail -20
trace() {
  name=$1; pat=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python "$@" > $REPO/gpurun_out/${P}_run_$name.log 2>&1)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/timeline_gaps.py $DB $pat gpurun_out/${P}_gaps_$name.txt > gpurun_out/${P}_gaps_$name.head.txt 2>&1
  rm -rf gpurun_out/prof_$name
}
trace torch_plain rollout_cost_kernel $REPO/tools/variants_bench.py torch "=MPPI"
trace torch_bounds rollout_cost_kernel $REPO/tools/variants_bench.py torch "bounds"
trace philox_m3 rollout_cost_kernel $REPO/tools/variants_bench.py philox "M=3"
VARIANTS_N=200 trace kmppi rollout_kmppi $REPO/tools/variants_bench.py philox "KMPPI"
timeout 300 python tools/variants_bench.py torch > gpurun_out/${P}_variants_torch.txt 2>&1
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/${P}_kmppi_bench.txt 2>&1
cat gpurun_out/${P}_variants_torch.txt gpurun_out/${P}_variants_philox.txt gpurun_out/${P}_kmppi_bench.txt | grep -v amdgpu.ids
for n in torch_plain torch_bounds philox_m3 kmppi; do echo "== $n"; head -14 gpurun_out/${P}_trace_$n.txt | cut -c1-200; cat gpurun_out/${P}_gaps_$n.head.txt | cut -c1-200; done
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_default.json")); print("bench", d["ms_per_step"], d["value"], d.get("other_rng_modes"))
PY

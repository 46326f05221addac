#!/bin/bash
# round 6: the pair kernel after the unified weighting loop: micro check at the defaults, the on-chip tests, the bench default
mkdir -p gpurun_out
P=${1:-r06_l}
O=gpurun_out/${P}_onchip_pair_check.txt
: > $O
for a in "65536 20000 0 64 0.001" "65536 20000 1 48 0.001" "60000 40 1 64 0.05" "50000 20000 1 33 0.001" "49152 20000 0 100 0.001"; do
  echo "== default, $a" >> $O
  timeout 120 tools/micro/onchip_pair_check_prod $a >> $O 2>&1
done
grep -E "^==|identical|MISMATCH|two waves|one wave|failed" $O
timeout 1200 python -m pytest tests/test_gpu_onchip.py -m gpu -q > gpurun_out/${P}_pytest_onchip.log 2>&1; echo "rc $?" >> gpurun_out/${P}_pytest_onchip.log
grep -E "passed|failed|rc |^FAILED|Error" gpurun_out/${P}_pytest_onchip.log | tail -8
for m in 1 0 1; do
  MPPI_ONCHIP_PAIR=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${P}_bench_pair$m.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_pair$m.json"))
r=d["roofline"]
print("pair=$m", "ms", round(d["ms_per_step"],5), "value", "%.4g"%d["value"], "kernel", r.get("kernel"), "k1_us", r.get("avg_launch_us"), "frac", round(r.get("frac"),4), "synced", d.get("latency_ms_synced",{}).get("median_ms"))
PY
done

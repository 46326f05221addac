#!/bin/bash
# round 6: the two-wave on-chip K1 in the product: the on-chip tests (incl. the new bit-for-bit ones), the bench default with the
# two-wave kernel and with MPPI_ONCHIP_PAIR=0
mkdir -p gpurun_out
P=${1:-r06_i}
timeout 1200 python -m pytest tests/test_gpu_onchip.py tests/test_gpu_fullsize_parity.py -m gpu -q -x > gpurun_out/${P}_pytest_onchip.log 2>&1; echo "rc $?" >> gpurun_out/${P}_pytest_onchip.log
grep -E "passed|failed|rc |^FAILED|Error" gpurun_out/${P}_pytest_onchip.log | tail -8
for m in 1 0 1 0; do
  MPPI_ONCHIP_PAIR=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${P}_bench_pair$m.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_pair$m.json"))
r=d["roofline"]
print("pair=$m", "ms", round(d["ms_per_step"],5), "value", "%.4g"%d["value"], "kernel", r.get("kernel"), "k1_us", r.get("avg_launch_us"), "frac", round(r.get("frac"),4), "synced", d.get("latency_ms_synced",{}).get("median_ms"))
PY
done
for m in 1 0; do MPPI_ONCHIP_PAIR=$m timeout 300 python tools/variants_bench.py philox "SMPPI" 2>&1 | grep -v amdgpu | sed "s/^/pair=$m /"; done
for m in 1 0; do MPPI_ONCHIP_PAIR=$m timeout 300 python tools/variants_bench.py philox "MPPI" 2>&1 | grep -v amdgpu | sed "s/^/pair=$m /"; done

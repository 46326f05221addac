#!/bin/bash
# round 6: the -m gpu suite and the driver's bench command at the head (after the trace.py split and the lower on-chip threshold)
mkdir -p gpurun_out
P=${1:-r06_n}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
grep -E "passed|failed|pytest rc|^FAILED|^ERROR" gpurun_out/${P}_pytest.log | tail -12 | cut -c1-300
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null
for i in 1 2 3; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default_$i.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_default_$i.json"))
r=d["roofline"]
print("run $i", "ms", round(d["ms_per_step"],5), "value", "%.4g"%d["value"], r.get("kernel"), "k1_us", round(r.get("avg_launch_us"),2), "frac", round(r.get("frac"),4), "stale", d.get("lookup_stale"), "small", {k: round(v["ms_per_step"],4) for k,v in d.get("small_sizes_on_c3_shape",{}).items()})
PY
done
python -c "from __graft_entry__ import smoke; smoke(); print('smoke ok')" 2>&1 | tail -1

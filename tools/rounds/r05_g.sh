#!/bin/bash
# round 5: engine-Philox draw-ahead -- tests, small / mid K sweep with and without it, the default line
mkdir -p gpurun_out
export TMPDIR=/tmp
P=${1:-r05_g}
timeout 900 python -m pytest tests/test_gpu_torch_stream.py -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -6 gpurun_out/${P}_pytest.log
python tools/small_k_sweep.py > gpurun_out/${P}_small_k_sweep.txt 2>&1
MPPI_DRAW_AHEAD=0 python tools/small_k_sweep.py > gpurun_out/${P}_small_k_sweep_noahead.txt 2>&1
paste -d'|' <(cut -c1-95 gpurun_out/${P}_small_k_sweep.txt) <(cut -c40-95 gpurun_out/${P}_small_k_sweep_noahead.txt) | head -50
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
python - <<PY
import json
d=json.load(open("gpurun_out/${P}_bench_default.json"))
print("default", d["ms_per_step"], d["value"], "synced", d.get("value_synced"))
print("other", d.get("other_rng_modes"))
print("fam", d.get("controller_family_on_c3_shape"))
PY

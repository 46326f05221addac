#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
P=${1:-r05_i}
timeout 1200 python -m pytest tests/test_gpu_from_torch.py -x -q -k "c4_shaped or learned_dynamics" > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -15 gpurun_out/${P}_pytest.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/parity_margins.json"))
for e in d["all"]:
    if "c4-shaped" in e["test"] or "learned" in e["test"]:
        print(e["test"], e["quantity"], e["err_over_scale"], e["floor_over_scale"], e.get("note"))
PY

#!/bin/bash
# round 5: full -m gpu suite on the head (draw-ahead K3 launch, devices= groups), then the SMPPI-over-MLP margin under the three K1 kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
P=${1:-r05_e}
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -12 gpurun_out/${P}_pytest.log
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null
for mode in split EXACT VALU; do
  ( [ $mode != split ] && export MPPI_MLP_$mode=1; timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py -q -k "smppi_with_mlp" > gpurun_out/${P}_smppi_mlp_$mode.log 2>&1; cp gpurun_out/parity_margins.json gpurun_out/${P}_smppi_mlp_margins_$mode.json )
  python - <<PY
import json
d=json.load(open("gpurun_out/${P}_smppi_mlp_margins_$mode.json"))
for e in d["all"]:
    if e["floor_over_scale"]:
        print("$mode", e["test"], e["quantity"], "err %.3g floor %.3g ratio %.2f" % (e["err_over_scale"], e["floor_over_scale"], e["err_over_scale"]/e["floor_over_scale"]))
PY
done

#!/bin/bash
mkdir -p gpurun_out
T=${TAG:-r}
timeout 1200 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02${T}_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02${T}_pytest.log
tail -8 gpurun_out/r02${T}_pytest.log

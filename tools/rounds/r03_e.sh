#!/bin/bash
# round 3, call E: whole GPU suite with the on-chip command as the default draw of large rng="philox" problems
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=10 > gpurun_out/r03e_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03e_pytest.log
tail -60 gpurun_out/r03e_pytest.log

#!/bin/bash
# round 6: the whole -m gpu suite on the head with the two-wave on-chip K1 as the default
mkdir -p gpurun_out
P=${1:-r06_j}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
grep -E "passed|failed|pytest rc|^FAILED|^ERROR" gpurun_out/${P}_pytest.log | tail -12 | cut -c1-300
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null

#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/all_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python tools/host_state_cost.py 2>&1 | grep "state as" | tee gpurun_out/host_state_cost.txt
timeout 300 python tools/kmppi_bench.py philox 2>&1 | grep KMPPI | tee gpurun_out/kmppi_bench6.txt

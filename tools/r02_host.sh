#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
timeout 600 python -m pytest tests/test_gpu_edge_semantics.py tests/test_gpu_api_behaviour.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python tools/host_state_cost.py 2>&1 | grep "state as" | tee gpurun_out/host_state_cost2.txt

#!/bin/bash
# round 6, second GPU session: device-group tests + host-issue table on the in-place exchange, the rng="torch" bounds / null-action
# question, and the seed distributions of the parity margins above 1e-5 (32 seeds, MLP scenarios under both matrix-core kernels)
mkdir -p gpurun_out
export TMPDIR=/tmp
P=r06_b
timeout 900 python -m pytest tests/test_gpu_devices.py tests/test_gpu_group_threads.py -q > gpurun_out/${P}_pytest_group.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest_group.log
tail -12 gpurun_out/${P}_pytest_group.log
timeout 900 python tools/group_host_issue.py gpurun_out/${P}_group_host_issue.txt 2>&1 | tail -20
timeout 1200 python tools/diag_torch_bounds.py gpurun_out/${P}_diag_torch_bounds.txt 2>&1 | tail -24
timeout 3600 python tools/margin_distributions.py ${SEEDS:-32} gpurun_out/r06_margin_distributions > gpurun_out/${P}_margin_distributions.log 2>&1
tail -80 gpurun_out/${P}_margin_distributions.log

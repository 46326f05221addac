#!/bin/bash
# round 6: the randomised parity sweeps with further seeds at the round's head (generic path, fused path, the on-chip command), and the
# device-group / worker-thread tests repeated (a soak of the submit / wait / re-arm protocol)
mkdir -p gpurun_out
P=${1:-r06_sweep}
N=${2:-300}
MPPI_EXTRA_SEEDS=$N timeout 1500 python -m pytest tests/test_gpu_random_configs.py tests/test_gpu_onchip.py -m gpu -q -k 'random or sweep or seed' > gpurun_out/${P}_random.log 2>&1
echo "rc $?" >> gpurun_out/${P}_random.log
tail -3 gpurun_out/${P}_random.log | cut -c1-300
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_group_threads.py tests/test_gpu_devices.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed"
done > gpurun_out/${P}_group_soak.log 2>&1
cat gpurun_out/${P}_group_soak.log

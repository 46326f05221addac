#!/bin/bash
# round 3, call G: on-chip with a full Sigma -- tests, then the controller family at C3-sized work (rng=philox)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_onchip.py tests/test_gpu_fullsize_parity.py -k "onchip or on_chip or full_sigma" -m gpu -q --no-header -rf > gpurun_out/r03g_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r03g_pytest.log
tail -12 gpurun_out/r03g_pytest.log
timeout 600 python tools/variants_bench.py philox > gpurun_out/r03g_variants_philox.txt 2>&1
cat gpurun_out/r03g_variants_philox.txt | grep -v amdgpu.ids

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02j_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02j_pytest.log
timeout 300 python tools/shard_overhead.py c3 philox > gpurun_out/r02j_shard_overhead.txt 2>&1
timeout 300 python tools/shard_overhead.py c4 philox >> gpurun_out/r02j_shard_overhead.txt 2>&1
tail -8 gpurun_out/r02j_pytest.log; grep -v "^$\|amdgpu.ids" gpurun_out/r02j_shard_overhead.txt | tail -20

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
P=${1:-r05_j}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_onchip.py tests/test_kernel_resources.py -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -6 gpurun_out/${P}_pytest.log
for rng in philox philox7 philox philox7; do
python bench.py --rng $rng --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$rng', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
done

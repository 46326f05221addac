#!/bin/bash
# round 5: the run-time round count of the engine's generator as a rolled loop -- what does it cost the forms that generate in the lane?
mkdir -p gpurun_out
export TMPDIR=/tmp
P=${1:-r05_k}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_onchip.py -x -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -4 gpurun_out/${P}_pytest.log
for rng in philox-fused philox-stream philox philox7; do
python bench.py --rng $rng --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$rng', d['ms_per_step'], d['value'], d['config']['draw'])"
done
for i in 1 2; do python bench.py --workload c2 --no-extras --no-cpu-baseline --steps 500 --warmup 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c2', d['ms_per_step'], d['config']['draw'])"; done
python tools/small_k_sweep.py 2>&1 | grep "rng=philox" | grep -v "nx=16" | cut -c1-110

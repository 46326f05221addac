#!/bin/bash
# round 4: on-chip command with the rows that do not fit waiting in memory (ABI 20) -- on-chip tests, default bench with / without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04i
python -m pytest tests/test_gpu_onchip.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04i/tests_onchip.txt
for sp in 1 0 1 0; do
  MPPI_ONCHIP_SPILL=$sp python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r04i/bench_spill${sp}.json
  python -c "
import json; d=json.load(open('gpurun_out/r04i/bench_spill${sp}.json')); print('spill', $sp, d['ms_per_step'], d['value'], d.get('onchip',{}).get('k1_avg_us'))"
done
cat gpurun_out/r04i/tests_onchip.txt

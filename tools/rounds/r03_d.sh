#!/bin/bash
# round 3, call D: the on-chip command -- its tests, then the bench line with it as the default draw
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_onchip.py tests/test_gpu_fullsize_parity.py -k "onchip or on_chip or c3_quadtoy or two_shards" -q --no-header -x -rf > gpurun_out/r03d_pytest_onchip.log 2>&1
echo "rc=$?" >> gpurun_out/r03d_pytest_onchip.log
tail -30 gpurun_out/r03d_pytest_onchip.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03d_bench_default.json 2> gpurun_out/r03d_bench_default.err
tail -3 gpurun_out/r03d_bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03d_bench_default.json'))
print('ms/step', d['ms_per_step'], 'value %.4g' % d['value'], d['config'].get('draw'), 'synced', d.get('latency_ms_synced', {}).get('median_ms'))
print(json.dumps(d.get('other_rng_modes'), indent=1))
print(json.dumps(d.get('roofline'), indent=1)[:1500])
PY

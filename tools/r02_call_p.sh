#!/bin/bash
mkdir -p gpurun_out
T=${TAG:-p}
timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02${T}_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02${T}_pytest.log
timeout 120 python tools/small_cmd_breakdown.py > gpurun_out/r02${T}_small.txt 2>&1
timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r02${T}_bench_c2.json 2> gpurun_out/r02${T}_bench_c2.err
tail -3 gpurun_out/r02${T}_pytest.log; grep "C2 single" gpurun_out/r02${T}_small.txt; python - <<PY
import json
d=json.load(open('gpurun_out/r02${T}_bench_c2.json')); r=d['roofline']
print('C2 ms/step',d['ms_per_step'],'K1 us',r and r['avg_launch_us'])
PY

#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
timeout 600 python -m pytest tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02d_bench_default.json 2> gpurun_out/r02d_bench_default.err; tail -3 gpurun_out/r02d_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d_bench_default.json'))
print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('frac_hbm_cold'))
print(d.get('controller_family_on_c3_shape'))
print({k:v['ms_per_step'] for k,v in d['other_workloads'].items()})
PY

#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/all_tests.txt
cat gpurun_out/all_tests.txt
timeout 300 python tools/kmppi_bench.py philox 2>&1 | grep KMPPI | tee gpurun_out/kmppi_bench4.txt
timeout 300 python tools/variants_bench.py philox > gpurun_out/r02c_variants.txt 2>&1
timeout 300 python tools/variants_bench.py torch >> gpurun_out/r02c_variants.txt 2>&1
grep "ms/command" gpurun_out/r02c_variants.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02c_bench_default.json 2> gpurun_out/r02c_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c_bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('frac_hbm_cold'))
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('other_workloads',{}).items()})
PY

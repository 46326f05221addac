#!/bin/bash
# round 4, call B: the suite on the head (KMPPI theta update inside K1, sharded stamp slots), KMPPI A/B, wider generation batches
# of the on-chip K1, fixed cost of a timed region, default bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04b_pytest.log
tail -4 gpurun_out/r04b_pytest.log
for v in head pb12_nta5 pb15_nta5 pb18_nta5 pb18_nta4 pb24_nta4; do timeout 60 tools/micro/onchip_parts_$v >> gpurun_out/r04b_onchip_variants.txt 2>&1; done
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/r04b_kmppi_bench.txt 2>&1
timeout 300 python tools/edge_overhead.py > gpurun_out/r04b_edge_overhead.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b_bench_default.json 2> gpurun_out/r04b_bench_default.err
cat gpurun_out/r04b_onchip_variants.txt gpurun_out/r04b_kmppi_bench.txt gpurun_out/r04b_edge_overhead.txt | grep -v amdgpu.ids
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04b_bench_default.json'))
print('default ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), 'onchip K1 us', (d.get('onchip') or {}).get('avg_launch_us'),
      'roofline', {k: d['roofline'][k] for k in ('frac', 'avg_launch_us', 'frac_hbm_cold') if k in d['roofline']})
print('family', d.get('controller_family_on_c3_shape')); print('rng modes', {k: round(v['ms_per_step'], 4) for k, v in d.get('other_rng_modes', {}).items()})
print('others', {k: (round(v['ms_per_step'], 4), round(v['k1_avg_us'], 1)) for k, v in d.get('other_workloads', {}).items()})
PY

#!/bin/bash
# round 3, call K: closing run on the head -- the -m gpu suite, smoke(), the two traced-callable examples, C2 / C4 bench lines
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --durations=8 > gpurun_out/r03k_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r03k_pytest.log
tail -14 gpurun_out/r03k_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python examples/learned_dynamics.py --steps 60 --retrain-every 20 > gpurun_out/r03k_example_learned.txt 2>&1; tail -6 gpurun_out/r03k_example_learned.txt
MPPI_AUTO_JIT=sync timeout 300 python examples/plain_callables.py --steps 100 > gpurun_out/r03k_example_plain.txt 2>&1; tail -4 gpurun_out/r03k_example_plain.txt
timeout 300 python bench.py --workload c2 --steps 300 --warmup 30 --no-extras --no-cpu-baseline > gpurun_out/r03k_bench_c2.json 2> gpurun_out/r03k_bench.err
timeout 300 python bench.py --workload c4 --no-extras --no-cpu-baseline > gpurun_out/r03k_bench_c4.json 2>> gpurun_out/r03k_bench.err
python - <<'PY'
import json
for n in ('bench_c2', 'bench_c4'):
    d = json.load(open('gpurun_out/r03k_%s.json' % n)); r = d['roofline'] or {}
    print(n, 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'K1 us %.2f frac %.3f' % (r.get('avg_launch_us', 0), r.get('frac', 0)))
PY

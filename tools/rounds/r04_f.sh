#!/bin/bash
# round 4: the whole GPU suite after the rng="torch" rows + the default bench line + the torch-rng bench with a kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04f
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04f/tests.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r04f/bench_default.json
python bench.py --rng torch --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r04f/bench_torch.json
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_torch -o torch -- python "$GRAFT_REPO_ROOT/bench.py" --rng torch --no-extras --steps 300 --warmup 30 > /dev/null 2>&1 )
f=$(find /tmp/prof_torch -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" > gpurun_out/r04f/torch_kernel_stats.csv
python tools/micro/noise_torch_test.py 2>/dev/null | tail -2 > gpurun_out/r04f/generator_vs_randn.txt
cat gpurun_out/r04f/tests.txt
python -c "
import json
for n in ('default','torch'):
    d=json.load(open(f'gpurun_out/r04f/bench_{n}.json')); print(n, d['value'], d['ms_per_step'], d.get('roofline'))
"
cat gpurun_out/r04f/torch_kernel_stats.csv

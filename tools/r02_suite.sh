#!/bin/bash
# whole GPU suite + smoke (+ the small-command numbers, which are the most sensitive to code-generation changes)
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/all_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --workload c2 --steps 300 --warmup 30 --no-extras --no-cpu-baseline > gpurun_out/r02_bench_c2.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r02_bench_c2.json')); print('c2', d['ms_per_step'], d['roofline']['avg_launch_us'])"
timeout 120 python tools/small_cmd_breakdown.py 2>&1 | grep C2 | tee gpurun_out/r02_small.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- python /root/repo/bench.py --workload c2 --steps 300 --warmup 30 --no-extras --no-cpu-baseline > /dev/null 2>&1; DB=$(find /tmp/pk -name '*.db' | head -1); python /root/repo/tools/prof_summary.py $DB /root/repo/gpurun_out/r02_trace_c2.txt > /dev/null 2>&1)
head -4 gpurun_out/r02_trace_c2.txt | cut -c1-170

#!/bin/bash
# round 4, call G: evidence for the rng="torch" command on the engine's rows -- its bench line, the same command under rocprofv3
# (kernel table + idle gaps of a command), and the default bench line again (other_rng_modes.torch is the driver-visible number)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
python bench.py --rng torch --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r04g_bench_torch.json
MPPI_TORCH_ROWS=0 python bench.py --rng torch --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r04g_bench_torch_randn_array.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_torch -o torch -- python $REPO/bench.py --rng torch --steps 300 --warmup 30 --no-extras --no-cpu-baseline > $REPO/gpurun_out/r04g_bench_under_rocprof_torch.json 2> $REPO/gpurun_out/r04g_prof_torch.log)
DB=$(find gpurun_out/prof_torch -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/r04g_trace_torch.txt > /dev/null
python tools/timeline_gaps.py $DB noise_fill_torch gpurun_out/r04g_timeline_gaps_torch.txt > /dev/null
rm -rf gpurun_out/prof_torch
python bench.py 2>/dev/null | tail -1 > gpurun_out/r04g_bench_default.json
python - <<'PY'
import json
for n in ("torch", "torch_randn_array", "default"):
    d = json.load(open(f"gpurun_out/r04g_bench_{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("other_rng_modes", {}).get("torch"))
PY
head -30 gpurun_out/r04g_trace_torch.txt; tail -12 gpurun_out/r04g_timeline_gaps_torch.txt

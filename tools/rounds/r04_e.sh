#!/bin/bash
# round 4, rng="torch" on the engine's rows: tests, then the default-rng bench with and without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_torch_stream.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04e/tests.txt
for rows in 1 0; do
  for wl in c3 c2; do
    MPPI_TORCH_ROWS=$rows python bench.py --workload $wl --rng torch --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r04e/bench_${wl}_rows${rows}.json
  done
done
python - <<'PY'
import json
for wl in ("c3","c2"):
    for rows in (1,0):
        try:
            d=json.load(open(f"gpurun_out/r04e/bench_{wl}_rows{rows}.json"))
            print(wl,"rows",rows,d["ms_per_step"],"ms", d.get("kernel_us"))
        except Exception as e: print(wl,rows,"failed",e)
PY
cat gpurun_out/r04e/tests.txt

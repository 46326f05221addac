#!/bin/bash
mkdir -p gpurun_out
for V in "" _v4 _v8 "" _v4 _v8; do
  MPPI_LIB_SUFFIX=$V timeout 300 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('VPU${V:-_v2}', 'ms/step %.4f' % d['ms_per_step'], 'K1 us %.1f' % r['avg_launch_us'])" >> gpurun_out/r02s_vpu.txt
done
cat gpurun_out/r02s_vpu.txt

#!/bin/bash
# round 6: what would a second wave per SIMD buy the on-chip K1?  The product kernel with the weighting phase and the keeping knocked out
# (tools/micro/onchip_parts.hip, MPPI_ONCHIP_EXP bits 1 | 4: generate + roll out), one workgroup per CU as in the product (K = 65536)
# against two co-resident workgroups per CU (bit 16, K = 131072): perfect overlap would make the second take as long as the first.
mkdir -p gpurun_out
P=${1:-r06_e}
O=gpurun_out/${P}_onchip_two_waves.txt
: > $O
for rep in 1 2; do
for v in "e5_pb12 65536" "e5_pb12 131072" "e21_pb12 131072" "e21_pb6 131072" "e5_pb6 65536" "e21_pb12 65536" "e4_pb12 65536"; do
  set -- $v
  MPPI_MICRO_SPILL=1 timeout 60 ./tools/micro/onchip_parts_$1 $2 | sed "s/^/$1 /" >> $O 2>&1
done
done
cat $O
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_group_threads.py tests/test_gpu_devices.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed"
done > gpurun_out/${P}_group_soak.log 2>&1
cat gpurun_out/${P}_group_soak.log

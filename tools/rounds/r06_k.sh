#!/bin/bash
# round 6: the pair kernel's weighting phase -- how many of a wave's last local super-steps are generated again (MPPI_PAIR_DLS) beside
# the fetch of the rest; C3-sized problem, healthy softmax; bits against the one-wave kernel each time
mkdir -p gpurun_out
P=${1:-r06_k}
O=gpurun_out/${P}_onchip_pair_check.txt
: > $O
for d in "" 3 5 6 7 8 9 10 12 15; do
  echo "== MPPI_PAIR_DLS=$d" >> $O
  MPPI_PAIR_DLS=$d timeout 120 tools/micro/onchip_pair_check_prod 65536 20000 0 64 0.001 >> $O 2>&1
done
for a in "65536 20000 1 48 0.001" "60000 40 1 64 0.05" "50000 20000 1 33 0.001" "49152 20000 0 100 0.001" "3000 20000 0 70 0.001"; do
  echo "== default, $a" >> $O
  timeout 120 tools/micro/onchip_pair_check_prod $a >> $O 2>&1
done
grep -E "^==|identical|MISMATCH|two waves|one wave|failed" $O

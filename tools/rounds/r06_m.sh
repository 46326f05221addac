#!/bin/bash
# round 6: chunk sizes of the pair kernel (rows generated together = super-steps per hand-over x 3): 9 / 12 / 15
mkdir -p gpurun_out
P=${1:-r06_m}
O=gpurun_out/${P}_onchip_pair_check.txt
: > $O
for rep in 1 2; do
for b in tools/micro/onchip_pair_check_*; do
  [ -x $b ] || continue
  echo "== $b" >> $O
  timeout 120 $b 65536 20000 0 64 0.001 >> $O 2>&1
done
done
grep -E "^==|identical|MISMATCH|two waves|one wave|failed" $O

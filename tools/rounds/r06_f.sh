#!/bin/bash
# round 6: the two-waves-per-sample on-chip K1 against the one-wave kernel (tools/micro/onchip_pair_check.hip): bits and time
mkdir -p gpurun_out
P=${1:-r06_f}
O=gpurun_out/${P}_onchip_pair_check.txt
: > $O
for b in tools/micro/onchip_pair_check_*; do
  [ -x $b ] || continue
  echo "== $b" >> $O
  timeout 120 $b 65536 40 >> $O 2>&1
  timeout 120 $b 65536 4000 1 >> $O 2>&1
  timeout 120 $b 50000 0.05 1 48 >> $O 2>&1
done
cat $O

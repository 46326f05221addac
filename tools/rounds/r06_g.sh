#!/bin/bash
mkdir -p gpurun_out
P=${1:-r06_g}
O=gpurun_out/${P}_onchip_pair_check.txt
: > $O
for b in tools/micro/onchip_pair_check_*; do
  [ -x $b ] || continue
  echo "== $b" >> $O
  timeout 120 $b 65536 20000 0 64 0.001 >> $O 2>&1
  timeout 120 $b 65536 25 0 64 0.002 nobounds >> $O 2>&1
done
cat $O

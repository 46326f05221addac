#!/bin/bash
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "kmppi or KMPPI or smppi or SMPPI" 2>&1 | tail -3
timeout 300 python tools/kmppi_bench.py philox 2>&1 | grep KMPPI | tee gpurun_out/kmppi_bench5.txt
timeout 300 python tools/kmppi_bench.py torch 2>&1 | grep KMPPI | tee -a gpurun_out/kmppi_bench5.txt

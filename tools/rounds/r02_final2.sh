#!/bin/bash
# late-round check: whole GPU suite, default bench line, family timings (after the KMPPI / SMPPI changes)
mkdir -p gpurun_out && cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/all_tests.txt
cat gpurun_out/all_tests.txt
timeout 600 python bench.py > gpurun_out/r02b_bench_default.json 2> gpurun_out/r02b_bench_default.err
head -c 1500 gpurun_out/r02b_bench_default.json; echo
timeout 300 python tools/variants_bench.py philox > gpurun_out/r02b_variants.txt 2>&1
timeout 300 python tools/variants_bench.py torch >> gpurun_out/r02b_variants.txt 2>&1
grep "ms/command" gpurun_out/r02b_variants.txt

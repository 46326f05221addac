#!/bin/bash
# round-2 GPU call B: persistent K1 A/B across K, rocprofv3 HBM-cold traces, gpu suite, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/micro/mfma_bf16_layout > gpurun_out/r02b_mfma_layout.txt 2>&1
for K in 65536 131072 262144 524288; do
  for P in 0 1; do
    K=$K MPPI_K1_PERSIST=$P timeout 300 python tools/k1_sweep.py 2>&1 | grep "^\[" | sed "s/^\[/[PERSIST=$P /" >> gpurun_out/r02b_k1_sweep.txt
  done
done
REPO=$PWD
for K in 65536 262144; do
  (cd /tmp && K=$K timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_k1cold_$K -o k1cold -- python $REPO/tools/k1_sweep.py > $REPO/gpurun_out/r02b_prof_k1cold_$K.log 2>&1)
  DB=$(find gpurun_out/prof_k1cold_$K -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/prof_summary.py $DB gpurun_out/r02b_k1_hbm_cold_trace_K$K.txt > /dev/null
done
timeout 900 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/r02b_pytest.log 2>&1
echo "suite rc=$?" >> gpurun_out/r02b_pytest.log
timeout 600 python bench.py > gpurun_out/r02b_bench_default.json 2> gpurun_out/r02b_bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_bench -o bench -- python $REPO/bench.py --no-extras --no-cpu-baseline > $REPO/gpurun_out/r02b_prof_bench.log 2>&1)
DB=$(find gpurun_out/prof_bench -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB gpurun_out/r02b_c3_kernel_trace.txt > /dev/null
cat gpurun_out/r02b_mfma_layout.txt; tail -4 gpurun_out/r02b_pytest.log; cat gpurun_out/r02b_k1_sweep.txt | grep K1; head -8 gpurun_out/r02b_k1_hbm_cold_trace_K65536.txt; head -8 gpurun_out/r02b_k1_hbm_cold_trace_K262144.txt; head -12 gpurun_out/r02b_c3_kernel_trace.txt

run() { python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
run base
MPPI_LIB_SUFFIX=_notanh MPPI_EXTRA_HIPCC_FLAGS="-DMPPI_MLP_NOTANH" run notanh
MPPI_LIB_SUFFIX=_nosb MPPI_EXTRA_HIPCC_FLAGS="-DMPPI_MLP_NOSB" run nosb
MPPI_LIB_SUFFIX=_nt4 MPPI_EXTRA_HIPCC_FLAGS="-DMPPI_MLP_NT=4" run nt4

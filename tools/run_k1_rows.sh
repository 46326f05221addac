b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'K1 us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],3))"; }
b R24
for R in 16 12 9 6; do MPPI_LIB_SUFFIX=_r$R MPPI_EXTRA_HIPCC_FLAGS="-DMPPI_K1_ROWS=$R" b R$R; done
for R in 9 6; do echo sweep R$R; MPPI_LIB_SUFFIX=_r$R MPPI_EXTRA_HIPCC_FLAGS="-DMPPI_K1_ROWS=$R" python tools/k1_sweep.py 2>&1 | grep K1; done

#!/bin/bash
# round 6, evidence on the head: the -m gpu suite, the default bench (driver's protocol), the same under rocprofv3 with the clock calibration
# on the same launches, the counter passes behind bench.py's lookups (stamped with the kernel source hashes of THIS tree: the evidence
# guard), rng="torch" / C2 / C4 lines, the M = 3 and KMPPI traces, family tables, the device group's host share, bench --gpus 2
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
P=${1:-r06_final}
python - <<PY
import json
from pytorch_mppi_amd import _build
json.dump({k: _build.kernel_sources_hash(k) for k in _build.KERNEL_UNITS}, open("gpurun_out/${P}_kernel_source_hashes.json", "w"), indent=1)
PY
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -4 gpurun_out/${P}_pytest.log | cut -c1-300
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${P}_bench_default.json
run_prof() {
  name=$1; shift
  (cd /tmp && MPPI_BENCH_DUMP_LAUNCHES=$REPO/gpurun_out/${P}_launches_$name.json timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python $REPO/bench.py "$@" --no-cpu-baseline > $REPO/gpurun_out/${P}_bench_under_rocprof_$name.json 2> $REPO/gpurun_out/${P}_prof_$name.log)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/clock_calibration.py $DB gpurun_out/${P}_launches_$name.json gpurun_out/${P}_clock_calibration_$name.txt
  python tools/timeline_gaps.py $DB $TLK gpurun_out/${P}_timeline_gaps_$name.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name
}
TLK=rollout_onchip run_prof c3 --steps 20 --warmup 5 --no-extras --hbm-cold
TLK=rollout_cost_kernel run_prof torch --rng torch --steps 300 --warmup 30 --no-extras
timeout 600 python bench.py --rng torch --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/${P}_bench_torch.json
# the one-wave on-chip K1 beside the default (two waves per sample group, csrc/rollout_onchip_pair.hpp): same command, same box
MPPI_ONCHIP_PAIR=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${P}_bench_one_wave_onchip.json
TLK=rollout_onchip MPPI_ONCHIP_PAIR=0 run_prof c3_one_wave --steps 20 --warmup 5 --no-extras
if [ -x tools/micro/onchip_pair_check_prod ]; then
  for a in "65536 20000 0 64 0.001" "65536 20000 1 48 0.001" "60000 40 1 64 0.05"; do timeout 120 tools/micro/onchip_pair_check_prod $a; done > gpurun_out/${P}_onchip_pair_check.txt 2>&1
fi
timeout 600 python bench.py --workload c4 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c4.json
timeout 600 python bench.py --workload c2 --no-extras 2>/dev/null | tail -1 > gpurun_out/${P}_bench_c2.json
run_pmc() {
  name=$1; ctr=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d $REPO/gpurun_out/pmc_$name -o $name -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $REPO/gpurun_out/${P}_pmc_$name.log 2>&1)
  DB=$(find gpurun_out/pmc_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_$name.txt > /dev/null
  rm -rf gpurun_out/pmc_$name
}
run_pmc c3_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run_pmc c3_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run_pmc c3_fetch "FETCH_SIZE"
run_pmc c3_write "WRITE_SIZE"
run_pmc stream_fetch "FETCH_SIZE" --rng philox-stream
run_pmc stream_write "WRITE_SIZE" --rng philox-stream
run_pmc torch_fetch "FETCH_SIZE" --rng torch
run_pmc torch_write "WRITE_SIZE" --rng torch
run_pmc c4_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c4
# the matrix-pipe share of the split-operand kernel on the further shapes (within 15 % of C4's)
for shp in "12 6 128" "16 8 256" "8 2 64" "12 6 256"; do
  tag=$(echo $shp | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc_mlp_$tag -o mlp -- python $REPO/tools/mlp_shape_run.py $shp > $REPO/gpurun_out/${P}_pmc_mlp_$tag.log 2>&1)
  DB=$(find gpurun_out/pmc_mlp_$tag -name "*.db" | head -1); [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_mlp_$tag.txt > /dev/null; rm -rf gpurun_out/pmc_mlp_$tag
done
python - <<PY
import re
def tab(path, kernel):
    out = {}
    try:
        for line in open(path):
            m = re.match(r"\\s*(\\S+)\\s+(\\d+)\\s+([\\d.]+)\\s+([\\d.]+)\\s+([\\d.]+)\\s+(.*)", line)
            if m and kernel in m.group(6):
                out[m.group(1)] = float(m.group(3))
    except Exception:
        pass
    return out
lines = ["# matrix-pipe busy share of rollout_mlp_split_kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024), K = 65536, T = 64 (rocprofv3 --pmc, means over the launches)"]
for name, f in (("c4 (16,4,256)", "gpurun_out/${P}_pmc_c4_mfma.txt"), ("(12,6,256)", "gpurun_out/${P}_pmc_mlp_12_6_256.txt"), ("(12,6,128)", "gpurun_out/${P}_pmc_mlp_12_6_128.txt"), ("(16,8,256)", "gpurun_out/${P}_pmc_mlp_16_8_256.txt"), ("(8,2,64)", "gpurun_out/${P}_pmc_mlp_8_2_64.txt")):
    t = tab(f, "rollout_mlp_split_kernel")
    if t.get("GRBM_GUI_ACTIVE"):
        lines.append(f"  {name:<16} busy {t['SQ_VALU_MFMA_BUSY_CYCLES'] / (t['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f}   (MFMA busy cycles {t['SQ_VALU_MFMA_BUSY_CYCLES']:.0f}, GRBM_GUI_ACTIVE {t['GRBM_GUI_ACTIVE']:.0f})")
    else:
        lines.append(f"  {name:<16} no counters")
open("gpurun_out/${P}_mlp_shapes_mfma_busy.txt", "w").write("\\n".join(lines) + "\\n")
print("\\n".join(lines))
PY
trace() {
  name=$1; pat=$2; shift; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$name -o $name -- python "$@" > $REPO/gpurun_out/${P}_run_$name.log 2>&1)
  DB=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB gpurun_out/${P}_trace_$name.txt > /dev/null
  python tools/timeline_gaps.py $DB $pat gpurun_out/${P}_gaps_$name.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name
}
VARIANTS_N=100 trace m3 rollout_copies_kernel $REPO/tools/variants_bench.py philox "M=3"
VARIANTS_N=200 trace kmppi rollout_kmppi $REPO/tools/variants_bench.py philox "KMPPI"
(cd /tmp && MPPI_MULTI_COPIES=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $REPO/gpurun_out/pmc_m3 -o m3 -- python $REPO/tools/variants_bench.py philox "M=3" > $REPO/gpurun_out/${P}_pmc_m3.log 2>&1)
DB=$(find gpurun_out/pmc_m3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/pmc_summary.py $DB gpurun_out/${P}_pmc_m3.txt > /dev/null; rm -rf gpurun_out/pmc_m3
timeout 300 python tools/kmppi_bench.py philox > gpurun_out/${P}_kmppi_bench.txt 2>&1
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
timeout 300 python tools/variants_bench.py torch > gpurun_out/${P}_variants_torch.txt 2>&1
MPPI_MULTI_COPIES=0 timeout 300 python tools/variants_bench.py philox "M=3" > gpurun_out/${P}_variants_philox_m3_one_lane_form.txt 2>&1
timeout 300 python tools/learned_bench.py > gpurun_out/${P}_learned_bench.txt 2>&1
timeout 900 python tools/group_host_issue.py gpurun_out/${P}_group_host_issue.txt > /dev/null 2>&1
for pm in auto devices spawn; do
  timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --process-model $pm 2>/dev/null | tail -1 > gpurun_out/${P}_bench_gpus2_$pm.json
done
python - <<PY
import json
P="gpurun_out/$P"
for n in ("default","one_wave_onchip","torch","c4","c2","gpus2_auto","gpus2_devices","gpus2_spawn"):
    try:
        d=json.load(open(f"{P}_bench_{n}.json")); print(n, d["ms_per_step"], d["value"], d.get("lookup_stale"))
    except Exception as e:
        print(n, "FAILED", e)
d=json.load(open(f"{P}_bench_default.json"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","bound","frac","achieved","bytes","avg_launch_us","frac_k1_bytes")})
print("streaming", d["streaming"]["ms_per_step"], {k: d["streaming"]["roofline"].get(k) for k in ("frac","frac_hbm_cold","avg_launch_us","median_launch_us_hbm_cold")})
print("synced", d.get("value_synced"), d.get("latency_ms_synced",{}).get("median_ms"))
print("other", d.get("other_rng_modes"))
print("fam", d.get("controller_family_on_c3_shape"))
for k, v in d.get("mlp_shapes", {}).items():
    print("mlp", k, v if isinstance(v, str) else {f: (round(r.get("ms_per_step", 0), 4), round(r.get("k1_algorithmic_tflops", 0), 1), r.get("kernel", r.get("error", ""))[:40]) for f, r in v.items()})
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "live_reference_over_port")})
PY
cat gpurun_out/${P}_group_host_issue.txt
grep -h "rollout_onchip\|weights_partial_rows\|rollout_mlp_split\|rollout_copies\|rollout_cost_kernel" gpurun_out/${P}_pmc_*.txt | cut -c1-170
grep -v amdgpu.ids gpurun_out/${P}_variants_philox.txt gpurun_out/${P}_variants_torch.txt gpurun_out/${P}_variants_philox_m3_one_lane_form.txt gpurun_out/${P}_kmppi_bench.txt
head -8 gpurun_out/${P}_trace_m3.txt | cut -c1-200

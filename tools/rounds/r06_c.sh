#!/bin/bash
# round 6, third GPU session: the whole -m gpu suite on the split modules / templated MLP kernel / KMPPI shift-ahead / device-group workers,
# the default bench line (with mlp_shapes), family timings with the host's share per command, bench --gpus 2 in every process model
mkdir -p gpurun_out
export TMPDIR=/tmp
P=r06_c
# the seed distributions first, with a hard limit (round 6's second session lost them to gpurun's 3600 s cap: the full-size oracle now runs on the GPU)
timeout 1700 python tools/margin_distributions.py ${SEEDS:-32} gpurun_out/r06_margin_distributions > gpurun_out/${P}_margin_distributions.log 2>&1
tail -60 gpurun_out/${P}_margin_distributions.log | cut -c1-220
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/${P}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${P}_pytest.log
tail -6 gpurun_out/${P}_pytest.log | cut -c1-300
cp gpurun_out/parity_margins.json gpurun_out/${P}_parity_margins.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${P}_bench_default.err | tail -1 > gpurun_out/${P}_bench_default.json
timeout 300 python tools/variants_bench.py philox > gpurun_out/${P}_variants_philox.txt 2>&1
timeout 300 python tools/variants_bench.py torch > gpurun_out/${P}_variants_torch.txt 2>&1
grep -v amdgpu.ids gpurun_out/${P}_variants_philox.txt gpurun_out/${P}_variants_torch.txt
timeout 900 python tools/group_host_issue.py gpurun_out/${P}_group_host_issue.txt 2>&1 | tail -14
for pm in auto devices spawn; do
  timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --process-model $pm 2>gpurun_out/${P}_bench_gpus2_$pm.err | tail -1 > gpurun_out/${P}_bench_gpus2_$pm.json
done
python - <<PY
import json
P="gpurun_out/$P"
d=json.load(open(f"{P}_bench_default.json"))
print("bench", d["ms_per_step"], d["value"], "lookup_stale", d.get("lookup_stale"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","bound","frac","achieved","bytes","avg_launch_us","frac_k1_bytes")})
print("other", d.get("other_rng_modes"))
print("fam", d.get("controller_family_on_c3_shape"))
for k, v in d.get("mlp_shapes", {}).items():
    print("mlp", k, v if isinstance(v, str) else {f: (round(r.get("ms_per_step", 0), 4), round(r.get("k1_algorithmic_tflops", 0), 1), r.get("kernel", r.get("error", ""))[:40]) for f, r in v.items()})
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "live_reference_over_port", "estimated_live_reference_value")})
for pm in ("auto", "devices", "spawn"):
    try:
        g=json.load(open(f"{P}_bench_gpus2_{pm}.json")); print("gpus2", pm, g["ms_per_step"], g["value"], g["config"].get("process_model", "")[:90], g["config"].get("process_model_choice"))
    except Exception as e:
        print("gpus2", pm, "FAILED", e)
PY

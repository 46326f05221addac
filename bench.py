#!/usr/bin/env python
"""bench.py -- rollouts/sec per `.command()` call (BASELINE.json metric) on N MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2|c4] [--rng philox|philox7|philox-stream|philox-fused|torch-native|torch]

A "step" is one full `MPPI.command(state)`: on-device noise draw, fused rollout+cost (K1),
exp-weighting + weighted update (K3/K4) and, for N>1, the single record all-gather + combine.
Default noise mode: rng="philox" -- the engine's own Philox4x32-10 + Box-Muller generator on the
device.  Since round 3 a command of C3's kind runs ON CHIP (csrc/rollout_onchip.hpp): one launch generates
every sample's normals, rolls out, keeps the bounded noise in accumulation registers / LDS and leaves a
partial record per workgroup, a second one combines them -- no (K,T,nu) array is written or read.
"philox-stream" is the form with the rows in memory (generator launch -> K1 streams them -> K3 re-reads
them -> K4), "philox-fused" lets K1 generate and store them; "torch-native" / "torch" draw with torch.randn.
Default workload = BASELINE.json configs[2] ("c3": 12-DoF quadratic toy dynamics, K=65536, T=64,
nx=16, nu=12, fp32) -- the configuration the north_star's roofline target is quoted on; K is
per GPU (weak scaling: the sample axis is sharded, K_global = N*65536).
Prints ONE JSON line (rank 0).  `roofline` is for K1 = rollout_cost_kernel (HBM-bound: it streams
the K*T*nu standard normals once): algorithmic bytes 4*K*T*nu + 4*K per launch
divided by its average DISPATCH duration over the K1 launches of the timed region.  The clock: every K1
launch stamps the device wall clock at workgroup entry and exit (C-ABI hook mppi_profile_enable /
mppi_profile_read_launches; min entry .. max exit = the kernel's own span, no extra packets in the timed
region), plus DISPATCH_OFFSET_US, the constant part of a dispatch that span cannot see (ramp in front
of the first wave, drain behind the last) -- calibrated against `rocprofv3 --kernel-trace` on the SAME
launches (profiles/r03_k1_clock_calibration_<workload>.txt), so that `avg_launch_us` is the figure rocprofv3
reports; `frac_device_span` is the span alone.  HIP events attached to a launch are NOT a neutral clock
on this stack (an event-carrying dispatch reads ~3 us longer on every clock, rocprofv3 included:
profiles/r03_event_clock.txt): they are sampled in a short pass of their own behind the timed region and
reported as `avg_launch_us_hip_events`.
`latency_ms_synced` is the reference's own protocol (tests/benchmark_mppi.py:84-113: reset, sync, one
command, sync; 3 warm-ups, 20 iterations, 10 % trimmed mean) beside the pipelined `ms_per_step`.
N > 1: under torch.distributed.run (what the driver does) every rank is one process on one GPU, `shard=(rank, N)`, RCCL.
`python bench.py --gpus N` outside a launcher picks the process model that scales on this box (`--process-model auto`,
choose_process_model): ONE process on N devices -- `MPPI(..., devices=[0..N-1])`, SURVEY 8b / 8e's process model, each device's
launches issued by a worker thread of the engine -- when the measured host-only issue time of a sharded command fits under the
single-GPU command and the group validates, else N self-started ranks (torch.distributed.run on 127.0.0.1); the line says which
ran and why (`config.process_model`, `config.process_model_choice`, `host_issue_us_per_device`).  On a box with fewer than N GPUs the shards / ranks share the devices and the record
exchange is staged (device copies / gloo): a test rig for the code path, labelled as such.
`cpu_baseline` times the oracle (CPU restatement of the reference path, kind "port") on the same
workload at the full K on the host cores (at most 3 timed calls); the live reference itself, timed
in the build container beside the port, is on record in profiles/r06_cpu_reference_vs_port.txt and its
ratio to the port travels in the line (`cpu_baseline.live_reference_over_port`, a lookup).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

WORKLOADS = {
    # name: (description, model factory, nx, nu, K per GPU, T, sigma, bounds)
    "c2": ("C2 pendulum K=8192 T=32 nx=2 nu=1 fp32", "pendulum", 2, 1, 8192, 32),
    "c3": ("C3 12-DoF quadratic toy (integrator) K=65536 T=64 nx=16 nu=12 fp32", "integrator", 16, 12, 65536, 64),
    "c4": ("C4 2-layer MLP dynamics nx=16 H=256 nu=4 K=65536 T=64 fp32", "mlp", 16, 4, 65536, 64),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
# What `rocprofv3 --kernel-trace` reports for a plain K1 dispatch on top of the kernel's own device-clock span
# (dispatch ramp in front of the first wave + end-of-kernel drain behind the last), measured on the SAME launches of
# this very command line under rocprofv3 (tools/clock_calibration.py -> profiles/r03_k1_clock_calibration_<workload>.txt:
# c3 (streaming K1) avg 1.82 / 1.55 / 1.83 / 1.45 / 1.40 / 1.47 in six runs, c4 1.39 / 1.40 / 1.42, c2 2.43 / 2.38 us; the on-chip K1
# 1.73 / 1.57 / 1.84 / 1.90; the HBM-cold launches, which do not start behind a draining generator kernel, located in the
# trace by their spans: + 1.56 / 1.59 / 1.57 us, profiles/r03_final_clock_calibration_c3.txt -- the 0.8 us used until then
# came from a round-2 comparison of two different runs).  The constants are the means; every one of those runs stays within
# 1 % of its rocprofv3 figure with them.
DISPATCH_OFFSET_US_BY_WORKLOAD = {"c3": 1.6, "c4": 1.4, "c2": 2.4}
DISPATCH_OFFSET_US_COLD = 1.57
DISPATCH_OFFSET_US_ONCHIP = 1.76     # profiles/r03_final_clock_calibration_c3.txt
K1_KERNEL_PATTERN = {"pendulum": "rollout_cost_kernel", "integrator": "rollout_cost_kernel", "mlp": "rollout_mlp_split_kernel"}
STAMPS_ONLY = 1 << 30      # mppi_profile_enable argument: device-clock stamps on every launch, HIP events on none
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32) = fp32 vector peak


def make_controller(pm, wl, device, rng, shard, K, devices=None):
    _, kind, nx, nu, _, T = WORKLOADS[wl]
    dtype = torch.float32
    torch.manual_seed(0)
    if kind == "pendulum":
        model = pm.models.Pendulum()
        sigma = torch.tensor(10.0, dtype=dtype)
        kw = dict(u_min=torch.tensor(-2.0, dtype=dtype), u_max=torch.tensor(2.0, dtype=dtype), lambda_=1.0)
        x0 = torch.tensor([3.141592653589793, 1.0], dtype=dtype)
    elif kind == "integrator":
        model = pm.models.Integrator(nx, nu)
        sigma = torch.eye(nu, dtype=dtype)
        kw = dict(lambda_=1.0)
        x0 = torch.randn(nx, dtype=dtype)
    else:
        model = pm.models.MLPResidual.random(nx, nu, 256, seed=2, dtype=dtype)
        sigma = torch.eye(nu, dtype=dtype)
        kw = dict(lambda_=1.0)
        x0 = torch.randn(nx, dtype=dtype)
    # small nominal sequence: the lambda-independent term sum(U*eps)/sigma^2 of the cost has std |U|_2,
    # which must stay O(1) for a healthy softmax (N_eff >> 1)
    U0 = torch.randn(T, nu, dtype=dtype) * 0.02
    ctrl = pm.MPPI(model.dynamics, model.running_cost, nx, sigma, num_samples=K, horizon=T, device=device,
                   U_init=U0, rng="philox7" if rng == "philox7" else ("philox" if rng.startswith("philox") else rng), seed=1234, shard=shard, devices=devices, **kw)
    if rng == "philox-fused":
        ctrl.philox_fill = False                  # force the generation into K1
        ctrl.philox_onchip = False
    if rng == "philox-stream":
        ctrl.philox_onchip = False                # rows in memory: generator launch (long horizons) -> K1 -> K3 -> K4
    return ctrl, x0.to(device), model


def _oracle_problem(wl, K):
    from oracle import dynamics as dyn
    from oracle import mppi_oracle as orc
    _, kind, nx, nu, _, T = WORKLOADS[wl]
    dtype = torch.float32
    if kind == "pendulum":
        f, q = dyn.pendulum_dynamics, dyn.pendulum_cost
        return orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=torch.tensor(10.0), K=K, T=T,
                           u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
    if kind == "integrator":
        f, q = dyn.make_quadtoy(nx, nu)
        return orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=torch.eye(nu), K=K, T=T)
    W = dyn.make_mlp_weights(nx, nu, 256, seed=2, dtype=dtype)
    W = tuple(w.to(torch.empty(0).device) for w in W)       # follows the ambient default device
    f, q = dyn.make_mlp(*W)
    return orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=torch.eye(nu), K=K, T=T)


def _time_oracle(wl, K, budget_s, max_calls, sync):
    from oracle import mppi_oracle as orc
    _, kind, nx, nu, Kfull, T = WORKLOADS[wl]
    p = _oracle_problem(wl, K)
    U = torch.randn(T, nu) * 0.3
    x0 = torch.randn(nx)
    times = []
    t_start = time.perf_counter()
    it = 0
    while True:
        sync()
        t0 = time.perf_counter()
        z = torch.randn(K, T, nu, dtype=torch.float32)    # the reference's draw, mppi.py:203
        r = orc.command(p, U, x0, z, True)
        U = r["U"]
        sync()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)                              # first call = warm-up
        it += 1
        if times and (time.perf_counter() - t_start > budget_s or len(times) >= max_calls):
            break
    return sorted(times)[len(times) // 2], len(times)


def cpu_baseline(wl, budget_s=15.0):
    """The oracle (oracle/mppi_oracle.py, a CPU restatement of the reference's command()) timed on
    the same workload at the full K (same T/nx/nu/model, incl. torch.randn; 1 warm-up + <= 3 calls).
    Beside it, for orientation only: the same restatement with its tensors on cuda:0 at the FULL K
    -- i.e. the reference's own formulation (one ATen launch per tensor op, ~10 per time step) on
    this GPU, the comparator SURVEY.md 8(d) asks for next to the CPU number."""
    _, kind, nx, nu, Kfull, T = WORKLOADS[wl]
    K = Kfull
    t, n = _time_oracle(wl, K, budget_s, 3, lambda: None)       # full K: 1 warm-up + at most 3 timed calls
    out = {"value": K / t, "unit": "rollouts/s", "cores": torch.get_num_threads(), "kind": "port",
           "host_cpus": os.cpu_count(),
           "sample": f"oracle command() incl. randn, K={K} of {Kfull}, T={T}, nx={nx}, nu={nu}, fp32, "
                     f"median of {n} calls ({t * 1e3:.1f} ms each)",
           "state_evals_per_s": K * T / t}
    try:
        # how the port stands to the LIVE reference on the same workload (timed side by side in the build container, where
        # /root/reference exists; a lookup -- this box has no reference to time): VERDICT r05 next #7
        ref = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")))
        out["live_reference_over_port"] = ref[wl]["live_reference_over_port"]
        out["live_reference_over_port_source"] = ref["source"] + " (reference time / port time, build container; a lookup)"
        out["estimated_live_reference_value"] = out["value"] / ref[wl]["live_reference_over_port"]
    except Exception:
        out["live_reference_over_port"] = None
    try:
        with torch.device("cuda"):
            tg, ng = _time_oracle(wl, Kfull, 5.0, 30, torch.cuda.synchronize)
        out["same_port_on_gpu"] = {
            "value": Kfull / tg, "unit": "rollouts/s", "ms_per_command": tg * 1e3,
            "sample": f"the same torch-op restatement with tensors on cuda:0 (ATen kernels), full K={Kfull}, "
                      f"median of {ng} calls"}
    except Exception as e:      # a comparator, never a reason to lose the bench line
        out["same_port_on_gpu"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


CLOCK_WARMUP_S = 0.03


def clock_warmup(ctrl, x0, seconds=CLOCK_WARMUP_S):
    """Untimed commands until the GPU has been busy for `seconds`: the chip clocks to its load, and a region timed right
    behind set-up work (allocations, a probe controller, a host copy: milliseconds of idle GPU) starts at idle clocks --
    measured with tools/edge_overhead.py: the same 20 C3 commands take 81.6 us each on a busy chip and 86-94 us behind a
    1-2 ms pause.  The driver's --warmup (5 commands = 0.4 ms of C3 work) is issued first and is far too short for that; what
    this adds is reported as `clock_warmup_commands`.  Returns the number of commands issued."""
    n = 0
    if ctrl._sharded():
        # every rank must issue the SAME number of sharded commands (each carries a collective): a fixed count, not a clock
        for _ in range(120):
            ctrl.command(x0)
        torch.cuda.synchronize()
        return 120
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            ctrl.command(x0)
        n += 10
        torch.cuda.synchronize()
    return n


def _stats(xs):
    xs = sorted(x for x in xs if x is not None and x > 0)
    if not xs:
        return None
    return {"n": len(xs), "avg": sum(xs) / len(xs), "median": xs[len(xs) // 2], "min": xs[0], "max": xs[-1]}


def k1_hbm_cold(ctrl, n=32):
    """K1 alone with its noise rows HBM-cold: the same launch (C-ABI mppi_rollout_cost on the last
    command's problem block) cycled over NBUF row buffers of 4*K*T*nu bytes each, >= 1.5 GiB in total,
    so that no launch finds its rows in the 256 MiB Infinity Cache (inside a command K1 runs right
    after the generator wrote the 201 MB draw and part of its reads are cache hits: `frac` above is
    that in-pipeline figure, this one is the pure-HBM one).  One untimed pass over every buffer first
    (warm-up, not in the statistics); then `n` launches, per-launch device-clock spans, MEDIAN quoted."""
    import ctypes as C
    from pytorch_mppi_amd import _native as N
    from pytorch_mppi_amd.mppi import _ptr
    lib = N.lib()
    p = ctrl._last
    if p is None or int(p.noise_src) != N.NOISE_TNK4 or int(p.noise_coloured):
        return None
    n_el = ctrl._zelems(ctrl.T)
    nbuf = max(4, -(-6 * (1 << 28) // (4 * n_el)))
    bufs = [torch.randn(n_el, device=ctrl.d, dtype=ctrl.dtype) for _ in range(nbuf)]
    st = ctrl._stream()
    z_save = p.z
    try:
        for i in range(nbuf):                                   # warm-up: one pass over every buffer
            p.z = _ptr(bufs[i])
            N.check(lib.mppi_rollout_cost(C.byref(p), st), "mppi_rollout_cost")
        torch.cuda.synchronize()
        lib.mppi_profile_enable(STAMPS_ONLY)                    # device-clock stamps only
        for i in range(n):
            p.z = _ptr(bufs[i % nbuf])
            N.check(lib.mppi_rollout_cost(C.byref(p), st), "mppi_rollout_cost")
        dev, _ = N.profile_read_launches()
        lib.mppi_profile_enable(0)
    finally:
        p.z = z_save
    st_ = _stats(dev)
    return {"launch_us_device_span": st_, "launch_us_device_span_all": list(dev), "warmup_launches_excluded": nbuf,
            "buffers": nbuf, "bytes_cycled": 4 * n_el * nbuf}


LOOKUPS_USED = {}      # lookup name -> stale? (the committed counter passes this line quotes)


def _lookup_stale(entry, kernel, name):
    """Is a committed counter-pass entry still evidence for the kernel this run timed?  Every entry of profiles/pmc_*.json records
    the sha256 of the kernel's translation unit (source, included headers, flags: pytorch_mppi_amd/_build.kernel_sources_hash) as it
    was when the counters were collected; a different hash now -- or none on record -- means the numbers describe another build."""
    from pytorch_mppi_amd import _build
    try:
        stale = entry.get("sources_sha256") != _build.kernel_sources_hash(kernel)
    except Exception:
        stale = True
    LOOKUPS_USED[name] = bool(stale)
    return bool(stale)


def _pmc_lookup(key, kernel):
    """(HBM bytes per launch, stale?) from the committed FETCH_SIZE / WRITE_SIZE passes"""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[key][kernel]
        return e["traffic_bytes"], _lookup_stale(e, kernel, f"pmc_traffic.json:{key}/{kernel}")
    except Exception:
        return None, None


def _c4_lookup():
    """Matrix-pipe busy share and VALU issue share of the C4 kernel from the committed counter pass / ISA budget (lookups)."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "pmc_c4_mfma.json")))
        return {"mfma_busy": c["mfma_busy"], "valu_issue_frac": c["valu_issue_frac"], "c4_lookup_source": c["source"],
                "c4_lookup_stale": _lookup_stale(c, "rollout_mlp_split_kernel", "pmc_c4_mfma.json")}
    except Exception:
        return {"mfma_busy": None, "valu_issue_frac": None}


def _onchip_valu_lookup(workload, k1_us):
    """What bounds the headline kernel: VALU issue -- the generator.  From the committed SQ counter passes of
    this command (profiles/pmc_onchip_valu.json; a LOOKUP, flagged `stale` when the kernel's sources have changed since): VALU
    instructions per wave, the share of the wave's cycles in which the VALU is executing one (a wave alone on its SIMD cannot issue
    while it waits for its own previous result, LDS or the scalar unit), and the time the same instruction stream would take with
    the VALU never idle."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "pmc_onchip_valu.json")))[f"{workload}/philox-onchip"]["rollout_onchip_kernel"]
    except Exception:
        return None
    w = c["waves"]
    wps = c.get("waves_per_simd", 1)                  # the two-wave kernel (round 6): a SIMD's VALU is busy when EITHER wave's is
    frac = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    return {"stale": _lookup_stale(c, "rollout_onchip_kernel", "pmc_onchip_valu.json"),
            "kernel": c.get("kernel", "rollout_onchip_kernel"), "waves_per_simd": wps,
            "valu_active_share_of_wave_cycles": frac, "valu_busy_share_of_simd_cycles": frac * wps,
            "valu_insts_per_wave": c["SQ_INSTS_VALU"] / w,
            "valu_active_cycles_per_wave": 4.0 * c["SQ_ACTIVE_INST_VALU"] / w, "wave_cycles": 4.0 * c["SQ_WAVE_CYCLES"] / w,
            "issue_floor_us": frac * wps * k1_us if k1_us else None,
            "cycles_per_valu_inst": 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
            "waiting_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "traffic_bytes": (2 * c["FETCH_SIZE_KiB"] + c["WRITE_SIZE_KiB"]) * 1024,
            "source": "profiles/pmc_onchip_valu.json (rocprofv3 --pmc passes: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_*; "
                      "FETCH_SIZE doubled per the gfx950 note) -- a lookup, not measured in this run; `issue_floor_us` = the share x this "
                      "run's measured launch time"}


def mlp_shapes(pm, device, lib, N):
    """Dense-MLP dynamics beyond C4's one shape (shape source /root/reference/tests/pendulum_approximate.py:47-67):
    K = 65536, T = 64, x' = x + 0.1 (W2 tanh(W1 [x;u] + b1) + b2), per (nx, nu, hidden) and per kernel form -- the split-operand
    matrix-core kernel (the default where it is instantiated), the exact-fp32 form (MPPI_MLP_EXACT=1: fp32 MFMA for (16,4), per lane
    otherwise) and the per-lane form (MPPI_MLP_VALU=1) -- the command's time, K1's launch time (device-clock stamps + the C4
    dispatch offset), the algorithmic fp32 TFLOP/s of the two dense layers over K1's time, and which kernel ran."""
    K, T = 65536, 64
    out = {}
    for nx, nu, H in ((16, 4, 256), (12, 6, 128), (8, 2, 64), (16, 8, 256), (32, 8, 256)):
        rec = {}
        flops = 2.0 * ((nx + nu) * H + H * nx) * K * T
        native = bool(N.model_supported(N.MODEL_MLP, nx, nu, N.F32, H))
        for form, env in (("split", {}), ("exact", {"MPPI_MLP_EXACT": "1"}), ("per_lane", {"MPPI_MLP_VALU": "1"})):
            if not native and form != "split":
                continue
            for k_, v_ in env.items():
                os.environ[k_] = v_
            try:
                torch.manual_seed(0)
                model = pm.models.MLPResidual.random(nx, nu, H, seed=2)
                c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device=device, lambda_=1.0,
                            U_init=torch.randn(T, nu) * 0.02, rng="philox", seed=1234, auto_jit=False)
                x = torch.randn(nx).to(device)
                c.command(x)
                c.lambda_ = float(c.cost_total.float().std())
                n = 8 if (form == "split" and native) else 3
                for _ in range(2):
                    c.command(x)
                n0 = int(lib.mppi_stat_mlp_split_launches())
                lib.mppi_profile_enable(STAMPS_ONLY)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    c.command(x)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                dv, _ = N.profile_read_launches()
                lib.mppi_profile_enable(0)
                dvs = _stats(dv)
                if not native:
                    kern = "callback path: no native kernel (nx > 16 needs a second output tile and layer-1 k-step: not built)"
                elif int(lib.mppi_stat_mlp_split_launches()) - n0 >= n:
                    kern = "rollout_mlp_split_kernel (bf16x3 / fp16x2 operands on v_mfma_f32_16x16x32)"
                elif form == "exact" and (nx, nu) == (16, 4):
                    kern = "rollout_mlp_mfma_kernel (v_mfma_f32_16x16x4_f32)"
                else:
                    kern = "rollout_cost_kernel<MlpModel> (one lane per sample, fma chains)"
                r = {"ms_per_step": dt * 1e3, "kernel": kern}
                if dvs and native:
                    k1 = dvs["avg"] + DISPATCH_OFFSET_US_BY_WORKLOAD["c4"]
                    r.update(k1_avg_us=k1, k1_algorithmic_tflops=flops / (k1 * 1e-6) / 1e12)
                rec[form] = r
                del c
            except Exception as e:                   # an extra: never a reason to lose the bench line
                rec[form] = {"error": f"{type(e).__name__}: {e}"[:200]}
            finally:
                for k_ in env:
                    os.environ.pop(k_, None)
        out[f"nx{nx}_nu{nu}_H{H}"] = rec
    out["note"] = ("K = 65536, T = 64, fp32, rng = philox (generator launch -> rows -> K1 -> K3 -> K4); k1_algorithmic_tflops = "
                   "2 ((nx + nu) H + H nx) K T / k1_avg_us.  The split kernel lays every shape into C4's 16-state x 32-slot tile: its "
                   "instruction stream -- and matrix-pipe share -- is C4's whatever (nx, nu <= 16, 8) is")
    return out


def latency_synced(ctrl, x0, warmup=3, iters=20):
    """The reference's timing protocol (/root/reference/tests/benchmark_mppi.py:84-113): warm-ups without the
    shift, then per iteration reset() + state.clone() outside the clock, synchronize, ONE command, synchronize.
    Returns milliseconds: the reference's 10 %-trimmed mean, plus median / min / max."""
    for _ in range(warmup):
        ctrl.command(x0, shift_nominal_trajectory=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        ctrl.reset()
        s_ = x0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctrl.command(s_)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    trim = max(1, len(ts) // 10)
    tr = ts[trim:-trim] if len(ts) > 2 * trim else ts
    return {"trimmed_mean_ms": sum(tr) / len(tr) * 1e3, "median_ms": ts[len(ts) // 2] * 1e3, "min_ms": ts[0] * 1e3,
            "max_ms": ts[-1] * 1e3, "iters": iters,
            "protocol": "reference tests/benchmark_mppi.py:84-113 (reset, sync, one command, sync)"}


def _self_spawn(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one process per GPU,
    rendezvous on 127.0.0.1) and hand their exit code back.  Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    if "MPPI_BENCH_BACKEND" not in env and torch.cuda.device_count() < n:
        # fewer GPUs than ranks (the 1-GPU development box): RCCL refuses two ranks on one device, so the
        # ranks share the GPUs round-robin and the record exchange is staged through gloo -- a rig for the
        # N > 1 CODE PATH, not a measurement; the JSON line says so (config.backend)
        env["MPPI_BENCH_BACKEND"] = "gloo"
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def host_issue_probe(pm, wl, devs, rng="philox", bursts=6, burst=20):
    """What the host needs to get ONE sharded command of THIS workload out (VERDICT r05 next #1a): the device group at the workload's
    own size (K per shard, T, model: the form the command takes -- on chip, streaming, ... -- decides what the host has to do),
    timed in short bursts that start from an idle queue, so that the host never waits for the GPU while it issues and the time per
    command is the host's own whatever the GPUs take.  Returns microseconds per command: total, the calling thread's own share
    (problem blocks + hand-over), and the wait for the engine's worker threads (one device's launches when the devices are
    distinct; on a rig whose shards share ONE device the workers' launches serialise on that device's queue and the wait grows
    with N -- `devices_distinct` says which it was)."""
    _, kind, nx, nu, Kper, T = WORKLOADS[wl]
    dev0 = torch.device("cuda", devs[0])
    ctrl, x0, _ = make_controller(pm, wl, dev0, rng, None, Kper * len(devs), devices=devs)
    uniq = sorted(set(devs))

    def sync():
        for d in uniq:
            torch.cuda.synchronize(d)
    for _ in range(30):
        ctrl.command(x0)
    best = (float("inf"), 0.0)
    for _ in range(bursts):
        sync()
        w0 = ctrl.wait_seconds
        t0 = time.perf_counter()
        for _ in range(burst):
            ctrl.command(x0)
        dt = (time.perf_counter() - t0) / burst
        if dt < best[0]:
            best = (dt, (ctrl.wait_seconds - w0) / burst)
    sync()
    return {"host_issue_us": best[0] * 1e6, "caller_us": (best[0] - best[1]) * 1e6, "wait_for_workers_us": best[1] * 1e6,
            "devices_distinct": len(uniq) == len(devs), "issued_by": ctrl.issue, "draw": ctrl.shards[0].last_draw}


def validate_devices(args):
    """`bench.py --validate-devices N` (run by `--process-model auto` in a subprocess with a timeout, so that a hang of a path that
    has never met this hardware cannot take the bench with it): a device group over N REAL devices commands five times; every
    device must hold bit-identical U, finite, and equal to the unsharded controller's to 2e-5.  Prints one JSON line."""
    import pytorch_mppi_amd as pm
    n = args.validate_devices
    devs = list(range(n))
    dev0 = torch.device("cuda", 0)
    torch.cuda.set_device(dev0)
    _, kind, nx, nu, Kper, T = WORKLOADS[args.workload]
    grp, x0, _ = make_controller(pm, args.workload, dev0, "philox", None, 4096 * n, devices=devs)
    one, _, _ = make_controller(pm, args.workload, dev0, "philox", None, 4096 * n)
    grp.lambda_ = one.lambda_ = 5.0
    ok, why = True, ""
    for i in range(5):
        a, b = grp.command(x0), one.command(x0)
        U0 = grp.shards[0].U.cpu()
        if not all(torch.equal(U0, s_.U.cpu()) for s_ in grp.shards[1:]):
            ok, why = False, f"command {i}: the devices hold different U"
        if not bool(torch.isfinite(a).all()) or float((a - b).abs().max()) > 2e-5 * max(1.0, float(b.abs().max())):
            ok, why = False, f"command {i}: the group's action differs from the unsharded controller's"
    print(json.dumps({"validate_devices": n, "ok": ok, "why": why, "exchange": grp.exchange, "issue": grp.issue}), flush=True)


def choose_process_model(args):
    """`--process-model auto`, `--gpus N` outside a launcher: the model that SCALES on this box for this workload.  One process on N
    devices (`MPPI(..., devices=[...])`) keeps the caller's loop a single process, but its host share per command must fit under
    the GPU's: chosen when (a) the measured host-only issue time of one sharded command is below 0.8 x the single-GPU time of one
    command of the workload, and (b) -- on N real devices -- a short validation run of the group passes (subprocess, timeout).
    Otherwise N self-started ranks (one process per GPU).  Returns (model, evidence dict)."""
    import subprocess
    import pytorch_mppi_amd as pm
    n = args.gpus
    have = torch.cuda.device_count()
    devs = list(range(n)) if have >= n else [i % max(1, have) for i in range(n)]
    ev = {"rule": "devices iff host_issue_us (measured: this workload's own group, bursts from an idle queue) < 0.8 * single_gpu_us_per_step "
                  "(and, on N real devices, the group validates); else spawn"}
    dev0 = torch.device("cuda", devs[0])
    torch.cuda.set_device(dev0)
    desc, kind, nx, nu, Kper, T = WORKLOADS[args.workload]
    one, x1, _ = make_controller(pm, args.workload, dev0, args.rng, None, Kper)
    for _ in range(20):
        one.command(x1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        one.command(x1)
    torch.cuda.synchronize()
    ev["single_gpu_us_per_step"] = (time.perf_counter() - t0) / 100 * 1e6
    del one
    try:
        if have >= n:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--validate-devices", str(n), "--workload", args.workload],
                               capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            ev["validation"] = json.loads(line[-1]) if line else {"ok": False, "why": (r.stderr or r.stdout)[-300:]}
            if not ev["validation"].get("ok"):
                return "spawn", ev
        else:
            ev["validation"] = f"skipped: {n} shards share {have} GPU(s) (a rig for the code path)"
        pr = host_issue_probe(pm, args.workload, devs, args.rng)
        ev.update(pr)
        ev["host_issue_us_per_device"] = pr["caller_us"] / n
        # what N DISTINCT devices would need per command: the caller's N blocks + ONE device's launches (each worker issues its own
        # device's in parallel) -- on a rig of one device the measured wait is N devices' launches one behind the other
        ev["host_issue_us_if_devices_were_distinct"] = pr["caller_us"] + pr["wait_for_workers_us"] / (1 if pr["devices_distinct"] else n)
    except Exception as e:                           # (subprocess timeout, a failing group: the per-process model needs neither)
        ev["error"] = f"{type(e).__name__}: {e}"[:300]
        return "spawn", ev
    fits = ev["host_issue_us"] < 0.8 * ev["single_gpu_us_per_step"]
    if not fits and not ev["devices_distinct"] and ev["host_issue_us_if_devices_were_distinct"] < 0.8 * ev["single_gpu_us_per_step"]:
        ev["note"] = ("on this rig (shards share a device) the group is bound by that one device anyway; with distinct devices the host "
                      "share would fit: `spawn` is chosen because the MEASURED figure decides")
    return ("devices" if fits else "spawn"), ev


def main_devices(args, choice=None):
    """`python bench.py --gpus N` outside any launcher: ONE Python process commanding on N devices through
    `MPPI(..., devices=[0..N-1])` (pytorch_mppi_amd/group.py; SURVEY.md 8b / 8e: ncclCommInitAll communicators, one worker thread per
    device inside the engine).  On a box with fewer than N GPUs the shards share the devices and the records are
    staged (a rig for the code path, labelled in config.process_model).  K is per GPU (weak scaling)."""
    import pytorch_mppi_amd as pm
    n = args.gpus
    have = torch.cuda.device_count()
    devs = list(range(n)) if have >= n else [i % max(1, have) for i in range(n)]
    desc, kind, nx, nu, Kper, T = WORKLOADS[args.workload]
    Kglobal = Kper * n
    dev0 = torch.device("cuda", devs[0])
    torch.cuda.set_device(dev0)

    def sync():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    ctrl, x0, _ = make_controller(pm, args.workload, dev0, args.rng, None, Kglobal, devices=devs)
    if kind != "pendulum":
        probe, _, _ = make_controller(pm, args.workload, dev0, args.rng, None, Kper)
        probe.command(x0)
        ctrl.lambda_ = float(probe.cost_total.float().std())
        del probe
    for _ in range(args.warmup):
        ctrl.command(x0)
    for _ in range(120):
        ctrl.command(x0)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctrl.command(x0)
    t_issue = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    identical = bool(all(torch.equal(ctrl.shards[0].U.cpu(), s_.U.cpu()) for s_ in ctrl.shards[1:]))
    one, x1, _ = make_controller(pm, args.workload, dev0, args.rng, None, Kper)
    one.lambda_ = float(ctrl.lambda_)
    for _ in range(args.warmup):
        one.command(x1)
    clock_warmup(one, x1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one.command(x1)
    torch.cuda.synchronize()
    d1 = time.perf_counter() - t1
    if choice is None:
        choice = host_issue_probe(pm, args.workload, devs, args.rng)
        choice.update(host_issue_us_per_device=choice["caller_us"] / n, rule="--process-model devices (forced)")
    out = {
        "metric": "rollouts/sec (K x T state evals) per .command() call",
        "value": Kglobal * args.steps / dt, "unit": "rollouts/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "K_per_gpu": Kper, "K_global": Kglobal, "T": T, "nx": nx, "nu": nu, "rng": args.rng,
                   "draw": ctrl.last_draw or "torch.randn", "lambda": float(ctrl.lambda_), "sharding": f"samples/{n}",
                   "devices": devs, "devices_hold_identical_U": identical,
                   "process_model": "ONE process, MPPI(..., devices=[...]) (pytorch_mppi_amd/group.py): " + ctrl.exchange + "; launches issued by "
                                    + ctrl.issue + ("" if have >= n else f" -- {n} shards share {have} GPU(s): exercises the code path, not a measurement"),
                   "process_model_choice": choice},
        "state_evals_per_s": Kglobal * T * args.steps / dt,
        "host_issue_us_per_command_in_the_timed_region": t_issue / args.steps * 1e6,
        "host_issue_us_per_device": choice.get("host_issue_us_per_device"),
        "roofline": None,
        "weak_scaling": {"single_gpu_ms_per_step": d1 / args.steps * 1e3, "single_gpu_rollouts_per_s": Kper * args.steps / d1,
                         "weak_scaling_speedup": n * d1 / dt, "weak_scaling_efficiency": d1 / dt,
                         "note": "against the same workload unsharded at K_per_gpu on device 0, timed in this run with the same loop.  Each "
                                 "device's launches are issued by a worker thread of the engine (csrc/group.hip): the calling thread's share "
                                 "is the N problem blocks (`host_issue_us_per_device` x N; config.process_model_choice has the split)"},
    }
    print(json.dumps(out), flush=True)


def exchange_time_us(ctrl, dist, barrier, n=30):
    """The collective part of a sharded command alone: all-gather of the (2 + T*nu)-element shard record + K5
    (rank-order combine), issued back to back `n` times on the last command's problem block; max over ranks by
    construction (barrier on both sides).  Path = what command() itself uses."""
    p = ctrl._last
    if p is None:
        return None
    comm = ctrl._shard.native_comm(ctrl.d)
    if comm is not None:
        path = "engine-owned RCCL communicator: ncclAllGather + K5 on the command's stream (mppi_exchange_combine)"
        one = lambda: ctrl._exchange_native(p, comm)
    else:
        path = ("torch.distributed all_gather_into_tensor (" + dist.get_backend() + ") + K5"
                + (", record staged through the host: TEST RIG" if dist.get_backend() == "gloo" else ""))
        one = lambda: ctrl._combine(p, ctrl._shard.all_gather(p._keep["record"]))
    for _ in range(3):
        one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    barrier()
    return {"us_per_exchange": (time.perf_counter() - t0) / n * 1e6, "path": path, "calls": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--rng", default="philox", choices=["torch", "torch-native", "philox", "philox7", "philox-stream", "philox-fused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--hbm-cold", action="store_true", help="with --no-extras: still run the HBM-cold K1 pass (for a rocprofv3 trace of those launches)")
    ap.add_argument("--process-model", default="auto", choices=["auto", "devices", "spawn"],
                    help="--gpus N outside torch.distributed.run: 'devices' = ONE process, MPPI(..., devices=[0..N-1]); 'spawn' = start N ranks "
                         "(one process per GPU, shard=(rank, N)) like the driver's launcher does; 'auto' = whichever scales here: devices "
                         "when the measured host-only issue time of a sharded command fits under the single-GPU command (and the group "
                         "validates on N real devices), else spawn (choose_process_model)")
    ap.add_argument("--validate-devices", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.validate_devices:
        return validate_devices(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            if os.environ.get("MPPI_BENCH_SPAWN_ONLY") == "1":
                _self_spawn(args.gpus, sys.argv[1:])           # does not return
            model, choice = args.process_model, None
            if model == "auto":
                model, choice = choose_process_model(args)
                choice["chosen"] = model
                os.environ["MPPI_BENCH_CHOICE_JSON"] = json.dumps(choice)
            if model == "spawn":
                _self_spawn(args.gpus, sys.argv[1:])           # does not return
            return main_devices(args, choice)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if os.environ.get("MPPI_BENCH_SPAWN_ONLY") == "1":
        # plumbing check without a GPU (tests/test_dist_gloo.py): the self-started ranks rendezvous on gloo, agree on
        # the world size, rank 0 prints one line
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            tot = torch.ones(1)
            dist.all_reduce(tot)
            ok = int(tot) == world
            dist.destroy_process_group()
        else:
            ok = True
        if rank == 0:
            print(json.dumps({"spawn_check": bool(ok), "n_gpus": world}), flush=True)
        return
    dist = None
    # MPPI_BENCH_BACKEND=gloo: test rig for the N>1 code path on a box with fewer GPUs than ranks (ranks share
    # the devices and the record all-gather is staged through the host); real runs use nccl = RCCL over xGMI
    backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pytorch_mppi_amd as pm

    desc, kind, nx, nu, Kper, T = WORKLOADS[args.workload]
    Kglobal = Kper * world                      # weak scaling over the sample axis
    shard = (rank, world) if world > 1 else None
    ctrl, x0, _ = make_controller(pm, args.workload, device, args.rng, shard, Kglobal)

    # healthy softmax: lambda ~ std of the rollout cost, measured by a throw-away
    # probe controller (its first update, made at lambda=1, is an argmin copy and is discarded)
    if kind != "pendulum":
        probe, _, _ = make_controller(pm, args.workload, device, args.rng, shard, Kglobal)
        probe.command(x0)
        lam = probe.cost_total.float().std()
        if world > 1:
            lam = lam.cpu() if backend != "nccl" else lam
            dist.all_reduce(lam, op=dist.ReduceOp.SUM)
            lam = lam / world
        ctrl.lambda_ = float(lam)
        del probe

    # torch-generator modes: every rank draws its own shard, so the ranks need distinct streams
    torch.manual_seed(1234 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from pytorch_mppi_amd import _native as N
    lib = N.lib()
    for _ in range(args.warmup):
        ctrl.command(x0)
    # (the timed region is `steps` commands -- 1.6 ms at the driver's 20: one collector pause inside it is 5-10 % of the figure.
    #  Collected and switched off HERE, in front of the clock warm-up: a collection takes milliseconds in which the GPU idles and
    #  clocks down -- put between the warm-up and the region it cost the on-chip K1 5 us per launch, 75.4 against 70.7)
    import gc
    gc.collect()
    gc.disable()
    n_clock_warmup = clock_warmup(ctrl, x0)
    # device-clock stamps on every K1 launch of the timed region: no extra packets, no events (see the docstring)
    lib.mppi_profile_enable(STAMPS_ONLY)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctrl.command(x0)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    k1_dev_us, _ = N.profile_read_launches()
    lib.mppi_profile_enable(0)
    dump = os.environ.get("MPPI_BENCH_DUMP_LAUNCHES")
    if dump and rank == 0:
        # per-launch device-clock spans of the timed region, for tools/clock_calibration.py (matched against the
        # rocprofv3 trace of this very process)
        onchip_now = ctrl.last_draw == "philox-onchip"
        json.dump({"warmup": args.warmup, "steps": args.steps, "regions": {
            "headline": {"pattern": "rollout_onchip" if onchip_now else K1_KERNEL_PATTERN[kind], "spans_us": k1_dev_us,
                         "launches_before": (1 if kind != "pendulum" else 0) + args.warmup + n_clock_warmup}}}, open(dump, "w"))
    if world > 1:
        tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ranks_identical = None
    exchange = None
    if world > 1:
        # every rank must hold bit-identical U after the rank-order combine (K5)
        mine = ctrl.U.detach().reshape(-1).contiguous()
        if backend != "nccl":
            mine = mine.cpu()
        allU = torch.empty(world * mine.numel(), device=mine.device, dtype=mine.dtype)
        dist.all_gather_into_tensor(allU, mine)
        allU = allU.view(world, -1)
        ranks_identical = bool(all(torch.equal(allU[0], allU[r]) for r in range(world)))
        exchange = exchange_time_us(ctrl, dist, barrier)
    ms_per_step = dt / args.steps * 1e3
    value = Kglobal * args.steps / dt
    n_eff = None
    if ctrl.omega is not None:
        s2 = (ctrl.omega.double() ** 2).sum()
        if world > 1:
            s2 = s2.cpu() if backend != "nccl" else s2
            dist.all_reduce(s2, op=dist.ReduceOp.SUM)      # omega is normalised globally by K5
        n_eff = 1.0 / float(s2)

    # ---- HIP events, the conventional clock, in a pass of their own (rank 0 reports; every rank runs the
    # commands so that sharded controllers stay in step) ----
    lib.mppi_profile_enable(1)
    for _ in range(12):
        ctrl.command(x0)
    _, k1_ev_us = N.profile_read_launches()
    lib.mppi_profile_enable(0)
    barrier()

    # ---- the headline command runs ON CHIP: its K1 reads no (K,T,nu) array, the HBM roofline does not describe it.
    # It gets its own object (`onchip`); `roofline` is measured on the STREAMING form of the same command (rows in
    # memory: what rng="torch", injected noise, a full Sigma, SMPPI / KMPPI and M > 1 run), timed here, right behind the
    # headline region, with the same clock ----
    onchip = None
    roof_ctrl = ctrl
    if ctrl.last_draw == "philox-onchip":
        oc_dev, oc_ev = _stats(k1_dev_us), _stats(k1_ev_us)
        cs, xs, _ = make_controller(pm, args.workload, device, "philox-stream", shard, Kglobal)
        cs.lambda_ = ctrl.lambda_
        for _ in range(args.warmup):
            cs.command(xs)
        n_cw_stream = clock_warmup(cs, xs)
        lib.mppi_profile_enable(STAMPS_ONLY)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            cs.command(xs)
        barrier()
        dts = time.perf_counter() - t1
        k1_dev_us, _ = N.profile_read_launches()
        lib.mppi_profile_enable(1)
        for _ in range(12):
            cs.command(xs)
        _, k1_ev_us = N.profile_read_launches()
        lib.mppi_profile_enable(0)
        barrier()
        if world > 1:
            tt = torch.tensor([dts], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt)
        if dump and rank == 0:
            dd = json.load(open(dump))
            dd["regions"]["streaming"] = {"pattern": K1_KERNEL_PATTERN[kind], "spans_us": k1_dev_us, "launches_before": args.warmup + n_cw_stream}
            json.dump(dd, open(dump, "w"))
        roof_ctrl = cs
        oc_us = (oc_dev["avg"] + DISPATCH_OFFSET_US_ONCHIP) if oc_dev else 0.0
        ext_bytes = 4 * ctrl.K_local * T * nu + 4 * ctrl.K_local
        spill_bytes = 4 * ctrl._spill[1].numel() if getattr(ctrl, "_spill", None) and ctrl._spill[1] is not None else 0
        oc_traffic = _pmc_lookup(f"{args.workload}/philox-onchip", "rollout_onchip_kernel") if world == 1 else (None, None)
        # which on-chip K1 ran: two waves per 64-sample group (csrc/rollout_onchip_pair.hpp, round 6) or one wave per SIMD
        pair = int(lib.mppi_stat_onchip_pair_launches()) > 0
        oc_kernel = "rollout_onchip_pair_kernel" if pair else "rollout_onchip_kernel"
        onchip = {"kernel": f"{oc_kernel} (csrc/{'rollout_onchip_pair.hpp: two waves per 64-sample group' if pair else 'rollout_onchip.hpp'}) + finalize_blocks_kernel",
                  "k1_kernel": oc_kernel,
                  "no_hbm_mode": spill_bytes == 0,
                  "spill_array_bytes": spill_bytes,
                  "avg_launch_us": oc_us, "avg_launch_us_device_span": oc_dev["avg"] if oc_dev else None,
                  "launch_us_device_span": oc_dev, "avg_launch_us_hip_events": oc_ev["avg"] if oc_ev else None,
                  "dispatch_offset_us": DISPATCH_OFFSET_US_ONCHIP,
                  "hbm_bytes_algorithmic": 4 * ctrl.K_local + 4 * (ctrl.K_local // 256 + 1) * (T * nu + 2) + 2 * spill_bytes,
                  "traffic": oc_traffic[0], "traffic_stale": oc_traffic[1],
                  "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; a lookup, "
                                    "not measured in this run -- `traffic_stale`: the kernel's sources changed since)",
                  "external_z_equivalent_GBs": ext_bytes / (oc_us * 1e-6) / 1e9 if oc_us else None,
                  "bound": "VALU: Philox4x32-10 + Box-Muller of the sample's T*nu normals, generated ONCE; the bounded noise waits for "
                           "its sample's weight in accumulation registers / LDS (105 of 192 rows at C3) and, since ABI 20, in the "
                           "spill array (87 rows: stored once, fetched once, 2 x 94 MB against the streaming command's 604 MB) "
                           "instead of being generated a second time (tools/micro/onchip_parts.hip, profiles/r04_onchip_spill.txt)",
                  "valu_lookup": _onchip_valu_lookup(args.workload, oc_us) if world == 1 else None,
                  "streaming_form_ms_per_step": dts / args.steps * 1e3,
                  "speedup_vs_streaming_form": (dts / args.steps) / (dt / args.steps),
                  "note": "SURVEY.md 8d: with the engine's generator inside K1 the kernel is RNG / VALU-bound and reads no (K,T,nu) array: "
                          "the top-level `roofline` prices its LIVE launch time against the algorithmic bytes of the work it replaces"}

    # ---- roofline of K1 ----
    dev_st = _stats(k1_dev_us)
    ev_st = _stats(k1_ev_us)
    k1_us_span = dev_st["avg"] if dev_st else 0.0
    DISPATCH_OFFSET_US = DISPATCH_OFFSET_US_BY_WORKLOAD[args.workload]
    k1_us = k1_us_span + DISPATCH_OFFSET_US if dev_st else 0.0
    Klocal = ctrl.K_local
    alg_bytes = 4 * Klocal * T * nu + 4 * Klocal
    clock_note = ("avg_launch_us = mean over the K1 launches of the timed region of (device wall-clock span: min workgroup "
                  f"entry .. max exit) + {DISPATCH_OFFSET_US} us dispatch offset = the rocprofv3 --kernel-trace figure "
                  f"(calibration on the same launches: profiles/r03_k1_clock_calibration_{args.workload}.txt); *_device_span = the span "
                  "alone; *_hip_events = hipExtLaunchKernelGGL start/stop pairs in a separate pass (reads ~3 us long: "
                  "profiles/r03_event_clock.txt)")
    roofline = None
    if dev_st and kind == "mlp":
        # C4/C5: the rollout is a chain of two dense layers per state evaluation -> fp32 MFMA bound
        flops = 2.0 * ((nx + nu) * 256 + 256 * nx) * Klocal * T
        ach = flops / (k1_us * 1e-6) / 1e12
        exact = os.environ.get("MPPI_MLP_EXACT") == "1"
        # what the matrix pipe actually executes in the split kernel: per (16 samples x timestep) 96 bf16
        # + 24 fp16 MFMAs of 16x16x32 (2*16*16*32 flop each); the exact kernel executes the algorithmic flops
        executed = flops if exact else 120 * 2.0 * 16 * 16 * 32 * (Klocal / 16) * T
        # the pipe the kernel runs on decides the peak: the exact kernel issues fp32 MFMAs (157.3 TFLOP/s dense), the split
        # kernel 16-bit MFMAs (2.5 PFLOP/s dense) -- 3.3x the algorithmic flops, the price of fp32-level results there.
        # `achieved` = flops the matrix pipe EXECUTES per launch / launch time (= its utilisation, what the PMC busy counter
        # measures); the algorithmic fp32 rate and its ratio to the fp32 MFMA peak are reported beside it.
        pipe_peak = MFMA_F32_PEAK_TFLOPS if exact else 2500.0
        pipe_ach = executed / (k1_us * 1e-6) / 1e12
        roofline = {"bound": "mfma", "achieved": pipe_ach, "peak": pipe_peak, "unit": "TFLOP/s",
                    "frac": pipe_ach / pipe_peak, "traffic": None,
                    "algorithmic_tflops": ach, "algorithmic_frac_of_fp32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS,
                    "kernel": "rollout_mlp_mfma_kernel (fp32 MFMA)" if exact else "rollout_mlp_split_kernel (bf16x3 / fp16x2 MFMA)",
                    "avg_launch_us": k1_us, "avg_launch_us_device_span": k1_us_span, "launch_us_device_span": dev_st,
                    "avg_launch_us_hip_events": ev_st["avg"] if ev_st else None, "timing": clock_note,
                    "algorithmic_flops": flops,
                    "executed_flops": executed,
                    "peak_note": ("peak = dense fp32 MFMA (157.3 TFLOP/s), the kernel issues v_mfma_f32_16x16x4_f32" if exact else
                                  "peak = dense bf16/fp16 MFMA (2.5 PFLOP/s): the kernel issues v_mfma_f32_16x16x32_bf16/_f16 on split "
                                  "operands (96 + 24 per 16 samples x timestep); `achieved` = executed 16-bit flops / launch time; "
                                  "`algorithmic_tflops` = the two dense fp32 layers / launch time"),
                    "limiter": "VALU issue (v_exp + v_rcp per hidden activation), not the matrix pipe"}
    elif dev_st:
        ach = alg_bytes / (k1_us * 1e-6) / 1e9
        if roof_ctrl.last_draw == "philox-k1":
            # no-HBM mode: the normals never exist in memory before K1; report the time against the
            # external-z byte count for orientation only
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "rollout_cost_kernel<..., PHILOX>", "avg_launch_us": k1_us,
                        "avg_launch_us_device_span": k1_us_span, "launch_us_device_span": dev_st,
                        "avg_launch_us_hip_events": ev_st["avg"] if ev_st else None, "timing": clock_note,
                        "note": "rng=philox: K1 generates the normals (Philox4x32-10 + Box-Muller) and WRITES them "
                                "once for K3 -- it is VALU/RNG-bound, 'achieved' is those bytes / time"}
        else:
            # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: separate
            # FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per the gfx950 note); null if not collected
            # for this exact workload/mode
            traffic, traffic_stale = _pmc_lookup(f"{args.workload}/{'philox-stream' if args.rng.startswith('philox') else args.rng}",
                                                 "rollout_cost_kernel") if world == 1 else (None, None)
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale,
                        "kernel": "rollout_cost_kernel", "avg_launch_us": k1_us,
                        "avg_launch_us_device_span": k1_us_span, "launch_us_device_span": dev_st,
                        "frac_device_span": alg_bytes / (k1_us_span * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "avg_launch_us_hip_events": ev_st["avg"] if ev_st else None,
                        "timing": clock_note,
                        "algorithmic_bytes": alg_bytes,
                        "frac_of_measured_copy_ceiling": ach / HBM_COPY_CEILING_GBS,
                        "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                          "passes of this command; a lookup, not measured in this run)" if traffic else None,
                        "frac_note": "`frac` is K1 inside the command pipeline, where part of the 201 MB draw the "
                                     "generator just wrote is still in the 256 MiB Infinity Cache; `frac_hbm_cold` is "
                                     "the same launch with rows that are in HBM only (median of the launches behind one "
                                     "untimed pass over every buffer, + the dispatch offset measured for those launches)"}
            # the HBM-cold pass allocates 1.6 GB and takes a while: rank 0 only, after the barrier above, with the
            # other ranks parked at the barrier below
            cold = k1_hbm_cold(roof_ctrl) if (rank == 0 and (not args.no_extras or args.hbm_cold)) else None
            if cold and cold["launch_us_device_span"]:
                us_c = cold["launch_us_device_span"]["median"] + DISPATCH_OFFSET_US_COLD
                if dump and rank == 0:     # the cold launches, for tools/clock_calibration.py (the LAST dispatches of K1 in the trace)
                    dd = json.load(open(dump))
                    dd["regions"]["hbm_cold"] = {"pattern": K1_KERNEL_PATTERN[kind], "spans_us": cold.pop("launch_us_device_span_all"),
                                                 "launches_before": -1}     # -1: the LAST dispatches of the kernel in the trace
                    json.dump(dd, open(dump, "w"))
                cold.pop("launch_us_device_span_all", None)
                cold["dispatch_offset_us"] = DISPATCH_OFFSET_US_COLD
                ach_c = alg_bytes / (us_c * 1e-6) / 1e9
                roofline.update({"frac_hbm_cold": ach_c / HBM_PEAK_GBS, "achieved_hbm_cold": ach_c,
                                 "median_launch_us_hbm_cold": us_c, "hbm_cold_sample": cold})

    out = {
        "metric": "rollouts/sec (K x T state evals) per .command() call",
        "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "clock_warmup_commands": n_clock_warmup,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "K_per_gpu": Kper, "K_global": Kglobal, "T": T, "nx": nx, "nu": nu,
                   "rng": args.rng, "draw": ctrl.last_draw or "torch.randn", "lambda": float(ctrl.lambda_), "n_eff": n_eff,
                   "sharding": f"samples/{world}" if world > 1 else "none",
                   "ranks_hold_identical_U": ranks_identical},
        "state_evals_per_s": value * T,
        "roofline": roofline,
    }
    if onchip is not None:
        # The headline's timed region runs rollout_onchip_kernel (+ finalize_blocks): `roofline` describes THAT kernel (VERDICT r04
        # item 1).  It reads no (K,T,nu) array, so it has no algorithmic HBM bytes to speak of and is bound by VALU issue (the
        # generator): `frac` = the VALU-active share of its wave cycles (committed SQ counter passes, a lookup like `traffic`),
        # and the two HBM-equivalent fractions price its LIVE launch time against the bytes of the work it replaces -- K1's
        # 4*K*T*nu + 4*K (SURVEY 8d's B1) and the streaming command's B_cmd = B1 + B3.  The HBM-bound K1 of the streaming form
        # (rows in memory: what rng="torch", injected noise, a full Sigma, KMPPI and M > 1 run) keeps its own object: `streaming`.
        b1 = 4 * ctrl.K_local * T * nu + 4 * ctrl.K_local
        b_cmd = b1 + 4 * ctrl.K_local * T * nu + 4 * ctrl.K_local + 8 * T * nu
        oc_us = onchip["avg_launch_us"]
        valu = onchip.pop("valu_lookup")
        ach = b_cmd / (oc_us * 1e-6) / 1e9 if oc_us else None
        # `frac` is recomputable from this object alone -- bytes / (avg_launch_us * 1e-6) / 1e9 / peak, every term
        # measured in this run (the launch time) or a formula of the workload (the bytes) -- nothing looked up.  The bytes are SURVEY
        # 8d's B_cmd = B1 + B3, the algorithmic HBM bytes of the rollout + update this ONE launch performs (K1 reads the K*T*nu
        # normals and writes K costs; K3 reads them again with the costs and writes the (T,nu) update): the kernel itself keeps the
        # normals on chip and is VALU-bound, so this is an HBM-EQUIVALENT rate (how fast the replaced streaming work would have to
        # move its bytes to keep up), not traffic; `frac_k1_bytes` prices the same launch against K1's bytes alone.
        head = {"bound": "hbm", "kernel": onchip["k1_kernel"], "unit": "GB/s",
                "achieved": ach, "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS if ach else None,
                "bytes": b_cmd, "bytes_kind": "algorithmic bytes of the work one launch does: B_cmd = B1 + B3 (SURVEY.md 8d) = "
                                              "2 * (4*K*T*nu + 4*K) + 8*T*nu -- HBM-equivalent, the kernel itself streams no (K,T,nu) array",
                "avg_launch_us": oc_us, "avg_launch_us_device_span": onchip["avg_launch_us_device_span"],
                "avg_launch_us_hip_events": onchip["avg_launch_us_hip_events"], "dispatch_offset_us": DISPATCH_OFFSET_US_ONCHIP,
                "recompute": "frac = bytes / (avg_launch_us * 1e-6) / 1e9 / peak",
                "bytes_k1": b1, "frac_k1_bytes": b1 / (oc_us * 1e-6) / 1e9 / HBM_PEAK_GBS if oc_us else None,
                "traffic": onchip["traffic"], "traffic_stale": onchip["traffic_stale"], "traffic_source": onchip["traffic_source"],
                "algorithmic_hbm_bytes_of_the_kernel_itself": onchip["hbm_bytes_algorithmic"],
                "valu_active_lookup": valu,
                "note": "kernel of the timed headline region; avg_launch_us is measured live in this run (device-clock stamps + dispatch "
                        "offset = the rocprofv3 figure, profiles/r05_*clock_calibration_c3.txt).  What bounds the kernel is VALU issue: "
                        "`valu_active_lookup` (committed counter passes, flagged when stale)"}
        if roofline is not None:
            roofline["measured_on"] = ("the streaming form of this command (rng=philox rows in memory: generator launch -> K1 -> K3 -> K4; "
                                       f"{onchip['streaming_form_ms_per_step']:.4f} ms per command here), timed in this run right behind the "
                                       "headline region")
        out["streaming"] = {"ms_per_step": onchip["streaming_form_ms_per_step"], "rollouts_per_s": ctrl.K_local * world / (onchip["streaming_form_ms_per_step"] * 1e-3),
                            "roofline": roofline}
        out["roofline"] = head
        out["onchip"] = onchip
    if world > 1:
        # ---- what north_star asks of N GPUs (VERDICT r03 missing #1): (a) weak scaling against THIS box's own single-GPU
        # number -- the same workload unsharded at K_per_gpu on this rank's GPU, timed here with the same loop -- and
        # (b) BASELINE.json configs[4] "C5": the MLP dynamics at K = 65536 x N sharded over the N GPUs, with ITS
        # single-GPU reference (C4).  SURVEY 8e: the record exchange is latency-bound (~10-25 us), so C3-sized commands
        # (~0.09 ms) cannot scale like C5-sized ones (~0.5 ms); both are on the line.
        def reduce_max(v):
            tt = torch.tensor([v], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt)

        def timed(c_, x_, steps, warmup):
            for _ in range(warmup):
                c_.command(x_)
            clock_warmup(c_, x_)
            barrier()
            t_ = time.perf_counter()
            for _ in range(steps):
                c_.command(x_)
            barrier()
            return reduce_max(time.perf_counter() - t_)

        def healthy_lambda(wl_, shard_, K_):
            pr, xp, _ = make_controller(pm, wl_, device, args.rng, shard_, K_)
            pr.command(xp)
            lam_ = pr.cost_total.float().std()
            if shard_ is not None:
                lam_ = lam_.cpu() if backend != "nccl" else lam_
                dist.all_reduce(lam_, op=dist.ReduceOp.SUM)
                lam_ = lam_ / world
            return float(lam_)

        def scaling_of(wl_, steps, warmup, sharded_dt=None, lam_=None):
            d_, kind_, nx_, nu_, K_, T_ = WORKLOADS[wl_]
            if lam_ is None and kind_ != "pendulum":
                lam_ = healthy_lambda(wl_, shard, K_ * world)
            if sharded_dt is None:
                cs_, xs_, _ = make_controller(pm, wl_, device, args.rng, shard, K_ * world)
                if lam_ is not None:
                    cs_.lambda_ = lam_
                sharded_dt = timed(cs_, xs_, steps, warmup)
                coll = exchange_time_us(cs_, dist, barrier)
                del cs_
            else:
                coll = exchange
            c1_, x1_, _ = make_controller(pm, wl_, device, args.rng, None, K_)       # unsharded, K_per_gpu, on this rank's own GPU
            if lam_ is not None:
                c1_.lambda_ = lam_
            single_dt = timed(c1_, x1_, steps, warmup)
            del c1_
            return {"workload": d_, "K_per_gpu": K_, "K_global": K_ * world, "n_gpus": world, "steps": steps,
                    "sharded_ms_per_step": sharded_dt / steps * 1e3, "rollouts_per_s": K_ * world * steps / sharded_dt,
                    "single_gpu_ms_per_step": single_dt / steps * 1e3, "single_gpu_rollouts_per_s": K_ * steps / single_dt,
                    "weak_scaling_speedup": world * single_dt / sharded_dt, "weak_scaling_efficiency": single_dt / sharded_dt,
                    "collective_us": coll["us_per_exchange"] if coll else None, "collective_path": coll["path"] if coll else None,
                    "note": "speedup = (K_global / sharded time) / (K_per_gpu / single-GPU time of the same workload, timed in this run "
                            "on this box with the same loop; max over ranks); the collective (record all-gather + K5) is timed alone, back to back"}
        out["weak_scaling"] = scaling_of(args.workload, args.steps, args.warmup, sharded_dt=dt, lam_=float(ctrl.lambda_))
        if args.workload != "c4" and not args.no_extras:
            out["c5"] = scaling_of("c4", max(5, args.steps // 4), 2)
            out["c5"]["config"] = "BASELINE.json configs[4]: the MLP dynamics (nx=16, hidden=256), K = 65536 x N sharded over N GPUs, one record all-gather per command"
        out["config"]["backend"] = ("nccl (RCCL over xGMI), one rank per GPU" if backend == "nccl" else
                                    f"{backend}: TEST RIG -- {world} ranks share {torch.cuda.device_count()} GPU(s), record "
                                    "exchange staged through the host; exercises the N > 1 code path, not a measurement")
        out["config"]["world_size"] = dist.get_world_size()
        out["config"]["process_model"] = "one process per GPU (torch.distributed.run ranks, shard=(rank, N)); engine-owned RCCL communicator"
        if os.environ.get("MPPI_BENCH_CHOICE_JSON"):
            out["config"]["process_model_choice"] = json.loads(os.environ["MPPI_BENCH_CHOICE_JSON"])
            out["host_issue_us_per_device"] = out["config"]["process_model_choice"].get("host_issue_us_per_device")
        out["config"]["collective_per_command"] = exchange

    if rank == 0 and world == 1:
        # `value` is the PIPELINED rate (commands issued back to back, one sync at the end: a control loop that does not read the
        # action back between commands); `value_synced` is SURVEY 8d's t_cmd by the reference's own protocol
        # (tests/benchmark_mppi.py:84-113: reset, sync, ONE command, sync) -- the per-command latency a blocking caller sees
        out["latency_ms_synced"] = latency_synced(ctrl, x0)
        out["value_kind"] = "pipelined: K x steps / wall time of `steps` back-to-back commands between two synchronisations"
        out["value_synced"] = Kglobal / (out["latency_ms_synced"]["median_ms"] * 1e-3)
        out["value_synced_kind"] = "K / median latency of one synchronised command (reference protocol, latency_ms_synced.median_ms)"
    if rank == 0 and world == 1 and not args.no_extras:
        # other noise modes of the same workload (short runs), for the record
        extras = {}
        for mode in ("philox", "philox7", "philox-stream", "philox-fused", "torch-native", "torch"):
            if mode == args.rng:
                continue
            c2, x2, _ = make_controller(pm, args.workload, device, mode, None, Kglobal)
            c2.lambda_ = ctrl.lambda_
            for _ in range(3):
                c2.command(x2)
            clock_warmup(c2, x2)
            t1 = time.perf_counter()
            n = max(40, args.steps)        # (5 commands were mostly pipeline edges: philox7 read 0.077 here and 0.071 over 200)
            for _ in range(n):
                c2.command(x2)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            extras[mode] = {"rollouts_per_s": Kglobal * n / d2, "ms_per_step": d2 / n * 1e3}
            del c2
        out["other_rng_modes"] = extras
        # the other single-GPU BASELINE.json configurations, short runs (parity-test cases, not the
        # headline): C2 pendulum 8192x32 and C4 MLP 65536x64 (fp32 MFMA rollout)
        others = {}
        for wl in ("c2", "c4"):
            if wl == args.workload:
                continue
            d_, kind_, nx_, nu_, K_, T_ = WORKLOADS[wl]
            cw, xw, _ = make_controller(pm, wl, device, args.rng, None, K_)
            if kind_ != "pendulum":
                pr, _, _ = make_controller(pm, wl, device, args.rng, None, K_)
                pr.command(xw)
                cw.lambda_ = float(pr.cost_total.float().std())
                del pr
            nw = 100 if wl == "c2" else 8
            for _ in range(3):
                cw.command(xw)
            clock_warmup(cw, xw)
            lib.mppi_profile_enable(STAMPS_ONLY)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nw):
                cw.command(xw)
            torch.cuda.synchronize()
            dw = time.perf_counter() - t1
            dv, _ = N.profile_read_launches()
            lib.mppi_profile_enable(0)
            dvs = _stats(dv)
            k1us = (dvs["avg"] + DISPATCH_OFFSET_US_BY_WORKLOAD[wl]) if dvs else 0.0
            rec = {"workload": d_, "rollouts_per_s": K_ * nw / dw, "ms_per_step": dw / nw * 1e3, "k1_avg_us": k1us,
                   "latency_ms_synced": latency_synced(cw, xw)}
            if kind_ == "mlp" and k1us > 0:
                # the split kernel runs on the 16-bit matrix pipe (96 bf16 + 24 fp16 MFMAs of 16x16x32 per 16 samples x timestep):
                # priced against THAT pipe's dense peak (2.5 PFLOP/s); the algorithmic fp32 rate stands beside it unpriced
                fl = 2.0 * ((nx_ + nu_) * 256 + 256 * nx_) * K_ * T_
                ex = 120 * 2.0 * 16 * 16 * 32 * (K_ / 16) * T_
                rec["k1_kernel"] = "rollout_mlp_split_kernel (bf16x3 / fp16x2 operands on v_mfma_f32_16x16x32)"
                rec["k1_algorithmic_tflops"] = fl / (k1us * 1e-6) / 1e12
                rec["k1_executed_16bit_tflops"] = ex / (k1us * 1e-6) / 1e12
                rec["k1_frac_of_16bit_mfma_peak"] = rec["k1_executed_16bit_tflops"] / 2500.0
                rec.update(_c4_lookup())
            others[wl] = rec
            del cw
        out["other_workloads"] = others
        if args.workload == "c3":
            out["mlp_shapes"] = mlp_shapes(pm, device, lib, N)
        # small / typical problem sizes at the headline's T, nx, nu (launch-bound, and not in the line until now):
        # which form the command takes and what it costs, pipelined and by the reference's synchronised protocol
        if args.workload == "c3":
            forms = {N.FORM_STREAMING: "streaming (generator, K1, K3, K4)", N.FORM_SINGLE_LAUNCH: "single launch", N.FORM_ONCHIP: "on chip"}
            small = {}
            for Ks in (1024, 8192, 32768):
                cs_, xs_, _ = make_controller(pm, "c3", device, args.rng, None, Ks)
                cs_.lambda_ = ctrl.lambda_
                for _ in range(10):
                    cs_.command(xs_)
                clock_warmup(cs_, xs_)
                t1 = time.perf_counter()
                for _ in range(200):
                    cs_.command(xs_)
                torch.cuda.synchronize()
                dws = time.perf_counter() - t1
                small[f"K{Ks}"] = {"ms_per_step": dws / 200 * 1e3, "rollouts_per_s": Ks * 200 / dws, "draw": cs_.last_draw,
                                   "form": forms.get(int(lib.mppi_last_command_form()), "?"),
                                   "latency_ms_synced_median": latency_synced(cs_, xs_)["median_ms"]}
                del cs_
            out["small_sizes_on_c3_shape"] = small
        # the rest of the controller family on the headline shape, same lambda recipe
        if args.workload == "c3":
            fam = {}
            d_, kind_, nx_, nu_, K_, T_ = WORKLOADS["c3"]
            m_ = pm.models.Integrator(nx_, nu_)
            sig_ = torch.eye(nu_, device=device)
            kwf = dict(num_samples=K_, horizon=T_, device=device, rng=args.rng, lambda_=float(ctrl.lambda_))
            cands = {
                "kmppi_S32": lambda: pm.KMPPI(m_.dynamics, m_.running_cost, nx_, sig_, num_support_pts=T_ // 2,
                                              kernel=pm.RBFKernel(sigma=2.0), **kwf),
                "smppi": lambda: pm.SMPPI(m_.dynamics, m_.running_cost, nx_, sig_, action_min=-torch.ones(nu_), action_max=torch.ones(nu_),
                                          w_action_seq_cost=1.0, delta_t=0.1, **kwf)}
            for name, mk in cands.items():
                cf_ = mk()
                for _ in range(5):
                    cf_.command(x0)
                clock_warmup(cf_, x0)
                t1 = time.perf_counter()
                for _ in range(30):
                    cf_.command(x0)
                torch.cuda.synchronize()
                dw = time.perf_counter() - t1
                fam[name] = {"ms_per_step": dw / 30 * 1e3, "rollouts_per_s": K_ * 30 / dw}
                del cf_
            fam["kmppi_interpolation_inside_k1_launches"] = int(lib.mppi_stat_kmppi_fused_rollouts())
            out["controller_family_on_c3_shape"] = fam
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        # every committed counter pass this line quoted, and whether the kernel it describes is still the one that was timed
        out["lookup_stale"] = any(LOOKUPS_USED.values()) if LOOKUPS_USED else False
        out["lookups"] = dict(LOOKUPS_USED)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

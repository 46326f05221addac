"""Build-container helper (needs /root/reference): time the LIVE reference
(`pytorch_mppi.MPPI.command`, device="cpu", loaded by oracle/ref_loader.py) beside the oracle port
(oracle/mppi_oracle.command, what bench.py's `cpu_baseline` leg times on the GPU box, where the
reference does not exist) on the SAME workloads at the FULL K, same host cores, same protocol as the
reference's own tests/benchmark_mppi.py:84-113 (warm-up, reset() before every timed command()).
The output is committed as profiles/r02_cpu_reference_vs_port.txt: it puts on record how
representative the port is of the reference's own CPU path.

    python tools/cpu_reference_vs_port.py [c2 c3 c4] > profiles/r02_cpu_reference_vs_port.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import mppi_oracle as orc  # noqa: E402
from oracle import ref_loader  # noqa: E402


def time_reference(wl, iters):
    mod, _ = ref_loader.load_reference()
    _, kind, nx, nu, K, T = bench.WORKLOADS[wl]
    p = bench._oracle_problem(wl, K)
    kw = {}
    if kind == "pendulum":
        kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
    ctrl = mod.MPPI(p.dynamics, p.running_cost, nx, p.noise_sigma if kind != "pendulum" else torch.tensor(10.0),
                    num_samples=K, horizon=T, device="cpu", lambda_=1.0, **kw)
    x0 = torch.randn(nx)
    ctrl.command(x0, shift_nominal_trajectory=False)            # warm-up (benchmark_mppi.py:86-88)
    ts = []
    for _ in range(iters):
        ctrl.reset()
        t0 = time.perf_counter()
        ctrl.command(x0.clone())
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def time_port(wl, iters):
    _, kind, nx, nu, K, T = bench.WORKLOADS[wl]
    t, n = bench._time_oracle(wl, K, 1e9, iters, lambda: None)
    return t


def main():
    wls = [a for a in sys.argv[1:] if a in bench.WORKLOADS] or ["c2", "c3", "c4"]
    print(f"# host: {os.cpu_count()} logical CPUs, torch threads {torch.get_num_threads()}, torch {torch.__version__}; fp32; "
          f"median of 3 timed command() calls after 1 warm-up, torch.randn included on both sides")
    print(f"# {'workload':58s} {'reference ms':>13s} {'port ms':>10s} {'port/ref':>9s} {'ref rollouts/s':>15s}")
    for wl in wls:
        desc, kind, nx, nu, K, T = bench.WORKLOADS[wl]
        torch.manual_seed(0)
        tr = time_reference(wl, 3)
        torch.manual_seed(0)
        tp = time_port(wl, 3)
        print(f"  {desc:58s} {tr * 1e3:13.1f} {tp * 1e3:10.1f} {tp / tr:9.3f} {K / tr:15.4g}")


if __name__ == "__main__":
    main()

"""Mid-size problems (K below a full chip, long horizon): per-command time and which form ran.
    [MPPI_LIB_SUFFIX=_x MPPI_EXTRA_HIPCC_FLAGS=...] python tools/mid_cmd.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

lib = N.lib()
for (nx, nu, T) in ((16, 12, 64), (6, 4, 64), (2, 1, 200)):
    m = pm.models.Integrator(nx, nu) if nu > 1 else pm.models.Pendulum()
    sig = torch.eye(nu) if nu > 1 else torch.tensor(1.0)
    for K in (1024, 4096, 16384, 32768):
        c = pm.MPPI(m.dynamics, m.running_cost, nx, sig, num_samples=K, horizon=T, device="cuda", lambda_=50.0, rng="philox", seed=1)
        x0 = torch.zeros(nx, device="cuda")
        for _ in range(5):
            c.command(x0)
        torch.cuda.synchronize()
        n0 = lib.mppi_stat_single_launch_commands()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            c.command(x0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"nx {nx:2d} nu {nu:2d} T {T:3d} K {K:6d}: {dt * 1e3:7.4f} ms/command  draw {c.last_draw}  single-launch commands {lib.mppi_stat_single_launch_commands() - n0}/{n}", flush=True)

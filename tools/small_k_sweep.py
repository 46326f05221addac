"""Dev helper (GPU): the small-K regime (launch-bound) at C3's T, nx, nu and at C2's shape: ms per command by rng mode, with the
form the command took (csrc: streaming chain / single launch / on chip) and the number of launches per command."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

FORMS = {N.FORM_NONE: "none", N.FORM_STREAMING: "streaming", N.FORM_SINGLE_LAUNCH: "single-launch", N.FORM_ONCHIP: "on-chip"}
lib = N.lib()
for (T, nx, nu, kind) in ((64, 16, 12, "integrator"), (32, 2, 1, "pendulum"), (20, 8, 4, "integrator")):
    for K in (256, 1024, 4096, 8192, 16384, 32768):
        for rng in ("philox", "torch"):
            m = pm.models.Integrator(nx, nu) if kind == "integrator" else pm.models.Pendulum()
            sigma = torch.eye(nu) if nu > 1 else torch.tensor(1.0)
            c = pm.MPPI(m.dynamics, m.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", lambda_=50.0, rng=rng, seed=3)
            x = torch.randn(nx, device="cuda")
            for _ in range(20):
                c.command(x)
            torch.cuda.synchronize()
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                c.command(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(100):
                c.command(x); torch.cuda.synchronize()
            lat = (time.perf_counter() - t0) / 100
            print(f"T={T:3d} nx={nx:2d} nu={nu:2d} K={K:6d} rng={rng:7s} {dt * 1e6:7.1f} us/command (pipelined)  {lat * 1e6:7.1f} us (command + sync)  "
                  f"draw={c.last_draw}  form={FORMS.get(int(lib.mppi_last_command_form()), '?')}", flush=True)
            del c

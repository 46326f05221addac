"""Dev helper (GPU): throughput of one controller against the number of samples K (C3's T, nx, nu;
rng="philox"), fp32 and fp64 -- where the launch-bound regime ends and how the HBM-bound regime
scales up to a 3.2 GB draw (288 GB of HBM leave room for K ~ 5e7 at this T, nu)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm

T, nx, nu = 64, 16, 12
for dtype in (torch.float32, torch.float64):
    for K in (1024, 4096, 16384, 65536, 262144, 1048576)[: (3 if os.environ.get("SMALL") else 6)]:
        if dtype == torch.float64 and K > 262144:
            continue
        m = pm.models.Integrator(nx, nu)
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu, dtype=dtype), num_samples=K, horizon=T, device="cuda",
                    lambda_=8000.0, U_init=torch.zeros(T, nu, dtype=dtype), rng="philox", seed=3)
        x = torch.randn(nx, device="cuda", dtype=dtype)
        for _ in range(5):
            c.command(x)
        torch.cuda.synchronize()
        n = 200 if K <= 65536 else 30
        t0 = time.perf_counter()
        for _ in range(n):
            c.command(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        gb = 3 * 4 * K * T * nu * (2 if dtype == torch.float64 else 1) / 1e9     # write + 2 reads of the draw
        print(f"{str(dtype)[6:]:8s} K={K:8d}  {dt * 1e3:8.4f} ms/command  {K / dt:10.3e} rollouts/s  "
              f"draw traffic {gb / dt / 1e3:5.2f} TB/s  ({c.last_draw})", flush=True)
        del c

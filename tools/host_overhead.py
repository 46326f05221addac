"""Dev helper (GPU): host-side cost of one command() (tiny problem => GPU time negligible)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch, cProfile, pstats
import pytorch_mppi_amd as pm
for rng in ("torch-native", "philox"):
    m = pm.models.Pendulum()
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0), num_samples=int(os.environ.get("K", 256)), horizon=8, device="cuda",
                u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng=rng)
    x = torch.tensor([3.14, 1.0], device="cuda")
    for _ in range(20): c.command(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 500
    for _ in range(n): c.command(x)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{rng}: host {t_host / n * 1e6:.1f} us/command, wall incl. GPU drain {t_all / n * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): c.command(x)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

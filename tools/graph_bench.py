"""Dev helper (GPU): eager vs HIP-graph-replayed command() wall time (C2: pendulum 8192 x 32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import pytorch_mppi_amd as pm
for K, T in ((8192, 32), (512, 15), (65536, 64)):
    if K == 65536:
        m = pm.models.Integrator(16, 12); nx = 16; sig = torch.eye(12); kw = {}
        x = torch.randn(16, device="cuda")
    else:
        m = pm.models.Pendulum(); nx = 2; sig = torch.tensor(10.0)
        kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
        x = torch.tensor([3.14, 1.0], device="cuda")
    c = pm.MPPI(m.dynamics, m.running_cost, nx, sig, num_samples=K, horizon=T, device="cuda", rng="torch-native", **kw)
    for _ in range(20): c.command(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for _ in range(n): c.command(x)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
    g = c.capture_command(x)
    for _ in range(20): g(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g(x)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    print(f"K={K} T={T}: eager {te * 1e6:.1f} us/command, HIP graph replay {tg * 1e6:.1f} us/command ({te / tg:.2f}x)")

# generic path: the reference's plugin API -- plain Python callables, T x (dynamics + cost) launches
print("generic path (Python callbacks, ~10 ATen launches per timestep):")
for K, T in ((8192, 32), (65536, 64)):
    def dyn(s, a):
        th, thd = s[:, 0:1], s[:, 1:2]
        u = torch.clamp(a[:, 0:1], -2.0, 2.0)
        nthd = torch.clamp(thd + (15.0 * torch.sin(th) + 3.0 * u) * 0.05, -8.0, 8.0)
        return torch.cat((th + nthd * 0.05, nthd), dim=1)

    def cost(s, a):
        th = ((s[:, 0] + torch.pi) % (2 * torch.pi)) - torch.pi
        return th ** 2 + 0.1 * s[:, 1] ** 2
    x = torch.tensor([3.14, 1.0], device="cuda")
    c = pm.MPPI(dyn, cost, 2, torch.tensor(10.0), num_samples=K, horizon=T, device="cuda", rng="torch-native",
                u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
    for _ in range(10): c.command(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): c.command(x)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
    g = c.capture_command(x)
    for _ in range(10): g(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g(x)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    print(f"K={K} T={T}: eager {te * 1e6:.1f} us/command, HIP graph replay {tg * 1e6:.1f} us/command ({te / tg:.2f}x)")

"""Dev helper (GPU): where the host time of a command goes (cProfile over 3000 commands of a tiny problem: the device is never the bound)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, pytorch_mppi_amd as pm
which = sys.argv[1] if len(sys.argv) > 1 else "builtin"
m = pm.models.Pendulum()
if which == "traced":
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traced_pendulum_bench.py")).read().split("for rng in")[0]
    exec(src)
    c = pm.MPPI(dynamics, running_cost, 2, torch.tensor(10.0), num_samples=256, horizon=4, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng="philox", auto_jit=True)
else:
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0), num_samples=256, horizon=4, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng=(which if which in ("torch", "philox") else "philox"))
x = torch.tensor([3.0, 1.0], device="cuda")
for _ in range(100): c.command(x)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): c.command(x)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])

import sys, time, math, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pytorch_mppi_amd as pm
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/traced_pendulum_bench.py")).read().split("for rng in")[0])
from pytorch_mppi_amd import _native as N
FORMS = {N.FORM_NONE: "none", N.FORM_STREAMING: "streaming", N.FORM_SINGLE_LAUNCH: "single-launch", N.FORM_ONCHIP: "on-chip"}
m = pm.models.Pendulum()
for name, f, q, kw in (("built-in", m.dynamics, m.running_cost, {}), ("traced", dynamics, running_cost, dict(auto_jit=True))):
    for K, T in ((256, 4), (8192, 32)):
        c = pm.MPPI(f, q, 2, torch.tensor(10.0), num_samples=K, horizon=T, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng="philox", **kw)
        x = torch.tensor([3.0, 1.0], device="cuda")
        for _ in range(50): c.command(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000): c.command(x)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name:9s} K={K:5d} T={T:3d}: issue {(t1 - t0) / 1000 * 1e6:6.1f} us/command, with the final wait {(t2 - t0) / 1000 * 1e6:6.1f}; form {FORMS[int(N.lib().mppi_last_command_form())]}, draw {c.last_draw}")

import sys, time, math; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, pytorch_mppi_amd as pm
def angle_normalize(x): return (((x + math.pi) % (2 * math.pi)) - math.pi)
def dynamics(state, perturbed_action):
    th = state[:, 0].view(-1, 1); thdot = state[:, 1].view(-1, 1)
    g, m, l, dt = 10, 1, 1, 0.05
    u = torch.clamp(perturbed_action, -2, 2)
    newthdot = thdot + (-3 * g / (2 * l) * torch.sin(th + math.pi) + 3. / (m * l ** 2) * u) * dt
    newth = th + newthdot * dt
    newthdot = torch.clamp(newthdot, -8, 8)
    return torch.cat((newth, newthdot), dim=1)
def running_cost(state, action):
    theta = state[:, 0]; theta_dt = state[:, 1]; action = action[:, 0]
    return angle_normalize(theta) ** 2 + .1 * theta_dt ** 2 + .001 * action ** 2
for rng in ("philox", "torch"):
    c = pm.MPPI(dynamics, running_cost, 2, torch.tensor(10.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng=rng, auto_jit=True)
    x = torch.tensor([3.0, 1.0], device="cuda")
    for _ in range(50): c.command(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): c.command(x)
    torch.cuda.synchronize(); print(rng, "traced pendulum lambdas 8192x32:", (time.perf_counter() - t0) / 500 * 1e6, "us/command", c.jit_note if hasattr(c, "jit_note") else "")

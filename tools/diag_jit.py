import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, numpy as np
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import jit
from oracle import mppi_oracle as orc
dt_, gx, gy, wT = 0.1, 1.5, -0.5, 3.0
def f(s, a):
    return torch.stack((s[:, 0] + dt_ * a[:, 0] * torch.cos(s[:, 2]), s[:, 1] + dt_ * a[:, 0] * torch.sin(s[:, 2]), s[:, 2] + dt_ * a[:, 1]), dim=1)
def q(s, a):
    return (s[:, 0] - gx) ** 2 + (s[:, 1] - gy) ** 2 + 0.01 * (a ** 2).sum(-1)
def term(states, actions):
    last = states[..., -1, :]
    return wT * ((last[..., 0] - gx) ** 2 + (last[..., 1] - gy) ** 2)
model = jit.compile_model("unicycle", 3, 2, dynamics=f, running_cost=q, terminal_state_cost=term, params=[dt_, gx, gy, wT],
    step="const T c = m_cos(x[2]), s = m_sin(x[2]); x[0] += p[0] * u[0] * c; x[1] += p[0] * u[0] * s; x[2] += p[0] * u[1];",
    cost="const T dx = x[0] - p[1], dy = x[1] - p[2]; return dx * dx + dy * dy + T(0.01) * (u[0] * u[0] + u[1] * u[1]);",
    terminal="const T dx = x[0] - p[1], dy = x[1] - p[2]; return p[3] * (dx * dx + dy * dy);")
K, T = 777, 25
g = torch.Generator().manual_seed(3)
U0 = torch.randn(T, 2, generator=g, dtype=torch.float64) * 0.1
x0 = torch.tensor([0.0, 0.0, 0.3], dtype=torch.float64)
sigma = torch.diag(torch.tensor([0.5, 1.0], dtype=torch.float64))
umax = torch.tensor([1.0, 2.0], dtype=torch.float64)
z = torch.randn(K, T, 2, generator=g, dtype=torch.float64)
kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=0.5, u_max=umax, U_init=U0)
fused = pm.MPPI(model.dynamics, model.running_cost, 3, sigma, terminal_state_cost=model.terminal_state_cost, **kw)
generic = pm.MPPI(f, q, 3, sigma, terminal_state_cost=term, **kw)
p = orc.Problem(dynamics=f, running_cost=q, nx=3, noise_sigma=sigma, K=K, T=T, lambda_=0.5, u_max=umax, terminal_state_cost=term)
r = orc.command(p, U0, x0, z, True)
for name, c in (("fused", fused), ("generic", generic)):
    c.inject_noise(z); a = c.command(x0.cuda())
    print(name, "action", a.cpu().numpy(), "cost err", float((c.cost_total.cpu() - r["cost_total"]).abs().max()), "pa err", float((c.perturbed_action.cpu() - r["perturbed_action"]).abs().max()))
print("oracle action", r["action"].numpy(), "neff", 1 / float((r["omega"] ** 2).sum()))

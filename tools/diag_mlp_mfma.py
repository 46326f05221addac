"""Dev helper (GPU): MFMA MLP rollout vs the per-lane VALU kernel and vs the fp64 oracle."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_util as gu
import pytorch_mppi_amd as pm
from oracle import mppi_oracle as orc, dynamics as dyn

K, T, nx, nu, H = 4096, 16, 16, 4, int(os.environ.get("H", 256))
g = torch.Generator().manual_seed(0)
model = pm.models.MLPResidual.random(nx, nu, H, seed=2)
U0 = torch.randn(T, nu, generator=g) * 0.05
x0 = torch.randn(nx, generator=g)
z = torch.randn(K, T, nu, generator=g)
sig = torch.tensor([[1.0, 0.3, 0, 0], [0.3, 0.8, 0, 0], [0, 0, 0.5, 0.1], [0, 0, 0.1, 1.2]]) if os.environ.get("FULL") else torch.eye(nu)
c = pm.MPPI(model.dynamics, model.running_cost, nx, sig, num_samples=K, horizon=T, device="cuda",
            lambda_=5.0, U_init=U0.clone(), u_min=torch.tensor([-1.5] * nu), u_max=torch.tensor([1.5] * nu),
            sample_null_action=True)
c.inject_noise(z)
a = c.command(x0.cuda())
W = [t.double() for t in (model.W1, model.b1, model.W2, model.b2)]
f, q = dyn.make_mlp(*W)
p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sig.double(), K=K, T=T, lambda_=5.0,
                u_min=torch.tensor([-1.5] * nu).double(), u_max=torch.tensor([1.5] * nu).double(), sample_null_action=True)
r = orc.command(p, U0.double(), x0.double(), z.double(), True)
ct = c.cost_total.cpu().double()
print("path:", "VALU" if os.environ.get("MPPI_MLP_VALU") else "MFMA", "H", H,
      "cost rel err max", float(((ct - r["cost_total"]).abs() / r["cost_total"].abs()).max()),
      "action abs err", float((a.cpu().double() - r["action"]).abs().max()),
      "U abs err", float((c.U.cpu().double() - r["U"]).abs().max()))

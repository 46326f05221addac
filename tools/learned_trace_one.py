import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, jit_fixtures as jf, pytorch_mppi_amd as pm
f, q, net = jf.approx_pendulum_callables(dtype=torch.float32); net.cuda()
x0 = torch.tensor([2.5, -0.8]).cuda()
c = pm.MPPI(f, q, 2, torch.tensor(1.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng="philox", seed=1, auto_jit=True)
for _ in range(300): c.command(x0)
torch.cuda.synchronize()

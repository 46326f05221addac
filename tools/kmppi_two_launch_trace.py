import sys, os, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_mppi_amd as pm
nx, nu, T, S, K = 16, 12, 64, 32, 65536
m = pm.models.Integrator(nx, nu)
torch.manual_seed(0)
x0 = torch.randn(nx, device="cuda")
c = pm.KMPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_support_pts=S, kernel=pm.RBFKernel(sigma=2.0), rng="philox", num_samples=K, horizon=T, device="cuda", lambda_=50.0)
c.fuse_interpolation = False
c.onchip_update = False
for _ in range(30):
    c.command(x0)
torch.cuda.synchronize()

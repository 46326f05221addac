"""Dev helper (GPU): time of mppi_noise_fill_philox alone at C3 size (50 M normals, 201 MB written)
next to torch.randn of the same size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

K, T, nx, nu = 65536, 64, 16, 12
m = pm.models.Integrator(nx, nu)
c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", rng="philox", seed=1)
p = c._problem()
c._attach_workspace(p)
rows4 = N.noise_rows4(T, nu)
z = torch.empty(rows4 * K * 4, device="cuda")
p.noise_src, p.call = N.NOISE_PHILOX, 1
lib = N.lib()
st = c._stream()


def timed(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"mppi_noise_fill_philox  {timed(lambda: N.check(lib.mppi_noise_fill_philox(C.byref(p), z.data_ptr(), st), 'fill')):7.1f} us")
print(f"torch.randn             {timed(lambda: torch.randn(rows4 * K * 4, device='cuda')):7.1f} us")
print(f"torch.empty().zero_()   {timed(lambda: z.zero_()):7.1f} us   (write-only floor)")
print("mean/std of the fill:", float(z.mean()), float(z.std()))

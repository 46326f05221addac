"""Dev helper (GPU): time K1 (mppi_rollout_cost) and K3 (mppi_weights_partial) alone on config C3,
with the noise either re-read from one buffer (Infinity-Cache-warm) or rotated over NBUF buffers
(> 256 MiB in total, HBM-cold)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N
from pytorch_mppi_amd.mppi import _ptr

K, T, nx, nu = int(os.environ.get("K", 65536)), 64, 16, 12
NBUF = int(os.environ.get("NBUF", 4))
m = pm.models.Integrator(nx, nu)
c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda",
            lambda_=9000.0, U_init=torch.randn(T, nu) * 0.02, rng="torch-native")
x0 = torch.randn(nx, device="cuda")
c.command(x0)
lib = N.lib()
p = c._last
rows4 = N.noise_rows4(T, nu)
zs = [torch.randn(rows4 * K * 4, device="cuda") for _ in range(NBUF)]
st = c._stream()
alg = 4 * K * T * nu + 4 * K

def timeit(fn, n=30, rotate=True):
    for i in range(3):
        p.z = _ptr(zs[i % NBUF]); fn()
    torch.cuda.synchronize()
    evs = []
    for i in range(n):
        p.z = _ptr(zs[i % NBUF] if rotate else zs[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3, t[0] * 1e3

k1 = lambda: N.check(lib.mppi_rollout_cost(C.byref(p), st), "k1")
k3 = lambda: N.check(lib.mppi_weights_partial(C.byref(p), st), "k3")
for name, fn in (("K1 rollout_cost", k1), ("K3 weights_partial", k3)):
    for rot in (True, False):
        med, mn = timeit(fn, rotate=rot)
        print(f"{name:20s} {'HBM-cold (rotating %d bufs)' % NBUF if rot else 'cache-warm (1 buf)':28s} "
              f"median {med:7.1f} us  min {mn:7.1f} us  -> {alg / med / 1e3:7.1f} GB/s ({alg / med / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)")

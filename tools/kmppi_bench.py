"""KMPPI at C3-sized work (K=65536, T=64, nx=16, nu=12, S=32): the interpolation inside K1
(mppi_rollout_cost_kmppi) against the two-launch form (mppi_kmppi_interp + K1 on raw actions).
Per-command wall time and K1's device-clock span (the C-ABI measurement hook).
    python tools/kmppi_bench.py [rng] [K]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

rng = sys.argv[1] if len(sys.argv) > 1 else "philox"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nx, nu, T, S = 16, 12, 64, 32
m = pm.models.Integrator(nx, nu)
torch.manual_seed(0)
x0 = torch.randn(nx, device="cuda")
lib = N.lib()
for fuse, fill, upd in ((True, None, True), (True, None, False), (True, False, True), (False, None, False)):
    c = pm.KMPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_support_pts=S, kernel=pm.RBFKernel(sigma=2.0),
                 rng=rng, num_samples=K, horizon=T, device="cuda", lambda_=50.0)
    c.fuse_interpolation = fuse
    c.onchip_update = upd            # theta update reduced inside K1 (mppi_command_kmppi) / by the stand-alone K3
    c.philox_fill = fill          # None: generator launch for the support-point rows; False: K1 / K3 generate them in-kernel
    for _ in range(5):
        c.command(x0)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        c.command(x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    lib.mppi_profile_enable(1 << 30)
    for _ in range(20):
        c.command(x0)
    torch.cuda.synchronize()
    a, b, cn, ce = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_int64(0)
    N.check(lib.mppi_profile_read2(C.byref(a), C.byref(b), C.byref(cn), C.byref(ce)), "read2")
    lib.mppi_profile_enable(0)
    k1 = b.value / max(1, cn.value) * 1e3
    macs = K * T * S * nu
    print(f"KMPPI S={S} K={K} rng={rng} {'interpolation inside K1' if fuse else 'two launches         '}{' rows generated in-kernel' if fill is False else ''}{' theta update inside K1' if upd else ' stand-alone K3'}: {dt * 1e3:.4f} ms/command, "
          f"K1 device clock {k1:.1f} us" + (f" = {2 * macs / k1 / 1e6:.1f} TFLOP/s of interpolation (fp32 MFMA peak 157.3)" if fuse else ""),
          flush=True)

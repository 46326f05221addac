"""Dev helper (GPU): time K1 (mppi_rollout_cost) and K3 (mppi_weights_partial) alone on config C3
(K from the environment), with the noise either re-read from one buffer (Infinity-Cache-warm) or
rotated over NBUF buffers (NBUF x 4*K*T*nu bytes >> 256 MiB in total: HBM-cold).  Two clocks per
row: torch events around the launch (stream time, includes the dispatch) and, for K1, the kernel's
own span on the device wall clock (the C-ABI measurement hook: min workgroup entry .. max exit =
what rocprofv3 --kernel-trace reports).  MPPI_K1_DMA=0|15|30 selects the register-ring kernel or
the LDS-DMA ring depth."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N
from pytorch_mppi_amd.mppi import _ptr

K, T, nx, nu = int(os.environ.get("K", 65536)), 64, 16, 12
alg = 4 * K * T * nu + 4 * K
NBUF = int(os.environ.get("NBUF", max(4, -(-6 * (1 << 28) // alg))))      # >= 1.5 GiB cycled
m = pm.models.Integrator(nx, nu)
c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda",
            lambda_=9000.0, U_init=torch.randn(T, nu) * 0.02, rng="torch-native")
x0 = torch.randn(nx, device="cuda")
c.command(x0)
lib = N.lib()
p = c._last
if os.environ.get("DENSE_PITCH") == "1":         # A/B: the dense layout (row pitch = K) against the engine's padded pitch
    p.noise_pitch = K
    c._zpitch_cache = ((c.K_local, c.dtype), K)
zs = [torch.randn(c._zelems(T), device="cuda") for _ in range(NBUF)]
st = c._stream()


def timeit(fn, n=40, rotate=True, hook=False):
    for i in range(3):
        p.z = _ptr(zs[i % NBUF]); fn()
    torch.cuda.synchronize()
    if hook:
        lib.mppi_profile_enable(1 << 30)        # device-clock stamps on every launch, no HIP events
    evs = []
    for i in range(n):
        p.z = _ptr(zs[i % NBUF] if rotate else zs[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    dev = None
    if hook:
        a, b, cn, ce = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_int64(0)
        N.check(lib.mppi_profile_read2(C.byref(a), C.byref(b), C.byref(cn), C.byref(ce)), "read2")
        lib.mppi_profile_enable(0)
        dev = b.value / max(1, cn.value) * 1e3
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3, t[0] * 1e3, dev


k1 = lambda: N.check(lib.mppi_rollout_cost(C.byref(p), st), "k1")
k3 = lambda: N.check(lib.mppi_weights_partial(C.byref(p), st), "k3")
tag = f"K={K} NBUF={NBUF} MPPI_K1_DMA={os.environ.get('MPPI_K1_DMA', 'auto')} MPPI_K3_DMA={os.environ.get('MPPI_K3_DMA', 'auto')}"
only_cold = os.environ.get("ONLY_COLD") == "1"       # K1, rotating buffers only (clean rocprofv3 trace)
for name, fn, hook in ((("K1 rollout_cost", k1, True),) if only_cold else (("K1 rollout_cost", k1, True), ("K3 weights_partial", k3, False))):
    for rot in ((True,) if only_cold else (True, False)):
        med, mn, dev = timeit(fn, rotate=rot, hook=hook)
        d = f"device-clock avg {dev:7.1f} us -> {alg / dev / 1e3:7.1f} GB/s ({alg / dev / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)" if dev else ""
        print(f"[{tag}] {name:20s} {'HBM-cold (rotating)' if rot else 'cache-warm (1 buf)':20s} "
              f"events median {med:7.1f} min {mn:7.1f} us  {d}")

"""Where the fixed cost of a timed region goes (the driver times 20 commands between two synchronisations: 0.0958 ms per
step against a steady-state period of 0.0881 ms -- ~150 us that do not scale with the number of steps).
Runs the headline controller the way bench.py does and prints, for several step counts: wall time, per-step time, the
host-side return time of every command of one region, and the same with a few variations (no stamps, a tiny kernel queued
before the clock starts)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N

dev = torch.device("cuda", 0)
ctrl, x0, _ = bench.make_controller(pm, "c3", dev, "philox", None, 65536)
probe, _, _ = bench.make_controller(pm, "c3", dev, "philox", None, 65536)
probe.command(x0)
ctrl.lambda_ = float(probe.cost_total.float().std())
lib = N.lib()
for _ in range(10):
    ctrl.command(x0)
torch.cuda.synchronize()


def region(steps, stamps=False, prime=False):
    if stamps:
        lib.mppi_profile_enable(1 << 30)
    torch.cuda.synchronize()
    if prime:
        torch.empty(1, device=dev).zero_()
    t0 = time.perf_counter()
    ts = []
    for _ in range(steps):
        ctrl.command(x0)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if stamps:
        lib.mppi_profile_enable(0)
    return dt, ts


for steps in (1, 2, 5, 10, 20, 50, 200):
    best = min(region(steps)[0] for _ in range(5))
    print(f"steps {steps:4d}: {best * 1e6:9.1f} us total, {best / steps * 1e6:8.2f} us per step", flush=True)
d200 = min(region(200)[0] for _ in range(3))
d20 = min(region(20)[0] for _ in range(5))
per = (d200 - d20) / 180
print(f"steady-state period (200 vs 20 steps): {per * 1e6:.2f} us; fixed cost of a region: {(d20 - 20 * per) * 1e6:.1f} us")
dt, ts = region(20)
print("host return times of the 20 commands (us):", " ".join(f"{t * 1e6:.0f}" for t in ts), f"| synchronised at {dt * 1e6:.0f}")
for name, kw in (("with device-clock stamps (bench.py)", dict(stamps=True)), ("a tiny kernel queued before t0", dict(prime=True))):
    b = min(region(20, **kw)[0] for _ in range(5))
    print(f"20 steps, {name}: {b / 20 * 1e6:.2f} us per step")

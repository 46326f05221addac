"""Dev helper (GPU): the on-chip command around the multiples of 65536 samples (one wave per SIMD = 65536 samples per round of the
chip) against the streaming form at the same K -- the "tail" VERDICT r04 weak #5 / item 7a asks about -- for rng = philox, philox7
and torch (C3's T, nx, nu; pipelined ms per command over 100 commands)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm

T, nx, nu = 64, 16, 12
m = pm.models.Integrator(nx, nu)
x = torch.randn(nx, device="cuda")


def run(K, rng, onchip=None):
    c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=8000.0,
                U_init=torch.zeros(T, nu), rng=rng, seed=3)
    if onchip is not None:
        c.philox_onchip = onchip
    for _ in range(10):
        c.command(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        c.command(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100 * 1e3, c.last_draw


print(f"{'K':>8s} | {'philox on chip':>16s} | {'philox streaming':>16s} | {'philox7 on chip':>16s} | {'torch':>10s}   (ms per command; rollouts/s of the best philox form)")
for K in (49152, 65536, 73728, 81920, 98304, 114688, 131072, 163840, 196608, 262144):
    a, da = run(K, "philox", True)
    b, db = run(K, "philox", False)
    c7, d7 = run(K, "philox7", True)
    t, dt_ = run(K, "torch")
    print(f"{K:8d} | {a:16.4f} | {b:16.4f} | {c7:16.4f} | {t:10.4f}   {K / (min(a, b) * 1e-3):.3e}  default picks: {run(K, 'philox')[1]}", flush=True)

"""Dev helper (GPU): soak of the on-chip command with its spill array -- two controllers on the same stream of commands, one with the
array, one generating twice; every output compared bit for bit every 500 commands, 60000 commands; then two controllers of different
shapes sharing the device (their arrays must not interfere)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, pytorch_mppi_amd as pm
m = pm.models.Integrator(16, 12)
mk = lambda K, T, spill, seed=5: (lambda c: (setattr(c, "onchip_spill", spill), c)[1])(pm.MPPI(m.dynamics, m.running_cost, 16, torch.eye(12) * 0.5, num_samples=K, horizon=T, device="cuda", lambda_=30.0, rng="philox", seed=seed, U_init=torch.zeros(T, 12)))
a, b = mk(65536, 64, True), mk(65536, 64, False)
x = torch.randn(16, generator=torch.Generator().manual_seed(1)).cuda()
t0 = time.time(); bad = 0
for i in range(60000):
    ua, ub = a.command(x), b.command(x)
    if i % 500 == 0:
        ok = torch.equal(ua, ub) and torch.equal(a.U, b.U) and torch.equal(a.cost_total, b.cost_total)
        bad += 0 if ok else 1
        x = x + 0.01 * ua.mean()            # a moving state: every block of commands differs
torch.cuda.synchronize()
print(f"60000 commands x 2 controllers in {time.time() - t0:.1f} s: {bad} mismatching checks of 120; forms {a.last_draw} / {b.last_draw}")
c, d = mk(65536, 64, True, seed=9), mk(131072, 40, True, seed=11)
c2, d2 = mk(65536, 64, False, seed=9), mk(131072, 40, False, seed=11)
bad = 0
for i in range(3000):
    r = [q.command(x) for q in (c, d, c2, d2)]
    if i % 100 == 0:
        bad += 0 if (torch.equal(r[0], r[2]) and torch.equal(r[1], r[3])) else 1
print(f"two shapes interleaved, 3000 commands each: {bad} mismatching checks of 30; arrays {c._spill[1].numel() * 4 >> 20} MiB and {(d._spill[1].numel() * 4 >> 20) if d._spill[1] is not None else 0} MiB")
